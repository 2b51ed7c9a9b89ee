// c_abi_demo.cpp -- the C ABI of include/dcmrta_env.h used from plain C++/HIP, no Python and no torch:
// upload instances, reset, play one random-policy episode per env in the persistent kernel, read the summary.
//
//   hipcc --offload-arch=gfx950 -O2 examples/c_abi_demo.cpp -Iinclude -Ldcmrta_amd -ldcmrta_hip \
//         -Wl,-rpath,$PWD/dcmrta_amd -o examples/c_abi_demo && ./examples/c_abi_demo 256
//
// Instances here are drawn with a tiny LCG (the Python host mirrors numpy's default_rng draw order of the reference
// instead, dcmrta_amd/instances.py); seeds follow the choice protocol's env_seed(base, e).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dcmrta_env.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define DCM_OK_(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, dcm_last_error()); return 1; } } while (0)

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, A = 20, T = 50;
    std::vector<double> depot(2 * (size_t)B), xy(2 * (size_t)B * T), dur((size_t)B * T, 5.0);
    std::vector<int32_t> req((size_t)B * T);
    std::vector<uint64_t> seeds(B);
    uint64_t s = 12345;
    auto u01 = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; };
    for (auto& v : depot) v = u01();
    for (auto& v : xy) v = u01();
    for (auto& v : req) v = 1 + (int)(u01() * 5.0);
    for (int e = 0; e < B; e++) seeds[e] = mix64(0x9E3779B97F4A7C15ULL * (uint64_t)(e + 1));   // env_seed(0, e)

    double *d_depot, *d_xy, *d_dur, *d_sum;
    int32_t* d_req;
    uint64_t* d_seeds;
    int64_t* d_steps;
    HIP_OK(hipMalloc(&d_depot, depot.size() * 8)); HIP_OK(hipMalloc(&d_xy, xy.size() * 8)); HIP_OK(hipMalloc(&d_dur, dur.size() * 8));
    HIP_OK(hipMalloc(&d_req, req.size() * 4)); HIP_OK(hipMalloc(&d_seeds, seeds.size() * 8));
    HIP_OK(hipMalloc(&d_steps, (size_t)B * 8)); HIP_OK(hipMalloc(&d_sum, (size_t)B * 64));
    HIP_OK(hipMemcpy(d_depot, depot.data(), depot.size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_xy, xy.data(), xy.size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_dur, dur.data(), dur.size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_req, req.data(), req.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_seeds, seeds.data(), seeds.size() * 8, hipMemcpyHostToDevice));

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    dcm_params p = {B, A, T, 0, 10.0, 100.0, 0, 0};
    dcm_env* env = nullptr;
    DCM_OK_(dcm_create(&p, &env));
    DCM_OK_(dcm_load_instances(env, d_depot, d_xy, d_req, d_dur, st));
    DCM_OK_(dcm_reset(env, d_seeds, st));
    DCM_OK_(dcm_rollout_random(env, 1, -1, nullptr, nullptr, nullptr, nullptr, d_steps, st));
    DCM_OK_(dcm_summary(env, d_sum, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<int64_t> steps(B);
    std::vector<double> sum((size_t)B * 8);
    HIP_OK(hipMemcpy(steps.data(), d_steps, (size_t)B * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(sum.data(), d_sum, (size_t)B * 64, hipMemcpyDeviceToHost));
    long long total = 0;
    double mk = 0, fin = 0;
    for (int e = 0; e < B; e++) { total += steps[e]; mk += sum[(size_t)e * 8 + 3]; fin += sum[(size_t)e * 8 + 1]; }
    printf("{\"envs\": %d, \"decisions\": %lld, \"mean_makespan\": %.6f, \"mean_finished_tasks\": %.3f}\n", B, total, mk / B, fin / B);
    DCM_OK_(dcm_destroy(env));
    return 0;
}
