// pmc_calib.hip -- calibrates the units of the SQ counters behind bench.py's issue-bound roofline on gfx950, per
// INSTRUCTION CLASS: what one wave64 instruction of each class costs in (a) SQ_ACTIVE_INST_VALU / _SCA / _LDS
// quad-cycles (the counter) and (b) shader clocks of SIMD time (s_memtime deltas at a known occupancy).
//
// Every kernel runs a known stream: N instructions of ONE class per wave, issued as 4 interleaved dependency chains, with 64
// active lanes unless the name says otherwise.  The launch puts W = 8 waves on every SIMD (8192 single-wave workgroups on
// 256 CUs x 4 SIMDs), which makes the stream throughput-bound, so
//     clocks per instruction per SIMD = kernel duration (HIP events) x shader clock / (W x N)
// with the shader clock = s_memtime delta / s_memrealtime delta x 100 MHz measured inside the same kernel.  (The per-wave
// s_memtime delta is NOT a throughput measure: the SIMD arbitrates by age, so the oldest waves run at their dependency-
// limited rate and finish first -- round-3 finding.)  The PMC passes (tools/calib/run_calib.sh) give SQ_INSTS_VALU,
// SQ_ACTIVE_INST_VALU, SQ_THREAD_CYCLES_VALU, SQ_INSTS_SALU, SQ_ACTIVE_INST_SCA, GRBM_GUI_ACTIVE ... per kernel.
//
//   hipcc --offload-arch=gfx950 -O2 tools/calib/pmc_calib.hip -o /tmp/pmc_calib && /tmp/pmc_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

enum Op { FMA_F64, ADD_F64, MIN_F64, CVT_F32_F64, FMA_F32, ADD_U32, CNDMASK_B32, MOV_B32, CMP_F64, MOV_DPP, MBCNT, READLANE,
          S_ADD_U32, S_MUL_I32, DS_READ_B64, DS_READ_B32, CNDMASK_SGPR, CNDMASK_MIX, CNDMASK_FRESH_VCC,
          MOV_B64, MIN_U32_DPP, CMP_U32_SGPR, CMP_F64_SGPR, BFE_U32, LSHL_B64, N_OPS };   // (round 6: the last six)
static const char* kNames[N_OPS] = {"v_fma_f64", "v_add_f64", "v_min_f64", "v_cvt_f32_f64", "v_fma_f32", "v_add_u32", "v_cndmask_b32",
                                    "v_mov_b32", "v_cmp_lt_f64", "v_mov_b32_dpp", "v_mbcnt_lo", "v_readlane_b32", "s_add_u32",
                                    "s_mul_i32", "ds_read_b64", "ds_read_b32", "v_cndmask_e64_sgpr", "v_cndmask+v_add_u32",
                                    "v_cmp+v_cndmask", "v_mov_b64", "v_min_u32_dpp", "v_cmp_lt_u32_e64", "v_cmp_lt_f64_e64", "v_bfe_u32",
                                    "v_lshlrev_b64"};

// one instruction of class OP on chain j (4 independent chains a[0..3]); inline asm so that the compiler can neither
// fuse, hoist nor reorder the stream
template <int OP>
__device__ __forceinline__ void one(double (&d)[4], float (&f)[4], unsigned (&u)[4], unsigned (&s)[4], int j, const unsigned char* lds) {
    if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 3]));
    if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 3]));
    if constexpr (OP == MIN_F64) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 3]));
    if constexpr (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(d[j]), "0"(f[j]));
    if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j]) : "v"(f[(j + 1) & 3]));
    if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 3]));
    if constexpr (OP == CNDMASK_B32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 3]) : );
    if constexpr (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 3]), "0"(u[j]));
    if constexpr (OP == CMP_F64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[j]), "v"(d[(j + 1) & 3]) : "vcc");
    if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[j]) : "v"(u[(j + 1) & 3]));
    if constexpr (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 3]));
    if constexpr (OP == READLANE) asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s[j]) : "v"(u[j]), "0"(s[j]));
    if constexpr (OP == S_ADD_U32) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[j]) : "s"(s[(j + 1) & 3]) : "scc");
    if constexpr (OP == S_MUL_I32) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s[j]) : "s"(s[(j + 1) & 3]));
    // v_cndmask_b32 variants: mask in an SGPR pair (VOP3), alternating with an integer add, mask freshly written by v_cmp
    if constexpr (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 3]), "s"(__builtin_amdgcn_read_exec()));
    if constexpr (OP == CNDMASK_MIX) {
        if (j & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 3]));
        else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 3]));
    }
    if constexpr (OP == CNDMASK_FRESH_VCC) {
        if (j & 1) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[j]), "v"(u[(j + 1) & 3]) : "vcc");
        else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 3]));
    }
    // round 6: the other classes the persistent kernels' decision loops are made of
    if constexpr (OP == MOV_B64) asm volatile("v_mov_b64 %0, %1" : "=v"(d[j]) : "v"(d[(j + 1) & 3]), "0"(d[j]));
    if constexpr (OP == MIN_U32_DPP) asm volatile("v_min_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[j]) : "v"(u[(j + 1) & 3]));
    if constexpr (OP == CMP_U32_SGPR) { unsigned long long m; asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(u[j]), "v"(u[(j + 1) & 3])); }
    if constexpr (OP == CMP_F64_SGPR) { unsigned long long m; asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(m) : "v"(d[j]), "v"(d[(j + 1) & 3])); }
    if constexpr (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[j]));
    if constexpr (OP == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[j]));
    if constexpr (OP == DS_READ_B64) asm volatile("ds_read_b64 %0, %1" : "=v"(d[j]) : "v"((unsigned)(threadIdx.x * 8 + j * 512)), "0"(d[j]) : "memory");
    if constexpr (OP == DS_READ_B32) asm volatile("ds_read_b32 %0, %1" : "=v"(u[j]) : "v"((unsigned)(threadIdx.x * 4 + j * 256)), "0"(u[j]) : "memory");
}

template <int OP, int LANES>
__global__ __launch_bounds__(64) void k_calib(unsigned long long* times, double* sink, int n) {
    __shared__ unsigned char lds[2048];
    double d[4];
    float f[4];
    unsigned u[4], s[4];
    for (int j = 0; j < 4; j++) {
        d[j] = 1.0 + 1e-9 * (threadIdx.x + j); f[j] = 1.0f + 1e-6f * (threadIdx.x + j); u[j] = threadIdx.x * 4 + j;
        s[j] = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x + j));
    }
    for (int i = threadIdx.x; i < 512; i += 64) ((unsigned*)lds)[i] = i;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if ((int)threadIdx.x < LANES) {
        t0 = __builtin_readcyclecounter();   // s_memtime: shader clock
        r0 = wall_clock64();                 // s_memrealtime: 100 MHz
#pragma unroll 1
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {   // 64 instructions per iteration, chains interleaved
                one<OP>(d, f, u, s, 0, lds); one<OP>(d, f, u, s, 1, lds); one<OP>(d, f, u, s, 2, lds); one<OP>(d, f, u, s, 3, lds);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_readcyclecounter();
        r1 = wall_clock64();
        sink[blockIdx.x * 64 + threadIdx.x] = d[0] + d[1] + d[2] + d[3] + f[0] + f[1] + f[2] + f[3] + (double)(u[0] ^ u[1] ^ u[2] ^ u[3]) +
                                              (double)(s[0] ^ s[1] ^ s[2] ^ s[3]);
    }
    if (threadIdx.x == 0) { times[2 * blockIdx.x] = t1 - t0; times[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP, int LANES>
static void run(unsigned long long* d_times, double* d_sink, int blocks, int n, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_calib<OP, LANES>), dim3(blocks), dim3(64), 0, 0, d_times, d_sink, n);
    hipEventRecord(e1, 0);
    std::vector<unsigned long long> h(2 * (size_t)blocks);
    if (hipMemcpy(h.data(), d_times, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { printf("memcpy failed\n"); return; }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; b++) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    const double mhz = cyc / real * 100.0;        // s_memtime ticks per 100 MHz s_memrealtime tick
    cyc /= blocks;
    const double insts = 64.0 * n;
    printf("%-20s lanes %2d  waves/SIMD %d  insts/wave %6.0f  kernel %8.1f us  shader clock %5.0f MHz  SIMD clocks/inst %6.2f  "
           "(one wave alone in its region: %6.2f clocks/inst)\n",
           kNames[OP], LANES, waves_per_simd, insts, ms * 1e3, mhz, ms * 1e-3 * mhz * 1e6 / (insts * waves_per_simd), cyc / insts);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const int W = 8, blocks = 256 * 4 * W, n = 1024;   // 8192 waves x 65536 instructions
    unsigned long long* d_times;
    double* d_sink;
    if (hipMalloc(&d_times, 2 * (size_t)blocks * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMalloc(&d_sink, (size_t)blocks * 64 * sizeof(double)) != hipSuccess) return 1;
    run<FMA_F64, 64>(d_times, d_sink, blocks, n, W);   // warm-up (clocks up)
    for (int rep = 0; rep < 2; rep++) {
        run<FMA_F64, 64>(d_times, d_sink, blocks, n, W);
        run<FMA_F64, 32>(d_times, d_sink, blocks, n, W);
        run<FMA_F64, 16>(d_times, d_sink, blocks, n, W);
        run<FMA_F64, 8>(d_times, d_sink, blocks, n, W);
        run<FMA_F64, 4>(d_times, d_sink, blocks, n, W);
        run<FMA_F64, 2>(d_times, d_sink, blocks, n, W);
        run<FMA_F64, 1>(d_times, d_sink, blocks, n, W);
        run<ADD_F64, 64>(d_times, d_sink, blocks, n, W);
        run<MIN_F64, 64>(d_times, d_sink, blocks, n, W);
        run<CMP_F64, 64>(d_times, d_sink, blocks, n, W);
        run<CVT_F32_F64, 64>(d_times, d_sink, blocks, n, W);
        run<FMA_F32, 64>(d_times, d_sink, blocks, n, W);
        run<ADD_U32, 64>(d_times, d_sink, blocks, n, W);
        run<ADD_U32, 16>(d_times, d_sink, blocks, n, W);
        run<ADD_U32, 4>(d_times, d_sink, blocks, n, W);
        run<ADD_U32, 1>(d_times, d_sink, blocks, n, W);
        run<CNDMASK_B32, 64>(d_times, d_sink, blocks, n, W);
        run<CNDMASK_SGPR, 64>(d_times, d_sink, blocks, n, W);
        run<CNDMASK_MIX, 64>(d_times, d_sink, blocks, n, W);
        run<CNDMASK_FRESH_VCC, 64>(d_times, d_sink, blocks, n, W);
        run<MOV_B32, 64>(d_times, d_sink, blocks, n, W);
        run<MOV_DPP, 64>(d_times, d_sink, blocks, n, W);
        run<MBCNT, 64>(d_times, d_sink, blocks, n, W);
        run<READLANE, 64>(d_times, d_sink, blocks, n, W);
        run<S_ADD_U32, 64>(d_times, d_sink, blocks, n, W);
        run<S_MUL_I32, 64>(d_times, d_sink, blocks, n, W);
        run<DS_READ_B64, 64>(d_times, d_sink, blocks, n, W);
        run<DS_READ_B32, 64>(d_times, d_sink, blocks, n, W);
        run<MOV_B64, 64>(d_times, d_sink, blocks, n, W);
        run<MIN_U32_DPP, 64>(d_times, d_sink, blocks, n, W);
        run<CMP_U32_SGPR, 64>(d_times, d_sink, blocks, n, W);
        run<CMP_F64_SGPR, 64>(d_times, d_sink, blocks, n, W);
        run<BFE_U32, 64>(d_times, d_sink, blocks, n, W);
        run<LSHL_B64, 64>(d_times, d_sink, blocks, n, W);
        // the product's occupancy: 4 waves per SIMD (4096 single-wave workgroups)
        run<FMA_F64, 64>(d_times, d_sink, blocks / 2, n, W / 2);
        run<ADD_U32, 64>(d_times, d_sink, blocks / 2, n, W / 2);
        run<S_ADD_U32, 64>(d_times, d_sink, blocks / 2, n, W / 2);
        run<MOV_B64, 64>(d_times, d_sink, blocks / 2, n, W / 2);
        run<MOV_DPP, 64>(d_times, d_sink, blocks / 2, n, W / 2);
        run<READLANE, 64>(d_times, d_sink, blocks / 2, n, W / 2);
        run<CNDMASK_SGPR, 64>(d_times, d_sink, blocks / 2, n, W / 2);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return 0;
}
