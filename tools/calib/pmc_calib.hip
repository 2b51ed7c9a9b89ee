// pmc_calib.hip -- calibrates the units of the SQ VALU counters used by bench.py's issue-bound roofline
// (SQ_ACTIVE_INST_VALU, SQ_THREAD_CYCLES_VALU, SQ_INSTS_VALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE) on gfx950.
// Three kernels with a known instruction stream: N dependent fp64 FMAs per wave executed by 64, 16 and 1 active lanes.
//   hipcc --offload-arch=gfx950 -O2 tools/calib/pmc_calib.hip -o /tmp/pmc_calib && rocprofv3 --pmc ... -- /tmp/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LANES>
__global__ __launch_bounds__(64) void k_calib(double* out, int n) {
    double x = (double)threadIdx.x * 1e-3, y = 1.0000001;
    if ((int)threadIdx.x < LANES) {
#pragma unroll 1
        for (int i = 0; i < n; i++) {
            // 16 dependent fp64 FMAs per iteration
#pragma unroll
            for (int j = 0; j < 16; j++) x = __builtin_fma(x, y, 1e-9);
        }
        out[blockIdx.x * 64 + threadIdx.x] = x;
    }
}

int main() {
    double* d;
    const int blocks = 4096, n = 4096;   // 4096 waves x 65536 FMAs
    if (hipMalloc(&d, blocks * 64 * sizeof(double)) != hipSuccess) return 1;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_calib<64>, dim3(blocks), dim3(64), 0, 0, d, n);
        hipLaunchKernelGGL(k_calib<16>, dim3(blocks), dim3(64), 0, 0, d, n);
        hipLaunchKernelGGL(k_calib<1>, dim3(blocks), dim3(64), 0, 0, d, n);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    printf("calib: %d waves x %d fp64 FMA instructions per wave per kernel\n", blocks, n * 16);
    return 0;
}
