// mfma_valu_probe.hip -- does gfx950 run f64 MFMA and f64 VALU FMA streams CONCURRENTLY, and what does the chip sustain?
// (MI355X: dense f64 MFMA peak = f64 vector FMA peak = 78.6 TFLOP/s; the two pipes are separate issue ports of a SIMD.)
// One workgroup of 512 threads per CU slot: waves 0-3 (one per SIMD) run a register-only chain of v_mfma_f64_16x16x4_f64,
// waves 4-7 (the second wave of each SIMD) a register-only chain of v_fma_f64.  Modes: 1 = MFMA waves only, 2 = VALU waves
// only, 3 = both.  Prints TFLOP/s per pipe and in total.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o tools/bin/mfma_valu_probe && tools/bin/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_probe(int mode, int iters, double* out, double seed) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4;
    if (mfma_wave && (mode & 1)) {
        d4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
        double a = seed + threadIdx.x * 1e-9, b = seed * 0.5 + threadIdx.x * 1e-9;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678) out[0] = s;
    } else if (!mfma_wave && (mode & 2)) {
        double c[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) c[i] = seed * i;
        double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-12;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 32; ++i) c[i] = __builtin_fma(c[i], b, a);
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += c[i];
        if (s == 12345.678) out[1] = s;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    double* out;
    hipMalloc(&out, 64);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
        for (int mode = 1; mode <= 3; ++mode) {
            const int grid = cus * wg_per_cu;
            hipLaunchKernelGGL(k_probe, dim3(grid), dim3(512), 0, 0, mode, 100, out, 1.0);   // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_probe, dim3(grid), dim3(512), 0, 0, mode, iters, out, 1.0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // per workgroup: 4 MFMA waves x iters x 8 MFMAs x 2048 flop; 4 VALU waves x iters x 128 FMAs x 64 lanes x 2 flop
            const double f_mfma = (mode & 1) ? (double)grid * 4 * iters * 8 * 2048.0 : 0.0;
            const double f_valu = (mode & 2) ? (double)grid * 4 * iters * 128 * 128.0 : 0.0;
            printf("%d CUs, %d workgroup(s) of 512 per CU, mode %d (%s): %.3f ms  MFMA %.1f TF/s  VALU %.1f TF/s  total %.1f TF/s\n", cus,
                   wg_per_cu, mode, mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both", ms, f_mfma / ms * 1e-9,
                   f_valu / ms * 1e-9, (f_mfma + f_valu) / ms * 1e-9);
        }
    return 0;
}
