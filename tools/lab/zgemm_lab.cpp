// Standalone timing harness for the library zgemm (kernel iteration without Python start-up).
//   zgemm_lab N 135491 259 259 [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/dftk_mi355x.h"
int zgemm_debug_clock(double* mhz, double* us);   // gemm_kernels.hip (GEMM_EXP_CLOCK builds)

__global__ void k_fill(double* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long x = (i + 1) * 6364136223846793005ull + seed * 1442695040888963407ull;
    x ^= x >> 29;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 32;
    p[i] = (double)(x & 0xFFFFFF) / 16777216.0 - 0.5;
}

int main(int argc, char** argv) {
    dftk_mi_basis* h = nullptr;
    if (dftk_mi_basis_create(8, 8, 8, 1.0, 0, &h) != 0) {
        fprintf(stderr, "basis_create failed: %s\n", dftk_mi_last_error());
        return 1;
    }
    if (argc > 1 && argv[1][0] == 'p') {   // peak <iters...>: MFMA ceiling + shader clock for growing durations
        for (int a = 2; a < argc; ++a)
            for (int w : {1, 2}) {
                double t = 0;
                dftk_mi_diag_mfma_peak(h, w, atoi(argv[a]), &t);
                printf("peak waves/SIMD=%d iters=%s: %.1f TF/s\n", w, argv[a], t);
            }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'c') {   // calibration for the TCC byte counters: 1 GiB device-to-device copy
        void *p1, *p2;
        hipMalloc(&p1, 1ull << 30);
        hipMalloc(&p2, 1ull << 30);
        hipMemset(p1, 1, 1ull << 30);
        hipMemcpy(p2, p1, 1ull << 30, hipMemcpyDeviceToDevice);
        hipDeviceSynchronize();
        argv += 1;
        argc -= 1;
    }
    for (int a = 1; a + 3 < argc; a += 4) {
        const char tr = argv[a][0];
        const long m = atol(argv[a + 1]), n = atol(argv[a + 2]), k = atol(argv[a + 3]);
        const size_t na = (size_t)m * k, nb = (size_t)k * n, nc = (size_t)m * n;
        double *A, *B, *C;
        hipMalloc(&A, na * 16);
        hipMalloc(&B, nb * 16);
        hipMalloc(&C, nc * 16);
        k_fill<<<(unsigned)((2 * na + 255) / 256), 256>>>(A, 2 * na, 1);
        k_fill<<<(unsigned)((2 * nb + 255) / 256), 256>>>(B, 2 * nb, 2);
        hipDeviceSynchronize();
        const long lda = tr == 'N' ? m : k;
        const dftk_mi_cplx one = {1, 0}, zero = {0, 0};
        double best = 1e30, sum = 0;
        const int reps = 6;
        for (int r = 0; r < reps; ++r) {
            dftk_mi_prof_enable(h, 1);
            if (dftk_mi_zgemm(h, tr, m, n, k, one, (const dftk_mi_cplx*)A, lda, (const dftk_mi_cplx*)B, k, zero,
                              (dftk_mi_cplx*)C, m) != 0) {
                fprintf(stderr, "zgemm failed: %s\n", dftk_mi_last_error());
                return 1;
            }
            double ms = 0, work = 0;
            int64_t nl = 0;
            dftk_mi_prof_get(h, 0, &ms, &work, &nl);
            if (r > 0) {
                sum += ms;
                if (ms < best) best = ms;
            }
        }
        const double fl = 8.0 * m * n * k;
        printf("%c m=%7ld n=%5ld k=%7ld: avg %8.3f ms %6.2f TF/s | best %8.3f ms %6.2f TF/s\n", tr, m, n, k,
               sum / (reps - 1), fl / (sum / (reps - 1) * 1e9), best, fl / (best * 1e9));
        if (getenv("DFTK_MI_GEMM_CLOCK")) {
            double mhz = 0, us = 0;
            zgemm_debug_clock(&mhz, &us);
            printf("    workgroup 0 of the last launch: %.1f us at %.0f MHz shader clock\n", us, mhz);
        }
        hipFree(A);
        hipFree(B);
        hipFree(C);
    }
    dftk_mi_basis_destroy(h);
    return 0;
}
