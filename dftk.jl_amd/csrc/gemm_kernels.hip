// gemm_kernels.hip -- complex fp64 GEMM on the f64 matrix cores (v_mfma_f64_16x16x4_f64, gfx950).
//
// Covers every BLAS3 call of the reference's hot path:
//   P' * psi, P * (D P' psi)                    src/terms/operators.jl:126-128          (K7, K9)
//   mul_hermi(Y', AY), BY' * X                  src/eigen/lobpcg_hyper_impl.jl:90-121,278 (K10, K15)
//   Y * cX, X * invR                            :124-132, 234                           (K12, K14)
// as  C = alpha * op(A) * B + beta * C  with op(A) in {A, A^H}, B never transposed, column-major.
//
// A complex product is four real MFMA streams accumulated into two tiles:
//   Cr += Ar*Br -/+ Ai*Bi ,  Ci += Ar*Bi +/- Ai*Br   (upper signs: A, lower signs: conj(A))
// Each lane loads one interleaved complex (16 B) per operand fragment straight from global /
// L2 -- the kernel is MFMA-bound (64 cycles per instruction per SIMD) so no LDS staging is
// needed to feed it.  One wave owns an (RM*16) x (RN*16) output tile, a workgroup is 4 waves
// stacked along m.  Long-K products (Gram matrices over n_G) are split along K into slabs that
// a second kernel reduces in a fixed order => bitwise reproducible results.
#include "common.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define GEMM_RM 2
#define GEMM_RN 4
#define GEMM_WAVES 4
#define GEMM_BM (GEMM_WAVES * GEMM_RM * 16)   // 128
#define GEMM_BN (GEMM_RN * 16)                // 64

// lane mapping of v_mfma_f64_16x16x4_f64:
//   A operand: lane l holds A[i = l & 15][k = l >> 4];  B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D: 4 values per lane, value r is C[row = (l >> 4) + 4 r][col = l & 15]
template <bool CONJA>
__global__ __launch_bounds__(GEMM_WAVES * 64) void k_zgemm_mfma(int m, int n, int K, int kchunk,
                                                                const cd* __restrict__ A, int64_t lda,
                                                                const cd* __restrict__ B, int64_t ldb,
                                                                cd* __restrict__ C, int64_t ldc, cd alpha, cd beta,
                                                                cd* __restrict__ slab) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.x * GEMM_BM + wave * (GEMM_RM * 16);
    const int j0 = blockIdx.y * GEMM_BN;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);

    v4d accR[GEMM_RM][GEMM_RN], accI[GEMM_RM][GEMM_RN];
#pragma unroll
    for (int a = 0; a < GEMM_RM; ++a)
#pragma unroll
        for (int b = 0; b < GEMM_RN; ++b) {
            accR[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            accI[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    // per-fragment base pointers (clamped so that every load stays in bounds)
    const cd* pa[GEMM_RM];
#pragma unroll
    for (int a = 0; a < GEMM_RM; ++a) {
        int i = i0 + a * 16 + li;
        if (i > m - 1) i = m - 1;
        pa[a] = CONJA ? (A + (int64_t)i * lda) : (A + i);
    }
    const cd* pb[GEMM_RN];
#pragma unroll
    for (int b = 0; b < GEMM_RN; ++b) {
        int j = j0 + b * 16 + li;
        if (j > n - 1) j = n - 1;
        pb[b] = B + (int64_t)j * ldb;
    }

    for (int k0 = kbeg; k0 < kend; k0 += 4) {
        const int kk = k0 + lk;
        const bool valid = kk < kend;
        const int kc = valid ? kk : (kend - 1);
        cd fa[GEMM_RM], fb[GEMM_RN];
#pragma unroll
        for (int a = 0; a < GEMM_RM; ++a) {
            cd v = CONJA ? pa[a][kc] : pa[a][(int64_t)kc * lda];
            fa[a] = valid ? v : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int b = 0; b < GEMM_RN; ++b) {
            cd v = pb[b][kc];
            fb[b] = valid ? v : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int a = 0; a < GEMM_RM; ++a) {
            const double ar = fa[a].x;
            const double ai = fa[a].y;
            const double nai = -ai;
#pragma unroll
            for (int b = 0; b < GEMM_RN; ++b) {
                accR[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, fb[b].x, accR[a][b], 0, 0, 0);
                accR[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(CONJA ? ai : nai, fb[b].y, accR[a][b], 0, 0, 0);
                accI[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, fb[b].y, accI[a][b], 0, 0, 0);
                accI[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(CONJA ? nai : ai, fb[b].x, accI[a][b], 0, 0, 0);
            }
        }
    }

    // epilogue
    const bool direct = (slab == nullptr);
    cd* sl = direct ? nullptr : slab + (int64_t)blockIdx.z * m * n;
#pragma unroll
    for (int a = 0; a < GEMM_RM; ++a)
#pragma unroll
        for (int b = 0; b < GEMM_RN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + a * 16 + lk + 4 * r;
                const int gj = j0 + b * 16 + li;
                if (gi < m && gj < n) {
                    const double vr = accR[a][b][r], vi = accI[a][b][r];
                    if (direct) {
                        cd* c = C + gi + (int64_t)gj * ldc;
                        cd o = make_double2(alpha.x * vr - alpha.y * vi, alpha.x * vi + alpha.y * vr);
                        if (beta.x != 0.0 || beta.y != 0.0) {
                            const cd old = *c;
                            o.x += beta.x * old.x - beta.y * old.y;
                            o.y += beta.x * old.y + beta.y * old.x;
                        }
                        *c = o;
                    } else {
                        sl[gi + (int64_t)gj * m] = make_double2(vr, vi);
                    }
                }
            }
}

// C = alpha * sum_z slab[z] + beta * C   (fixed summation order)
__global__ void k_zgemm_reduce(int m, int n, int nsplit, const cd* __restrict__ slab, cd* __restrict__ C,
                               int64_t ldc, cd alpha, cd beta) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)m * n) return;
    const int j = (int)(idx / m);
    const int i = (int)(idx - (int64_t)j * m);
    double sr = 0.0, si = 0.0;
    for (int z = 0; z < nsplit; ++z) {
        const cd v = slab[(int64_t)z * m * n + idx];
        sr += v.x;
        si += v.y;
    }
    cd* c = C + i + (int64_t)j * ldc;
    cd o = make_double2(alpha.x * sr - alpha.y * si, alpha.x * si + alpha.y * sr);
    if (beta.x != 0.0 || beta.y != 0.0) {
        const cd old = *c;
        o.x += beta.x * old.x - beta.y * old.y;
        o.y += beta.x * old.y + beta.y * old.x;
    }
    *c = o;
}

// Reference kernel without matrix cores (debug path, env DFTK_MI_GEMM=naive): one thread per C entry.
template <bool CONJA>
__global__ void k_zgemm_naive(int m, int n, int K, const cd* __restrict__ A, int64_t lda,
                              const cd* __restrict__ B, int64_t ldb, cd* __restrict__ C, int64_t ldc, cd alpha,
                              cd beta) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)m * n) return;
    const int j = (int)(idx / m);
    const int i = (int)(idx - (int64_t)j * m);
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < K; ++k) {
        cd a = CONJA ? A[k + (int64_t)i * lda] : A[i + (int64_t)k * lda];
        if (CONJA) a.y = -a.y;
        const cd b = B[k + (int64_t)j * ldb];
        sr += a.x * b.x - a.y * b.y;
        si += a.x * b.y + a.y * b.x;
    }
    cd* c = C + i + (int64_t)j * ldc;
    cd o = make_double2(alpha.x * sr - alpha.y * si, alpha.x * si + alpha.y * sr);
    if (beta.x != 0.0 || beta.y != 0.0) {
        const cd old = *c;
        o.x += beta.x * old.x - beta.y * old.y;
        o.y += beta.x * old.y + beta.y * old.x;
    }
    *c = o;
}

int ensure_ws(dftk_mi_basis* b, size_t bytes) {
    if (bytes <= b->ws_bytes) return 0;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->ws) HIPCHK(hipFree(b->ws));
    b->ws = nullptr;
    b->ws_bytes = 0;
    size_t want = bytes + bytes / 4;
    HIPCHK(hipMalloc(&b->ws, want));
    b->ws_bytes = want;
    return 0;
}

int zgemm(dftk_mi_basis* b, char transA, int64_t m, int64_t n, int64_t k, cd alpha, const cd* A, int64_t lda,
          const cd* B, int64_t ldb, cd beta, cd* C, int64_t ldc) {
    if (m <= 0 || n <= 0) return 0;
    const bool conja = (transA == 'C' || transA == 'c');
    if (!conja && !(transA == 'N' || transA == 'n')) {
        dftk_set_error("zgemm: transA must be 'N' or 'C'");
        return DFTK_MI_EINVAL;
    }
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return DFTK_MI_EINVAL;
    if (k <= 0) {   // C = beta * C : run the reduce kernel over zero slabs
        hipLaunchKernelGGL(k_zgemm_reduce, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, b->stream, (int)m,
                           (int)n, 0, (const cd*)nullptr, C, ldc, alpha, beta);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int slot = prof_begin(b, PROF_ZGEMM, 8.0 * (double)m * (double)n * (double)k);
    struct ProfGuard {
        dftk_mi_basis* b;
        int s;
        ~ProfGuard() { prof_end(b, s); }
    } guard{b, slot};
    if (!b->use_mfma) {
        const unsigned blocks = (unsigned)((m * n + 255) / 256);
        if (conja)
            hipLaunchKernelGGL(k_zgemm_naive<true>, dim3(blocks), dim3(256), 0, b->stream, (int)m, (int)n, (int)k, A,
                               lda, B, ldb, C, ldc, alpha, beta);
        else
            hipLaunchKernelGGL(k_zgemm_naive<false>, dim3(blocks), dim3(256), 0, b->stream, (int)m, (int)n, (int)k,
                               A, lda, B, ldb, C, ldc, alpha, beta);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int gm = (int)((m + GEMM_BM - 1) / GEMM_BM);
    const int gn = (int)((n + GEMM_BN - 1) / GEMM_BN);
    // split K so that the launch has ~2 workgroups per CU; only worth it for long K
    int nsplit = 1;
    const int64_t tiles = (int64_t)gm * gn;
    if (k >= 2048 && tiles < 512) {
        nsplit = (int)((512 + tiles - 1) / tiles);
        const int64_t max_by_k = k / 512;
        if (nsplit > max_by_k) nsplit = (int)max_by_k;
        if (nsplit > 256) nsplit = 256;
        if (nsplit < 1) nsplit = 1;
    }
    int kchunk = (int)((k + nsplit - 1) / nsplit);
    kchunk = (kchunk + 3) & ~3;
    nsplit = (int)((k + kchunk - 1) / kchunk);
    cd* slab = nullptr;
    if (nsplit > 1) {
        CHK(ensure_ws(b, (size_t)nsplit * m * n * sizeof(cd)));
        slab = (cd*)b->ws;
    }
    dim3 grid(gm, gn, nsplit);
    if (conja)
        hipLaunchKernelGGL(k_zgemm_mfma<true>, grid, dim3(GEMM_WAVES * 64), 0, b->stream, (int)m, (int)n, (int)k,
                           kchunk, A, lda, B, ldb, C, ldc, alpha, beta, slab);
    else
        hipLaunchKernelGGL(k_zgemm_mfma<false>, grid, dim3(GEMM_WAVES * 64), 0, b->stream, (int)m, (int)n, (int)k,
                           kchunk, A, lda, B, ldb, C, ldc, alpha, beta, slab);
    if (nsplit > 1) {
        hipLaunchKernelGGL(k_zgemm_reduce, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, b->stream, (int)m,
                           (int)n, nsplit, slab, C, ldc, alpha, beta);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
