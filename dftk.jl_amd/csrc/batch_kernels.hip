// batch_kernels.hip -- batched forms of the small device operations of LOBPCG for many small k-blocks (batch.h):
// one launch over all k-blocks of a scheduling round instead of one launch per k-block.  Every kernel takes a table
// of per-item arguments (staged through the round's pinned ring) and a grid whose z dimension is the item index; x / y
// are sized for the largest item, smaller items let their surplus workgroups exit at once.
//
//   element-wise / column kernels   the arithmetic of dense_kernels.hip's k_col_reduce, k_residual, k_tpa, ... per item
//   small products                  C = alpha A^H B + beta C  (m, n <= 96; the long dimension is reduced inside ONE
//                                   workgroup per 16 x 16 tile: deterministic) and C = alpha A B + beta C (inner
//                                   dimension <= 128), honouring zgemm()'s UPPER / B_UPPER flags
//   Cholesky + inverse + normest    n <= 64, one workgroup per matrix, everything in LDS
//                                   (safe_cholesky, src/eigen/lobpcg_hyper_impl.jl:190-212)
//   Hermitian eigensolver           n <= 64, one workgroup per matrix: cyclic Jacobi with round-robin ordering in LDS
//                                   (rayleigh_ritz, :141-171), eigenvalues ascending
// These are latency problems (n_G ~ 1e3, 6-8 bands, matrices of order <= 3 M): no roofline applies; what counts is
// launches and host synchronisations per LOBPCG iteration.
#include "batch.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace dftk_batch {   // (a NAMED namespace: kernels of an anonymous one show up without names in rocprofv3 traces)

__device__ __forceinline__ double b_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// block-wide sum for 256 threads; result valid in thread 0
__device__ __forceinline__ double b_block_sum(double v, double* sh) {
    v = b_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}

struct EwItem {
    int64_t n, lda, ldb, ldc;
    int m, mode, flags, i0;
    const void *A, *B, *W, *W2, *W3;
    void *C, *D, *E, *F;
    double s0;
    size_t bytes;
};

// ---- column reductions / residual / TPA: one workgroup per (column, item), same trees as the per-block kernels ----
__global__ __launch_bounds__(256) void k_b_colred(const EwItem* __restrict__ items) {
    const EwItem it = items[blockIdx.z];
    const int c = blockIdx.x;
    if (c >= it.m) return;
    __shared__ double sh[4];
    const cd* x = reinterpret_cast<const cd*>(it.A) + (int64_t)c * it.lda;
    const cd* y = it.B ? reinterpret_cast<const cd*>(it.B) + (int64_t)c * it.ldb : nullptr;
    const double* w = reinterpret_cast<const double*>(it.W);
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < it.n; i += 256) {
        const cd a = x[i];
        if (it.mode == 1) {
            const cd bb = y[i];
            acc += a.x * bb.x + a.y * bb.y;
        } else if (it.mode == 4) {
            const cd bb = y[i];
            acc += a.x * bb.y - a.y * bb.x;
        } else if (it.mode == 2) {
            acc += w[i] * (a.x * a.x + a.y * a.y);
        } else {
            acc += a.x * a.x + a.y * a.y;
        }
    }
    const double r = b_block_sum(acc, sh);
    if (threadIdx.x == 0) reinterpret_cast<double*>(it.C)[c] = (it.mode == 0) ? sqrt(r) : r;
}

__global__ __launch_bounds__(256) void k_b_residual(const EwItem* __restrict__ items) {
    const EwItem it = items[blockIdx.z];
    const int c = blockIdx.x;
    if (c >= it.m) return;
    __shared__ double sh[4];
    const cd* AX = reinterpret_cast<const cd*>(it.A);
    const cd* X = reinterpret_cast<const cd*>(it.B);
    cd* R = reinterpret_cast<cd*>(it.C);
    const double* kin = reinterpret_cast<const double*>(it.W2);
    // (W3: the Rayleigh quotients of the start block, lam = <x, Ax> / <x, x>, formed here instead of on the host)
    const double l = it.W3 ? reinterpret_cast<const double*>(it.W)[c] / reinterpret_cast<const double*>(it.W3)[c]
                           : reinterpret_cast<const double*>(it.W)[c];
    double acc = 0.0, acck = 0.0, accx = 0.0;
    for (int64_t i = threadIdx.x; i < it.n; i += 256) {
        const cd a = AX[(int64_t)c * it.lda + i];
        const cd x = X[(int64_t)c * it.ldb + i];
        const cd r = make_double2(a.x - l * x.x, a.y - l * x.y);
        R[(int64_t)c * it.ldc + i] = r;
        acc += r.x * r.x + r.y * r.y;
        const double x2 = x.x * x.x + x.y * x.y;
        accx += x2;
        if (kin) acck += kin[i] * x2;
    }
    const double s = b_block_sum(acc, sh);
    const double sk = b_block_sum(acck, sh);
    const double sx = b_block_sum(accx, sh);
    if (threadIdx.x == 0) {
        reinterpret_cast<double*>(it.D)[c] = sqrt(s);
        if (kin) reinterpret_cast<double*>(it.E)[c] = sk;
        if (it.F) reinterpret_cast<double*>(it.F)[c] = sx;
    }
}

__global__ __launch_bounds__(256) void k_b_tpa(const EwItem* __restrict__ items) {
    const EwItem it = items[blockIdx.z];
    const int c = blockIdx.x;
    if (c >= it.m) return;
    __shared__ double sh[4];
    const cd* src = reinterpret_cast<const cd*>(it.A);
    cd* dst = reinterpret_cast<cd*>(it.C);
    const double* kin = reinterpret_cast<const double*>(it.W);
    const double* mean_kin = reinterpret_cast<const double*>(it.W2);
    const double mk = (kin && mean_kin) ? mean_kin[c] : 0.0;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < it.n; i += 256) {
        cd r = src[(int64_t)c * it.lda + i];
        if (kin) {
            const double f = mean_kin ? mk / (mk + kin[i]) : 1.0 / (kin[i] + it.s0);
            r.x *= f;
            r.y *= f;
        }
        dst[(int64_t)c * it.ldc + i] = r;
        acc += r.x * r.x + r.y * r.y;
    }
    const double s = b_block_sum(acc, sh);
    if (threadIdx.x == 0) reinterpret_cast<double*>(it.D)[c] = sqrt(s);
}

// ---- element-wise: grid (row blocks of the largest item, columns of the widest item, items) ----
// kind 0: scale columns (s or 1/s), 1: copy, 2: gather columns through perm, 3: zero fill of `bytes`
__global__ __launch_bounds__(256) void k_b_rows(const EwItem* __restrict__ items, int kind) {
    const EwItem it = items[blockIdx.z];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (kind == 3) {
        double* p = reinterpret_cast<double*>(it.C);
        const int64_t nd = (int64_t)(it.bytes / sizeof(double));
        for (int64_t t = i; t < nd; t += (int64_t)gridDim.x * 256) p[t] = 0.0;
        return;
    }
    if (c >= it.m || i >= it.n) return;
    cd* C = reinterpret_cast<cd*>(it.C);
    if (kind == 0) {
        const double s = reinterpret_cast<const double*>(it.W)[c];
        const double f = it.flags ? 1.0 / s : s;
        cd v = C[(int64_t)c * it.ldc + i];
        v.x *= f;
        v.y *= f;
        C[(int64_t)c * it.ldc + i] = v;
    } else {
        const cd* A = reinterpret_cast<const cd*>(it.A);
        const int sc = kind == 2 ? reinterpret_cast<const int*>(it.W)[c] : c;
        C[(int64_t)c * it.ldc + i] = A[(int64_t)sc * it.lda + i];
    }
}

// kind 0: C[i0 + a, a] -= 1 (a < m, i0 + a < n); 1: C[i, i] += s0 (i < m); 2: hermitise from the upper triangle (m x m);
// 3: C = A^H (m x m)
__global__ __launch_bounds__(256) void k_b_small(const EwItem* __restrict__ items, int kind) {
    const EwItem it = items[blockIdx.z];
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    cd* C = reinterpret_cast<cd*>(it.C);
    if (kind == 0) {
        if (idx < it.m && it.i0 + idx < it.n) C[(it.i0 + idx) + idx * it.ldc].x -= 1.0;
    } else if (kind == 1) {
        if (idx < it.m) C[idx + idx * it.ldc].x += it.s0;
    } else {
        const int n = it.m;
        if (idx >= (int64_t)n * n) return;
        const int j = (int)(idx / n), i = (int)(idx - (int64_t)j * n);
        if (kind == 2) {
            if (i == j) C[i + (int64_t)j * it.ldc].y = 0.0;
            if (i < j) {
                const cd v = C[i + (int64_t)j * it.ldc];
                C[j + (int64_t)i * it.ldc] = make_double2(v.x, -v.y);
            }
        } else {
            const cd v = reinterpret_cast<const cd*>(it.A)[j + (int64_t)i * it.lda];
            C[i + (int64_t)j * it.ldc] = make_double2(v.x, -v.y);
        }
    }
}

// ---- host <-> device traffic of a round: many tiny copies as ONE copy + a scatter / gather kernel ----
struct CopyItem {
    const void* src;
    void* dst;
    int64_t words;   // 4-byte words
};
__global__ __launch_bounds__(256) void k_b_copy_words(const CopyItem* __restrict__ items) {
    const CopyItem it = items[blockIdx.y];
    const uint32_t* s = reinterpret_cast<const uint32_t*>(it.src);
    uint32_t* d = reinterpret_cast<uint32_t*>(it.dst);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < it.words; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}

// ---- small products ----
struct GemmItem {
    int m, n, k, flags;
    int64_t lda, ldb, ldc;
    const cd *A, *B;
    cd* C;
    cd alpha, beta;
};
#define BG_T 16
#define BG_KC 64
// C (m x n) = alpha A^H B + beta C, A: k x m, B: k x n (k = the long dimension).  One workgroup per 16 x 16 tile of C,
// the whole k range inside it (fixed order: bitwise reproducible).  flags & 1 (UPPER): tiles strictly below the
// diagonal are skipped and left untouched, as zgemm() does.
// nsplit > 1: the k range is cut into nsplit equal chunks (blockIdx.y), every chunk writes its partial 16 x 16 tile to
// part[(item * nsplit + chunk) * pm * pn + ...]; k_b_gemm_c_reduce adds them in chunk order (deterministic) and applies
// alpha / beta.  One chunk of a 4 653-row product (graphene) is 5 LDS tiles instead of 73 in a row.
__global__ __launch_bounds__(256) void k_b_gemm_c(const GemmItem* __restrict__ items, int tiles_n, int nsplit,
                                                  cd* __restrict__ part, int pm, int pn) {
    GemmItem it = items[blockIdx.z];
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int i0 = tm * BG_T, j0 = tn * BG_T;
    if (i0 >= it.m || j0 >= it.n) return;
    if ((it.flags & 1) && i0 > j0 + BG_T - 1) return;
    int kbeg = 0;
    if (nsplit > 1) {
        const int kc = ((it.k + nsplit - 1) / nsplit + BG_KC - 1) / BG_KC * BG_KC;
        kbeg = blockIdx.y * kc;
        it.k = min(it.k, kbeg + kc);      // (an empty chunk writes zeros)
    }
    __shared__ cd As[BG_KC][BG_T + 1];
    __shared__ cd Bs[BG_KC][BG_T + 1];
    const int tid = threadIdx.x;
    const int ti = tid & 15, tj = tid >> 4;
    const int lr = tid & 63, lc = tid >> 6;   // loader: row lr of the chunk, columns lc, lc + 4, ...
    double ar = 0.0, ai = 0.0;
    for (int k0 = kbeg; k0 < it.k; k0 += BG_KC) {
        const int kr = k0 + lr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lc + 4 * q;
            cd va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
            if (kr < it.k) {
                if (i0 + c < it.m) va = it.A[kr + (int64_t)(i0 + c) * it.lda];
                if (j0 + c < it.n) vb = it.B[kr + (int64_t)(j0 + c) * it.ldb];
            }
            As[lr][c] = va;
            Bs[lr][c] = vb;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < BG_KC; ++r) {
            const cd a = As[r][ti], bb = Bs[r][tj];
            ar += a.x * bb.x + a.y * bb.y;     // conj(a) * b
            ai += a.x * bb.y - a.y * bb.x;
        }
        __syncthreads();
    }
    const int i = i0 + ti, j = j0 + tj;
    if (nsplit > 1) {
        if (i < it.m && j < it.n)
            part[((size_t)blockIdx.z * nsplit + blockIdx.y) * pm * pn + i + (size_t)j * pm] = make_double2(ar, ai);
        return;
    }
    if (i < it.m && j < it.n) {
        cd* cp = it.C + i + (int64_t)j * it.ldc;
        cd out = make_double2(it.alpha.x * ar - it.alpha.y * ai, it.alpha.x * ai + it.alpha.y * ar);
        if (it.beta.x != 0.0 || it.beta.y != 0.0) {
            const cd o = *cp;
            out.x += it.beta.x * o.x - it.beta.y * o.y;
            out.y += it.beta.x * o.y + it.beta.y * o.x;
        }
        *cp = out;
    }
}

__global__ __launch_bounds__(256) void k_b_gemm_c_reduce(const GemmItem* __restrict__ items, int nsplit,
                                                         const cd* __restrict__ part, int pm, int pn) {
    const GemmItem it = items[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= it.m * it.n) return;
    const int j = idx / it.m, i = idx - j * it.m;
    if ((it.flags & 1) && (i / BG_T) * BG_T > (j / BG_T) * BG_T + BG_T - 1) return;   // tile skipped by UPPER: untouched
    double ar = 0.0, ai = 0.0;
    for (int z = 0; z < nsplit; ++z) {
        const cd v = part[((size_t)blockIdx.y * nsplit + z) * pm * pn + i + (size_t)j * pm];
        ar += v.x;
        ai += v.y;
    }
    cd* cp = it.C + i + (int64_t)j * it.ldc;
    cd out = make_double2(it.alpha.x * ar - it.alpha.y * ai, it.alpha.x * ai + it.alpha.y * ar);
    if (it.beta.x != 0.0 || it.beta.y != 0.0) {
        const cd o = *cp;
        out.x += it.beta.x * o.x - it.beta.y * o.y;
        out.y += it.beta.x * o.y + it.beta.y * o.x;
    }
    *cp = out;
}

#define BN_C 8
#define BN_KMAX 128
// C (m x n) = alpha A B + beta C, A: m x k (m = the long dimension, k <= 128), B: k x n.  One thread per row, 8
// columns per workgroup column block; flags & 2 (B_UPPER): B[kk][j] = 0 for kk > j (never read, as zgemm()).
__global__ __launch_bounds__(256) void k_b_gemm_n(const GemmItem* __restrict__ items) {
    const GemmItem it = items[blockIdx.z];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * BN_C;
    if (j0 >= it.n || (int64_t)blockIdx.x * 256 >= it.m) return;
    __shared__ cd Bs[BN_KMAX][BN_C];
    for (int t = threadIdx.x; t < it.k * BN_C; t += 256) {
        const int kk = t / BN_C, jj = t - kk * BN_C;
        const int j = j0 + jj;
        cd v = make_double2(0.0, 0.0);
        if (j < it.n && !((it.flags & 2) && kk > j)) v = it.B[kk + (int64_t)j * it.ldb];
        Bs[kk][jj] = v;
    }
    __syncthreads();
    if (r >= it.m) return;
    double accr[BN_C], acci[BN_C];
#pragma unroll
    for (int jj = 0; jj < BN_C; ++jj) accr[jj] = acci[jj] = 0.0;
    // triangular B: column block j0 needs kk <= j0 + 7 only
    const int kend = (it.flags & 2) ? min(it.k, j0 + BN_C) : it.k;
    for (int kk = 0; kk < kend; ++kk) {
        const cd a = it.A[r + (int64_t)kk * it.lda];
#pragma unroll
        for (int jj = 0; jj < BN_C; ++jj) {
            const cd bb = Bs[kk][jj];
            accr[jj] += a.x * bb.x - a.y * bb.y;
            acci[jj] += a.x * bb.y + a.y * bb.x;
        }
    }
#pragma unroll
    for (int jj = 0; jj < BN_C; ++jj) {
        const int j = j0 + jj;
        if (j >= it.n) break;
        cd* cp = it.C + r + (int64_t)j * it.ldc;
        cd out = make_double2(it.alpha.x * accr[jj] - it.alpha.y * acci[jj], it.alpha.x * acci[jj] + it.alpha.y * accr[jj]);
        if (it.beta.x != 0.0 || it.beta.y != 0.0) {
            const cd o = *cp;
            out.x += it.beta.x * o.x - it.beta.y * o.y;
            out.y += it.beta.x * o.y + it.beta.y * o.x;
        }
        *cp = out;
    }
}

// ---- Cholesky + inverse + normest, one workgroup per matrix (n <= 64), everything in LDS ----
struct DenseItem {
    int n;
    int64_t lda, ldb;
    cd *A, *B;        // POTRF: A in / R out (upper), B = inv(R) out;  HEEV: A in (destroyed), B = eigenvectors out
    double* res;      // POTRF: 8 doubles {info, max|diag R|, sum|offdiag R|^2, bad R, same three for inv R, 0}
                      // HEEV : n eigenvalues ascending + {converged, non-finite}
    double* ev;       // HEEV : optional DEVICE copy of the eigenvalues (read by later kernels of the same round)
};
extern __shared__ __attribute__((aligned(16))) char b_smem[];

__global__ __launch_bounds__(256) void k_b_potrf(const DenseItem* __restrict__ items, int pitch) {
    const DenseItem it = items[blockIdx.x];
    const int n = it.n, tid = threadIdx.x, nthr = blockDim.x;   // 256 threads, or ONE wave for n <= 32 (cheap barriers)
    cd* S = reinterpret_cast<cd*>(b_smem);            // S[i * pitch + j]: row i, column j (upper part used)
    cd* Iv = S + (size_t)pitch * pitch;
    __shared__ double sh[4];
    __shared__ int s_info;
    __shared__ double s_piv;
    if (tid == 0) s_info = 0;
    for (int t = tid; t < n * n; t += nthr) {
        const int j = t / n, i = t - j * n;
        S[i * pitch + j] = it.A[i + (int64_t)j * it.lda];
        Iv[i * pitch + j] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        if (tid == 0) {
            const cd d = S[j * pitch + j];
            if (!(d.x > 0.0) || !isfinite(d.x) || !isfinite(d.y)) {
                if (s_info == 0) s_info = j + 1;
                s_piv = 1.0;
            } else {
                s_piv = sqrt(d.x);
                S[j * pitch + j] = make_double2(s_piv, 0.0);
            }
        }
        __syncthreads();
        if (s_info != 0) break;
        const double inv = 1.0 / s_piv;
        for (int c = j + 1 + tid; c < n; c += nthr) {
            cd v = S[j * pitch + c];
            v.x *= inv;
            v.y *= inv;
            S[j * pitch + c] = v;
        }
        __syncthreads();
        // trailing update of the upper triangle: S[i][c] -= conj(S[j][i]) S[j][c],  j < i <= c < n
        const int rem = n - j - 1;
        for (int t = tid; t < rem * rem; t += nthr) {
            const int a = t / rem, bq = t - a * rem;
            const int i = j + 1 + a, c = j + 1 + bq;
            if (i <= c) {
                const cd u = S[j * pitch + i], v = S[j * pitch + c];
                cd w = S[i * pitch + c];
                w.x -= u.x * v.x + u.y * v.y;
                w.y -= u.x * v.y - u.y * v.x;
                S[i * pitch + c] = w;
            }
        }
        __syncthreads();
    }
    const int info = s_info;
    if (info == 0) {
        // inverse of the upper triangular factor, one thread per column: x_j = 1 / R_jj, x_i = -(sum_{k>i} R_ik x_k) / R_ii
        if (tid < n) {
            const int j = tid;
            Iv[j * pitch + j] = make_double2(1.0 / S[j * pitch + j].x, 0.0);
            for (int i = j - 1; i >= 0; --i) {
                double sr = 0.0, si = 0.0;
                for (int k = i + 1; k <= j; ++k) {
                    const cd rr = S[i * pitch + k], x = Iv[k * pitch + j];
                    sr += rr.x * x.x - rr.y * x.y;
                    si += rr.x * x.y + rr.y * x.x;
                }
                const double d = S[i * pitch + i].x;
                Iv[i * pitch + j] = make_double2(-sr / d, -si / d);
            }
        }
        __syncthreads();
    }
    // write back R (upper triangle of A) and inv(R) (upper, zeros below); normest pieces of both
    double mx[2] = {0.0, 0.0}, off[2] = {0.0, 0.0}, bad[2] = {0.0, 0.0};
    for (int t = tid; t < n * n; t += nthr) {
        const int j = t / n, i = t - j * n;
        if (i <= j) {
            const cd r = S[i * pitch + j], x = Iv[i * pitch + j];
            it.A[i + (int64_t)j * it.lda] = r;
            it.B[i + (int64_t)j * it.ldb] = x;
            const cd v2[2] = {r, x};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!(isfinite(v2[q].x) && isfinite(v2[q].y))) bad[q] = 1.0;
                const double a2 = v2[q].x * v2[q].x + v2[q].y * v2[q].y;
                if (i == j)
                    mx[q] = fmax(mx[q], sqrt(a2));
                else
                    off[q] += a2;
            }
        } else {
            it.B[i + (int64_t)j * it.ldb] = make_double2(0.0, 0.0);
        }
    }
    __shared__ double smax[2][256];
    smax[0][tid] = mx[0];
    smax[1][tid] = mx[1];
    __syncthreads();
    const double o0 = b_block_sum(off[0], sh), o1 = b_block_sum(off[1], sh);
    const double b0 = b_block_sum(bad[0], sh), b1 = b_block_sum(bad[1], sh);
    if (tid == 0) {
        double m0 = 0.0, m1 = 0.0;
        for (int i = 0; i < nthr; ++i) {
            m0 = fmax(m0, smax[0][i]);
            m1 = fmax(m1, smax[1][i]);
        }
        it.res[0] = (double)info;
        it.res[1] = m0;
        it.res[2] = o0;
        it.res[3] = b0;
        it.res[4] = m1;
        it.res[5] = o1;
        it.res[6] = b1;
        it.res[7] = 0.0;
    }
}

// ---- Hermitian eigensolver, one workgroup per matrix (n <= 64): cyclic Jacobi, round-robin ordering, in LDS ----
// Rotation of the pair (p, q) as in dense_kernels.hip (k_jacobi_round): with a = S_pp, g = S_qq, b = S_pq,
// d = (g - a) / 2: t = sgn(d) / (|d| + sqrt(d^2 + |b|^2)), c = 1 / sqrt(1 + t^2 |b|^2), s = t c b;
// rows (p, q) <- (c row_p - s row_q, conj(s) row_p + c row_q), columns (p, q) <- (c col_p - conj(s) col_q, s col_p + c col_q).
__device__ __forceinline__ void rr_pair(int np, int round, int k, int& p, int& q) {
    // round-robin: player np - 1 stays, the others rotate
    if (k == 0) {
        p = np - 1;
        q = round;
    } else {
        p = (round + k) % (np - 1);
        q = (round - k + (np - 1)) % (np - 1);
    }
    if (p > q) {
        const int t = p;
        p = q;
        q = t;
    }
}
__global__ __launch_bounds__(256) void k_b_heev(const DenseItem* __restrict__ items, int pitch) {
    const DenseItem it = items[blockIdx.x];
    const int n = it.n, tid = threadIdx.x, nthr = blockDim.x;   // 256 threads, or ONE wave for n <= 32 (cheap barriers)
    const int np = n + (n & 1);                     // padded to even; the pad is a decoupled large diagonal entry
    cd* S = reinterpret_cast<cd*>(b_smem);          // S[i * pitch + j]
    cd* V = S + (size_t)pitch * pitch;
    __shared__ double sh[4];
    __shared__ double s_c[32];
    __shared__ cd s_s[32];
    __shared__ int s_p[32], s_q[32];
    __shared__ double s_red[2];
    double dg = 0.0, of = 0.0;
    for (int t = tid; t < np * np; t += nthr) {
        const int j = t / np, i = t - j * np;
        cd v = make_double2(0.0, 0.0);
        if (i < n && j < n) v = it.A[i + (int64_t)j * it.lda];
        S[i * pitch + j] = v;
        V[i * pitch + j] = make_double2(i == j ? 1.0 : 0.0, 0.0);
        const double a2 = v.x * v.x + v.y * v.y;
        if (i == j)
            dg += a2;
        else
            of += a2;
    }
    const double dg2 = b_block_sum(dg, sh), of2 = b_block_sum(of, sh);
    if (tid == 0) {
        s_red[0] = dg2 + of2;
        s_red[1] = of2;
    }
    __syncthreads();
    const double fro2 = s_red[0];
    const bool finite_in = isfinite(fro2);
    const double fro = sqrt(fro2);
    if (np > n && tid == 0) S[n * pitch + n] = make_double2(2.0 * fro + 1.0, 0.0);
    __syncthreads();
    const double tol = 1e-14;
    bool done = !finite_in || s_red[1] == 0.0;
    double prev_off = -1.0;
    const int npairs = np / 2;
    for (int sweep = 0; sweep < 40 && !done; ++sweep) {
        for (int round = 0; round < np - 1; ++round) {
            if (tid < npairs) {
                int p, q;
                rr_pair(np, round, tid, p, q);
                const cd beta = S[p * pitch + q];
                const double al = S[p * pitch + p].x, ga = S[q * pitch + q].x;
                const double b2 = beta.x * beta.x + beta.y * beta.y;
                double c = 1.0;
                cd s = make_double2(0.0, 0.0);
                if (b2 > 1e-300 && b2 > 1e-36 * (fabs(al * ga) + 1e-300)) {
                    const double d = 0.5 * (ga - al);
                    const double den = fabs(d) + sqrt(d * d + b2);
                    const double u = 1.0 / den;
                    c = 1.0 / sqrt(1.0 + b2 * u * u);
                    const double f = (d >= 0.0 ? c : -c) * u;
                    s = make_double2(f * beta.x, f * beta.y);
                }
                s_c[tid] = c;
                s_s[tid] = s;
                s_p[tid] = p;
                s_q[tid] = q;
            }
            __syncthreads();
            // S <- J^H S J and V <- V J in ONE phase: the 2 x 2 block of S at (rows of pair k) x (columns of pair l) is
            // transformed by both rotations in registers -- the blocks are disjoint, so the update is in place with a single
            // barrier (the former rows-then-columns form needed two phases and touched every element twice); the rows of V
            // take the column rotation of pair l in the same phase.
            for (int t = tid; t < npairs * npairs + np * npairs; t += nthr) {
                if (t < npairs * npairs) {
                    const int k = t / npairs, l = t - k * npairs;
                    const int p = s_p[k], q = s_q[k], r = s_p[l], s2 = s_q[l];
                    const double ck = s_c[k], cl = s_c[l];
                    const cd sk = s_s[k], sl = s_s[l];
                    const cd a = S[p * pitch + r], bq = S[p * pitch + s2], c2 = S[q * pitch + r], d = S[q * pitch + s2];
                    // rows: (row_p, row_q) <- (c row_p - s row_q, conj(s) row_p + c row_q)
                    const cd a1 = make_double2(ck * a.x - (sk.x * c2.x - sk.y * c2.y), ck * a.y - (sk.x * c2.y + sk.y * c2.x));
                    const cd b1 = make_double2(ck * bq.x - (sk.x * d.x - sk.y * d.y), ck * bq.y - (sk.x * d.y + sk.y * d.x));
                    const cd c1 = make_double2(sk.x * a.x + sk.y * a.y + ck * c2.x, sk.x * a.y - sk.y * a.x + ck * c2.y);
                    const cd d1 = make_double2(sk.x * bq.x + sk.y * bq.y + ck * d.x, sk.x * bq.y - sk.y * bq.x + ck * d.y);
                    // columns: (col_r, col_s) <- (c col_r - conj(s) col_s, s col_r + c col_s)
                    S[p * pitch + r] = make_double2(cl * a1.x - (sl.x * b1.x + sl.y * b1.y), cl * a1.y - (sl.x * b1.y - sl.y * b1.x));
                    S[p * pitch + s2] = make_double2(sl.x * a1.x - sl.y * a1.y + cl * b1.x, sl.x * a1.y + sl.y * a1.x + cl * b1.y);
                    S[q * pitch + r] = make_double2(cl * c1.x - (sl.x * d1.x + sl.y * d1.y), cl * c1.y - (sl.x * d1.y - sl.y * d1.x));
                    S[q * pitch + s2] = make_double2(sl.x * c1.x - sl.y * c1.y + cl * d1.x, sl.x * c1.y + sl.y * c1.x + cl * d1.y);
                } else {
                    const int u = t - npairs * npairs;
                    const int l = u / np, i = u - l * np;
                    const int r = s_p[l], s2 = s_q[l];
                    const double cl = s_c[l];
                    const cd sl = s_s[l];
                    const cd xp = V[i * pitch + r], xq = V[i * pitch + s2];
                    V[i * pitch + r] = make_double2(cl * xp.x - (sl.x * xq.x + sl.y * xq.y), cl * xp.y - (sl.x * xq.y - sl.y * xq.x));
                    V[i * pitch + s2] = make_double2(sl.x * xp.x - sl.y * xp.y + cl * xq.x, sl.x * xp.y + sl.y * xp.x + cl * xq.y);
                }
            }
            __syncthreads();
        }
        double o = 0.0;
        for (int t = tid; t < np * np; t += nthr) {
            const int j = t / np, i = t - j * np;
            if (i != j) {
                const cd v = S[i * pitch + j];
                o += v.x * v.x + v.y * v.y;
            }
        }
        const double o2 = b_block_sum(o, sh);
        if (tid == 0) s_red[1] = o2;
        __syncthreads();
        const double off = sqrt(s_red[1]);
        if (!isfinite(off)) break;
        if (off <= tol * fro) done = true;
        if (!done && prev_off >= 0.0 && off > 0.5 * prev_off && off <= 1e-12 * fro) done = true;
        prev_off = off;
        __syncthreads();
    }
    // eigenvalues ascending (stable rank sort), eigenvector columns in that order; the pad sorts last and is dropped
    double* res = it.res;
    if (tid < np) {
        const double di = S[tid * pitch + tid].x;
        int rank = 0;
        for (int j = 0; j < np; ++j) {
            const double dj = S[j * pitch + j].x;
            if (dj < di || (dj == di && j < tid)) rank += 1;
        }
        if (rank < n) {
            res[rank] = di;
            if (it.ev) it.ev[rank] = di;
            for (int i = 0; i < n; ++i) it.B[i + (int64_t)rank * it.ldb] = V[i * pitch + tid];
        }
    }
    if (tid == 0) {
        res[n] = (done && finite_in) ? 1.0 : 0.0;
        res[n + 1] = finite_in ? 0.0 : 1.0;
    }
}

// Y = D X per item: banded real D (n_p x n_p, half bandwidth bw), X and Y n_p x nb complex (k_apply_D of dense_kernels.hip)
struct ApplyDItem {
    int n_p, nb, bw;
    const double* D;
    const cd* X;
    cd* Y;
};
__global__ __launch_bounds__(256) void k_b_apply_D(const ApplyDItem* __restrict__ items) {
    const ApplyDItem it = items[blockIdx.y];
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)it.n_p * it.nb) return;
    const int c = (int)(idx / it.n_p), i = (int)(idx - (int64_t)c * it.n_p);
    const int j0 = max(0, i - it.bw), j1 = min(it.n_p - 1, i + it.bw);
    double sr = 0.0, si = 0.0;
    for (int j = j0; j <= j1; ++j) {
        const double d = it.D[i + (int64_t)j * it.n_p];
        const cd x = it.X[j + (int64_t)c * it.n_p];
        sr += d * x.x;
        si += d * x.y;
    }
    it.Y[idx] = make_double2(sr, si);
}


// ---- fused ortho!(X) / ortho!(X, Y) of a small block: the whole adaptive loop inside one kernel ----
// lobpcg_hyper_impl.jl:216-323 for B = I: Cholesky-QR passes until eps cond(R)^2 < tol (safe_cholesky's shift-and-retry
// included, :190-210), projection against Y until ||Y'X||_F < tol or the growth factor says it is not needed.  One
// workgroup per block; X stays in global memory (L2-resident: 1350 x 6 complex = 130 KB), the m x m algebra (m <= 8) runs
// in the registers of the first wave, fully unrolled on an identity-padded 8 x 8 matrix.  The host-driven loop of
// lobpcg.cpp (ortho_XY / ortho_X) spends 2-4 host synchronisations per call on these few kiloflops -- in the batched
// multi-k driver 2-4 scheduling rounds per LOBPCG iteration.  Rare branches that need the host (drop_small!'s random
// columns, the SVD fallback) are NOT taken here: the kernel reports status 1 and the driver restarts on its general path.
#define OR_M 8
#define OR_NY 16
#define OR_T 512
struct OrthoItem {
    int64_t n, ldx, ldy;
    int m, ny;
    cd* X;
    const cd* Y;
    const double* norms;   // column norms of X from its producer, or null (computed here); used with Y only
    double tol;
    double* res;           // {status, ortho!(X, Y) rounds, Cholesky count of the last ortho!(X), growth factor}
};

// s_out[a + na * j] = sum_r conj(A[r, a]) B[r, j] (block-wide, fixed summation order)
__device__ __forceinline__ void o_block_dots(const cd* __restrict__ A, int64_t lda, int na, const cd* __restrict__ B,
                                             int64_t ldb, int nb, int64_t n, cd* s_out, cd* s_part) {
    const int tid = threadIdx.x;
    const int P = na * nb;                 // <= 128
    const int G = OR_T / P;                // row groups per entry (>= 4); consecutive threads walk consecutive rows
    const int p = tid / G, g = tid - p * G;
    double sr = 0.0, si = 0.0;
    if (p < P) {
        const int a = p % na, j = p / na;
        const cd* ap = A + (int64_t)a * lda;
        const cd* bp = B + (int64_t)j * ldb;
#pragma unroll 8
        for (int64_t r = g; r < n; r += G) {     // (unrolled: eight independent load pairs in flight per thread)
            const cd u = ap[r], v = bp[r];
            sr += u.x * v.x + u.y * v.y;
            si += u.x * v.y - u.y * v.x;
        }
    }
    s_part[tid] = make_double2(sr, si);
    __syncthreads();
    if (tid < P) {
        double tr = 0.0, ti = 0.0;
        for (int q = 0; q < G; ++q) {
            const cd v = s_part[tid * G + q];
            tr += v.x;
            ti += v.y;
        }
        s_out[tid] = make_double2(tr, ti);
    }
    __syncthreads();
}

// block-wide sums of OR_M doubles per thread -> s_out[0 .. OR_M)
__device__ __forceinline__ void o_block_sum8(const double (&v)[OR_M], double* s_out, double* s_red /* [OR_T / 64][OR_M] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < OR_M; ++j) {
        const double s = b_wave_sum(v[j]);
        if (lane == 0) s_red[w * OR_M + j] = s;
    }
    __syncthreads();
    if (threadIdx.x < OR_M) {
        double t = 0.0;
        for (int q = 0; q < OR_T / 64; ++q) t += s_red[q * OR_M + threadIdx.x];
        s_out[threadIdx.x] = t;
    }
    __syncthreads();
}

// safe_cholesky + inverse + normest of the m x m Gram matrix s_O by the FIRST WAVE, lane-parallel on LDS copies (the other
// waves wait at the caller's barrier).  Synchronisation inside the wave: its LDS operations execute in program order;
// the fences / wave barriers below only keep the compiler from moving accesses across the steps.
// s_inv: inverse of the upper factor (zeros below the diagonal, identity outside m x m);
// s_stat = {nchol (10000: gave up), normest R, normest inv R, non-finite Gram matrix}
__device__ __forceinline__ void o_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double o_wave_sum_all(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double o_wave_max_all(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
// value of `v` in lane `src` (a compile-time lane): two v_readlane_b32
__device__ __forceinline__ double o_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// Lane c < 8 keeps column c of the identity-padded 8 x 8 matrix in REGISTERS; the right-looking Cholesky steps and the
// column-wise inverse fetch the entries of other columns with v_readlane (wave-uniform values): no LDS round trips, no
// wave synchronisation -- the LDS form of this routine was 10 000 cycles per call (clock64), a quarter of the kernel.
__device__ __forceinline__ void o_chol_inv(const cd* s_O, int m, cd* s_R, cd* s_inv, double* s_stat) {
    (void)s_R;
    const double EPSD = 2.220446049250313e-16;
    const int lane = threadIdx.x & 63, col = lane & (OR_M - 1);
    int nchol = 0;
    double alpha = 100.0, shift = 0.0, nR = 0.0, nI = 0.0;
    bool nonfinite = false;
    double xr[OR_M], xi[OR_M];
    for (;;) {
        if (nchol >= 5) {
            nchol = 10000;
            break;
        }
        nchol += 1;
        double Sr[OR_M], Si[OR_M];
#pragma unroll
        for (int i = 0; i < OR_M; ++i) {
            const cd v = s_O[i + OR_M * col];
            Sr[i] = v.x + ((i == col && i < m) ? shift : 0.0);
            Si[i] = (i == col) ? 0.0 : v.y;
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < OR_M; ++j) {
            const double d = o_bcast(Sr[j], j);                   // the pivot S[j][j] (uniform)
            if (!(d > 0.0) || !isfinite(d)) ok = false;
            const double piv = ok ? sqrt(d) : 1.0, inv = 1.0 / piv;
            if (col > j) {
                Sr[j] *= inv;
                Si[j] *= inv;
            } else if (col == j) {
                Sr[j] = piv;
                Si[j] = 0.0;
            }
#pragma unroll
            for (int i = j + 1; i < OR_M; ++i) {                   // S[i][col] -= conj(S[j][i]) S[j][col], col >= i
                const double ur = o_bcast(Sr[j], i), ui = o_bcast(Si[j], i);
                if (col >= i) {
                    Sr[i] -= ur * Sr[j] + ui * Si[j];
                    Si[i] = (col == i) ? 0.0 : Si[i] - (ur * Si[j] - ui * Sr[j]);
                }
            }
        }
        // column `col` of the inverse: x_col = 1 / R_cc, x_i = -(sum_{k > i} R_ik x_k) / R_ii for i < col, zero below
#pragma unroll
        for (int i = OR_M - 1; i >= 0; --i) {
            const double rii = o_bcast(Sr[i], i);
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = i + 1; k < OR_M; ++k) {
                const double rr = o_bcast(Sr[i], k), ri = o_bcast(Si[i], k);
                sr += rr * xr[k] - ri * xi[k];
                si += rr * xi[k] + ri * xr[k];
            }
            xr[i] = i == col ? 1.0 / rii : (i < col ? -sr / rii : 0.0);
            xi[i] = i < col ? -si / rii : 0.0;
        }
        // normest = max |diag| + ||strict upper||_F of both factors, finiteness (columns of the m x m part only, once each)
        double mxR = 0.0, offR = 0.0, mxI = 0.0, offI = 0.0, bflag = 0.0;
        if (lane < m) {
#pragma unroll
            for (int i = 0; i < OR_M; ++i) {
                if (i <= col) {
                    if (!(isfinite(Sr[i]) && isfinite(Si[i]) && isfinite(xr[i]) && isfinite(xi[i]))) bflag = 1.0;
                    if (i == col) {
                        mxR = fabs(Sr[i]);
                        mxI = fabs(xr[i]);
                    } else {
                        offR += Sr[i] * Sr[i] + Si[i] * Si[i];
                        offI += xr[i] * xr[i] + xi[i] * xi[i];
                    }
                }
            }
        }
        const bool bad = o_wave_max_all(bflag) != 0.0;
        nR = o_wave_max_all(mxR) + sqrt(o_wave_sum_all(offR));
        nI = o_wave_max_all(mxI) + sqrt(o_wave_sum_all(offI));
        if (ok && !bad) break;
        // O += alpha eps ||O||_F I  (the Frobenius norm of the Hermitian matrix as it stands, earlier shifts included)
        double f = 0.0;
        {
            const int i = lane % OR_M, j = lane / OR_M;
            if (i < m && j < m) {
                cd v = i <= j ? s_O[i + OR_M * j] : s_O[j + OR_M * i];
                if (i == j) v = make_double2(v.x + shift, 0.0);
                f = v.x * v.x + v.y * v.y;
            }
        }
        const double f2 = o_wave_sum_all(f);
        if (!isfinite(f2)) {
            nonfinite = true;
            break;
        }
        shift += alpha * EPSD * sqrt(f2);
        alpha *= 10.0;
    }
    if (lane < OR_M) {
#pragma unroll
        for (int i = 0; i < OR_M; ++i) s_inv[i + OR_M * col] = make_double2(xr[i], xi[i]);
    }
    if (lane == 0) {
        s_stat[0] = (double)nchol;
        s_stat[1] = nR;
        s_stat[2] = nI;
        s_stat[3] = nonfinite ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(OR_T) void k_b_ortho(const OrthoItem* __restrict__ items) {
    const OrthoItem it = items[blockIdx.x];
    const double EPSD = 2.220446049250313e-16;
    const int tid = threadIdx.x, m = it.m, ny = it.ny;
    const int64_t n = it.n;
    __shared__ cd s_part[OR_T];
    __shared__ cd s_byx[OR_NY * OR_M];
    __shared__ cd s_O[OR_M * OR_M];
    __shared__ cd s_inv[OR_M * OR_M];
    __shared__ cd s_R[OR_M * OR_M];
    __shared__ double s_red[(OR_T / 64) * OR_M];
    __shared__ double s_nrm[OR_M];
    __shared__ double s_stat[4];
    double status = 0.0, growth_last = 1.0;
    int nchol_last = 0, rounds = 0;

    // ortho!(X): Cholesky-QR passes; returns false when the caller has to stop (status set)
    auto ortho_x = [&]() __attribute__((always_inline)) -> bool {
        double growth = 1.0;
        int nchol_total = 0;
        for (int pass = 0;; ++pass) {
            if (pass >= 30) {
                status = 1.0;
                return false;
            }
            o_block_dots(it.X, it.ldx, m, it.X, it.ldx, m, n, s_byx, s_part);      // s_byx[a + m * j] = <x_a, x_j>
            if (tid < OR_M * OR_M) {
                const int i = tid % OR_M, j = tid / OR_M;
                cd v = make_double2(i == j ? 1.0 : 0.0, 0.0);                        // identity padding
                if (i < m && j < m) {
                    // hermitised from the upper triangle (ew_hermitize_upper): real diagonal, lower = conj(upper)
                    const cd u = i <= j ? s_byx[i + m * j] : s_byx[j + m * i];
                    v = i == j ? make_double2(u.x, 0.0) : (i < j ? u : make_double2(u.x, -u.y));
                }
                s_O[i + OR_M * j] = v;
            }
            __syncthreads();
            if (tid < 64) o_chol_inv(s_O, m, s_R, s_inv, s_stat);
            __syncthreads();
            if (s_stat[3] != 0.0) {
                status = 2.0;
                return false;
            }
            const int nchol = (int)s_stat[0];
            if (nchol > 10) {                  // "Ortho(X) is failing badly, falling back to SVD": host path
                status = 1.0;
                return false;
            }
            nchol_total += nchol;
            // X <- X inv(R), row by row in registers
            for (int64_t r = tid; r < n; r += OR_T) {
                cd x[OR_M];
#pragma unroll
                for (int j = 0; j < OR_M; ++j) x[j] = j < m ? it.X[r + (int64_t)j * it.ldx] : make_double2(0.0, 0.0);
#pragma unroll
                for (int j = OR_M - 1; j >= 0; --j) {      // last column first: column j needs the old columns k <= j
                    if (j < m) {
                        double orr = 0.0, oi = 0.0;
#pragma unroll
                        for (int k = 0; k <= j; ++k) {
                            const cd u = s_inv[k + OR_M * j];
                            orr += x[k].x * u.x - x[k].y * u.y;
                            oi += x[k].x * u.y + x[k].y * u.x;
                        }
                        it.X[r + (int64_t)j * it.ldx] = make_double2(orr, oi);
                    }
                }
            }
            __syncthreads();                   // (global writes of this block are visible to it after the barrier)
            const double nR = s_stat[1], nI = s_stat[2];
            growth *= nI;
            const double condR = nR * nI;
            __syncthreads();                   // s_stat / s_inv are rewritten by the next pass
            if (nchol == 1 && EPSD * condR * condR < it.tol) break;
        }
        growth_last = growth;
        nchol_last = nchol_total;
        return true;
    };

    if (ny == 0) {
        ortho_x();
    } else {
        // X ./= norms is folded into the first projection round (scale factors sc[j])
        if (it.norms) {
            if (tid < OR_M) s_nrm[tid] = tid < m ? it.norms[tid] : 1.0;
            __syncthreads();
        } else {
            double acc[OR_M];
#pragma unroll
            for (int j = 0; j < OR_M; ++j) acc[j] = 0.0;
            for (int64_t r = tid; r < n; r += OR_T)
#pragma unroll
                for (int j = 0; j < OR_M; ++j)
                    if (j < m) {
                        const cd v = it.X[r + (int64_t)j * it.ldx];
                        acc[j] += v.x * v.x + v.y * v.y;
                    }
            o_block_sum8(acc, s_nrm, s_red);
            if (tid < OR_M) s_nrm[tid] = tid < m ? sqrt(s_nrm[tid]) : 1.0;
            __syncthreads();
        }
        double sc[OR_M];
#pragma unroll
        for (int j = 0; j < OR_M; ++j) sc[j] = 1.0 / s_nrm[j];
        __syncthreads();
        for (int niter = 1;; ++niter) {
            rounds = niter;
            // BYX = Y' X ; X -= Y BYX ; column norms of the new X and ||BYX||_F^2 on the way
            o_block_dots(it.Y, it.ldy, ny, it.X, it.ldx, m, n, s_byx, s_part);      // s_byx[a + ny * j]
            if (niter == 1) {
                if (tid < ny * m) {
                    const int j = tid / ny;
                    double f = 1.0;
#pragma unroll
                    for (int q = 0; q < OR_M; ++q)
                        if (q == j) f = sc[q];
                    s_byx[tid].x *= f;
                    s_byx[tid].y *= f;
                }
                __syncthreads();
            }
            double acc[OR_M];
#pragma unroll
            for (int j = 0; j < OR_M; ++j) acc[j] = 0.0;
            for (int64_t r = tid; r < n; r += OR_T) {
                cd x[OR_M];
#pragma unroll
                for (int j = 0; j < OR_M; ++j) {
                    x[j] = j < m ? it.X[r + (int64_t)j * it.ldx] : make_double2(0.0, 0.0);
                    if (niter == 1) {
                        x[j].x *= sc[j];
                        x[j].y *= sc[j];
                    }
                }
#pragma unroll 2
                for (int a = 0; a < ny; ++a) {
                    const cd y = it.Y[r + (int64_t)a * it.ldy];
#pragma unroll
                    for (int j = 0; j < OR_M; ++j) {
                        const cd bq = s_byx[a + ny * (j < m ? j : 0)];
                        if (j < m) {
                            x[j].x -= y.x * bq.x - y.y * bq.y;
                            x[j].y -= y.x * bq.y + y.y * bq.x;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < OR_M; ++j)
                    if (j < m) {
                        it.X[r + (int64_t)j * it.ldx] = x[j];
                        acc[j] += x[j].x * x[j].x + x[j].y * x[j].y;
                    }
            }
            o_block_sum8(acc, s_nrm, s_red);        // (ends with a barrier: the new X is visible to the whole block)
            double byx2 = 0.0;
#pragma nounroll
            for (int t = 0; t < ny * m; ++t) {
                const cd v = s_byx[t];
                byx2 += v.x * v.x + v.y * v.y;
            }
            bool nonfinite = false, drop = false;
#pragma unroll
            for (int j = 0; j < OR_M; ++j)
                if (j < m) {
                    const double nj = sqrt(s_nrm[j]);
                    if (!isfinite(nj)) nonfinite = true;
                    if (nj <= it.tol) drop = true;
                }
            __syncthreads();
            if (nonfinite) {
                status = 2.0;
                break;
            }
            if (drop) {                         // drop_small!: a random column from the host's generator -- general path
                status = 1.0;
                break;
            }
            if (sqrt(byx2) < it.tol && niter > 1) break;
            if (!ortho_x()) break;
            if (growth_last * EPSD < it.tol) break;
            if (niter > 10) {                   // "Ortho(X, Y) is failing badly, falling back to SVD"
                status = 1.0;
                break;
            }
        }
    }
    if (tid == 0) {
        it.res[0] = status;
        it.res[1] = (double)rounds;
        it.res[2] = (double)nchol_last;
        it.res[3] = growth_last;
    }
}


// ---- the same loops with the block in REGISTERS (n <= RPT * OR_T rows): every thread keeps RPT rows of X for the whole
// kernel; the tall products are per-thread partial sums over those rows, reduced by a reduce-scatter butterfly inside each
// wave (63 lane exchanges for 64 values instead of 6 per value) and a fixed-order sum over the waves.  The streaming kernel
// above walks X through L2 in every phase: its dependent-load loops made it ~180 us per call on the 1350 x 6 blocks of
// the Al workload (rocprofv3, round 6), i.e. half of a scheduling round.
#define OR_V 32          // values per wave reduction (a 2 x 8 complex tile)
#define OR_SLOTS 8       // tiles reduced between two barriers (ny <= 16 rows of BYX = 8 tiles; 4 tiles of the Gram matrix)
// the value lane `lane` receives from its partner of the butterfly step H.  Partners: lane ^ 32 / ^ 16 (v_permlane*_swap),
// then the DPP mirrors inside a row of 16 -- 15 - i, 7 - i, 3 - i, i ^ 1: each differs from the lane in bit H (and below),
// which is all a reduce-scatter step needs (the masks 16, 15, 7, 3, 1 are linearly independent: after the five steps a
// lane has summed over its whole half-wave).  No LDS traffic (ds_bpermute) in the reduction.
template <int H>
__device__ __forceinline__ double o_recv(double send, int lane) {
    int lo = __double2loint(send), hi = __double2hiint(send);
    if constexpr (H == 1) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);      // quad_perm:[1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
    } else if constexpr (H == 2) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x1B, 0xF, 0xF, false);      // quad_perm:[3,2,1,0]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x1B, 0xF, 0xF, false);
    } else if constexpr (H == 4) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false);     // row_half_mirror
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false);
    } else if constexpr (H == 8) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false);     // row_mirror
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false);
    } else if constexpr (H == 16) {
        // odd rows of the first operand <-> even rows of the second: lanes 16-31 get lanes 0-15 in [0], lanes 0-15 get 16-31 in [1]
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        lo = (lane & 16) ? (int)a[0] : (int)a[1];
        hi = (lane & 16) ? (int)c[0] : (int)c[1];
    } else {
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto c = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        lo = (lane & 32) ? (int)a[0] : (int)a[1];
        hi = (lane & 32) ? (int)c[0] : (int)c[1];
    }
    return __hiloint2double(hi, lo);
}
template <int L>
__device__ __forceinline__ void o_rs_step(double (&v)[OR_V], int lane) {
    constexpr int H = L / 2;
    const bool up = (lane & H) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double send = up ? v[i] : v[i + H];
        const double keep = up ? v[i + H] : v[i];
        v[i] = keep + o_recv<H>(send, lane);
    }
}
// sums over the wave of OR_V values per lane: reduce-scatter butterfly (31 lane exchanges), the two half-waves added, the
// result of value l left by lane l < 32 in s_w[(slot * NW + wave) * OR_V + l].  No barrier: a caller reduces several tiles
// back to back and calls o_finish once.
template <int NW>
__device__ __forceinline__ void o_wave_reduce_store(double (&v)[OR_V], int slot, double* s_w) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    o_rs_step<32>(v, lane);
    o_rs_step<16>(v, lane);
    o_rs_step<8>(v, lane);
    o_rs_step<4>(v, lane);
    o_rs_step<2>(v, lane);
    const double t0 = v[0] + o_recv<32>(v[0], lane);
    if (lane < OR_V) s_w[(slot * NW + w) * OR_V + lane] = t0;
}
// dest[slot * OR_V + l] = sum over the waves (in index order: deterministic) of the tiles stored since the last call
template <int NW>
__device__ __forceinline__ void o_finish(int nslots, double* dest, const double* s_w) {
    __syncthreads();
    if ((int)threadIdx.x < nslots * OR_V) {
        const int slot = threadIdx.x / OR_V, idx = threadIdx.x - slot * OR_V;
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) t += s_w[(slot * NW + q) * OR_V + idx];
        dest[threadIdx.x] = t;
    }
    __syncthreads();
}

template <int RPT>
__global__ __launch_bounds__(OR_T) void k_b_ortho_reg(const OrthoItem* __restrict__ items) {
    const OrthoItem it = items[blockIdx.x];
    const double EPSD = 2.220446049250313e-16;
    constexpr int NW = OR_T / 64;
    const int tid = threadIdx.x, m = it.m, ny = it.ny;
    const int64_t n = it.n;
    __shared__ double s_w[OR_SLOTS * NW * OR_V];
    __shared__ double s_t[OR_V];                    // one reduced tile: (row-in-chunk * OR_M + column) * 2 + {re, im}
    __shared__ cd s_byx[OR_NY * OR_M];              // BYX[a][j] at a * OR_M + j
    __shared__ cd s_O[OR_M * OR_M];
    __shared__ cd s_inv[OR_M * OR_M];
    __shared__ cd s_R[OR_M * OR_M];
    __shared__ double s_nrm[OR_M];
    __shared__ double s_stat[4];
    double status = 0.0, growth_last = 1.0;
    int nchol_last = 0, rounds = 0;
    // shader clocks per phase (res[4 ..]): {total, load, BYX, update + norms, Gram, Cholesky, X inv(R), store}
    long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tc = clock64();
    const long long tc0 = tc;
    auto lap = [&](int k) {
        const long long now = clock64();
        clk[k] += now - tc;
        tc = now;
    };
    cd x[RPT][OR_M];
    bool live[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int64_t r = tid + (int64_t)q * OR_T;
        live[q] = r < n;
#pragma unroll
        for (int j = 0; j < OR_M; ++j)
            x[q][j] = (live[q] && j < m) ? it.X[r + (int64_t)j * it.ldx] : make_double2(0.0, 0.0);
    }

    lap(1);
    auto ortho_x = [&]() __attribute__((always_inline)) -> bool {
        double growth = 1.0;
        int nchol_total = 0;
        for (int pass = 0;; ++pass) {
            if (pass >= 30) {
                status = 1.0;
                return false;
            }
            // Gram matrix, two columns per tile: tile entry (jl, i) = <x_i, x_{2 c + jl}>; all tiles between two barriers.
            // The layout of a reduced tile IS the layout of s_O (column 2 c + jl, row i, re / im): finish writes it in place.
            // (unrolled so that every index into x[][] is a compile-time constant -- a runtime column index, even a select
            //  between entries, makes the compiler keep x[][] in scratch memory; the scheduling barriers keep the four tiles'
            //  accumulators from being live together, which spilled hundreds of registers)
#pragma unroll
            for (int c = 0; c < OR_M / 2; ++c) {
                if (2 * c < m) {
                    double v[OR_V];
#pragma unroll
                    for (int t = 0; t < OR_V; ++t) v[t] = 0.0;
#pragma unroll
                    for (int q = 0; q < RPT; ++q)
#pragma unroll
                        for (int jl = 0; jl < 2; ++jl) {
                            const cd xj = x[q][2 * c + jl];
#pragma unroll
                            for (int i = 0; i < OR_M; ++i) {
                                v[(jl * OR_M + i) * 2] += x[q][i].x * xj.x + x[q][i].y * xj.y;
                                v[(jl * OR_M + i) * 2 + 1] += x[q][i].x * xj.y - x[q][i].y * xj.x;
                            }
                        }
                    o_wave_reduce_store<NW>(v, c, s_w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            o_finish<NW>((m + 1) / 2, reinterpret_cast<double*>(s_O), s_w);
            // hermitise from the upper triangle, identity padding
            if (tid < OR_M * OR_M) {
                const int i = tid % OR_M, j = tid / OR_M;
                cd vv = make_double2(i == j ? 1.0 : 0.0, 0.0);
                if (i < m && j < m) {
                    const cd u = i <= j ? s_O[i + OR_M * j] : s_O[j + OR_M * i];
                    vv = i == j ? make_double2(u.x, 0.0) : (i < j ? u : make_double2(u.x, -u.y));
                }
                s_O[i + OR_M * j] = vv;
            }
            __syncthreads();
            lap(4);
            if (tid < 64) o_chol_inv(s_O, m, s_R, s_inv, s_stat);
            __syncthreads();
            lap(5);
            if (s_stat[3] != 0.0) {
                status = 2.0;
                return false;
            }
            const int nchol = (int)s_stat[0];
            if (nchol > 10) {
                status = 1.0;
                return false;
            }
            nchol_total += nchol;
            // X <- X inv(R) in place, last column first (column j needs the OLD columns k <= j only); one column of
            // inv(R) in registers at a time
#pragma unroll
            for (int j = OR_M - 1; j >= 0; --j) {
                cd u[OR_M];
#pragma unroll
                for (int k = 0; k <= j; ++k) u[k] = s_inv[k + OR_M * j];
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    double orr = 0.0, oi = 0.0;
#pragma unroll
                    for (int k = 0; k <= j; ++k) {
                        orr += x[q][k].x * u[k].x - x[q][k].y * u[k].y;
                        oi += x[q][k].x * u[k].y + x[q][k].y * u[k].x;
                    }
                    x[q][j] = j < m ? make_double2(orr, oi) : make_double2(0.0, 0.0);
                }
            }
            const double nR = s_stat[1], nI = s_stat[2];
            growth *= nI;
            const double condR = nR * nI;
            __syncthreads();
            lap(6);
            if (nchol == 1 && EPSD * condR * condR < it.tol) break;
        }
        growth_last = growth;
        nchol_last = nchol_total;
        return true;
    };

    // (ONE call site of ortho_x below: inlined, so that the rows x[][] stay in registers -- with two call sites the lambda
    //  was outlined and x lived in scratch memory, 390 scratch instructions on the critical path)
    if (ny > 0) {
        // column norms: from the producer or computed here; X ./= norms on the registers
        if (it.norms) {
            if (tid < OR_M) s_nrm[tid] = tid < m ? it.norms[tid] : 1.0;
        } else {
            double v[OR_V];
#pragma unroll
            for (int t = 0; t < OR_V; ++t) v[t] = 0.0;
#pragma unroll
            for (int q = 0; q < RPT; ++q)
#pragma unroll
                for (int j = 0; j < OR_M; ++j) v[j] += x[q][j].x * x[q][j].x + x[q][j].y * x[q][j].y;
            o_wave_reduce_store<NW>(v, 0, s_w);
            o_finish<NW>(1, s_t, s_w);
            if (tid < OR_M) s_nrm[tid] = tid < m ? sqrt(s_t[tid]) : 1.0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OR_M; ++j) {
            const double f = 1.0 / s_nrm[j];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                x[q][j].x *= f;
                x[q][j].y *= f;
            }
        }
        __syncthreads();
    }
    for (int niter = 1;; ++niter) {
        if (ny > 0) {
            rounds = niter;
            // BYX = Y' X, two rows of Y' per tile, all tiles between two barriers; a reduced tile is rows (2 s, 2 s + 1) of s_byx.
            // The loads of FOUR rows of Y' (all RPT rows of the thread) are issued together, then two tiles are reduced from
            // them: a load batch exposes one trip to L2 / the Infinity Cache (~3 000 cycles measured with clock64), and the
            // two-row form exposed one per tile.
#pragma nounroll
            for (int a0 = 0; a0 < ny; a0 += 4) {
                cd yv[RPT][4];
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int64_t r = tid + (int64_t)q * OR_T;
#pragma unroll
                    for (int al = 0; al < 4; ++al) {
                        yv[q][al] = make_double2(0.0, 0.0);
                        if (live[q] && a0 + al < ny) yv[q][al] = it.Y[r + (int64_t)(a0 + al) * it.ldy];
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (a0 + 2 * h < ny) {
                        double v[OR_V];
#pragma unroll
                        for (int t = 0; t < OR_V; ++t) v[t] = 0.0;
#pragma unroll
                        for (int q = 0; q < RPT; ++q)
#pragma unroll
                            for (int al = 0; al < 2; ++al) {
                                const cd y = yv[q][2 * h + al];
#pragma unroll
                                for (int j = 0; j < OR_M; ++j) {
                                    v[(al * OR_M + j) * 2] += y.x * x[q][j].x + y.y * x[q][j].y;
                                    v[(al * OR_M + j) * 2 + 1] += y.x * x[q][j].y - y.y * x[q][j].x;
                                }
                            }
                        o_wave_reduce_store<NW>(v, a0 / 2 + h, s_w);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            o_finish<NW>((ny + 1) / 2, reinterpret_cast<double*>(s_byx), s_w);
            lap(2);
            // X -= Y BYX (four columns of Y per load batch, all rows of the thread), column norms of the result, ||BYX||_F^2
            {
#pragma nounroll
                for (int a0 = 0; a0 < ny; a0 += 4) {
                    cd yv[RPT][4];
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const int64_t r = tid + (int64_t)q * OR_T;
#pragma unroll
                        for (int al = 0; al < 4; ++al) {
                            yv[q][al] = make_double2(0.0, 0.0);
                            if (live[q] && a0 + al < ny) yv[q][al] = it.Y[r + (int64_t)(a0 + al) * it.ldy];
                        }
                    }
#pragma unroll
                    for (int al = 0; al < 4; ++al) {
                        if (a0 + al < ny) {
#pragma unroll
                            for (int j = 0; j < OR_M; ++j) {
                                const cd bq = s_byx[(a0 + al) * OR_M + j];
                                if (j < m) {
#pragma unroll
                                    for (int q = 0; q < RPT; ++q) {
                                        x[q][j].x -= yv[q][al].x * bq.x - yv[q][al].y * bq.y;
                                        x[q][j].y -= yv[q][al].x * bq.y + yv[q][al].y * bq.x;
                                    }
                                }
                            }
                        }
                    }
                }
                double v[OR_V];
#pragma unroll
                for (int t = 0; t < OR_V; ++t) v[t] = 0.0;
#pragma unroll
                for (int q = 0; q < RPT; ++q)
#pragma unroll
                    for (int j = 0; j < OR_M; ++j) v[j] += x[q][j].x * x[q][j].x + x[q][j].y * x[q][j].y;
                o_wave_reduce_store<NW>(v, 0, s_w);
                o_finish<NW>(1, s_t, s_w);
            }
            double byx2 = 0.0;
#pragma nounroll
            for (int a = 0; a < ny; ++a)
#pragma unroll
                for (int j = 0; j < OR_M; ++j)
                    if (j < m) {
                        const cd vv = s_byx[a * OR_M + j];
                        byx2 += vv.x * vv.x + vv.y * vv.y;
                    }
            bool nonfinite = false, drop = false;
#pragma unroll
            for (int j = 0; j < OR_M; ++j)
                if (j < m) {
                    const double nj = sqrt(s_t[j]);
                    if (!isfinite(nj)) nonfinite = true;
                    if (nj <= it.tol) drop = true;
                }
            __syncthreads();
            lap(3);
            if (nonfinite) {
                status = 2.0;
                break;
            }
            if (drop) {
                status = 1.0;
                break;
            }
            if (sqrt(byx2) < it.tol && niter > 1) break;
        }
        if (!ortho_x()) break;
        if (ny == 0) break;
        if (growth_last * EPSD < it.tol) break;
        if (niter > 10) {
            status = 1.0;
            break;
        }
    }
    // the block goes back to memory once (not at all when a rare branch takes over: the driver starts again anyway)
    if (status == 0.0) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int64_t r = tid + (int64_t)q * OR_T;
            if (live[q])
#pragma unroll
                for (int j = 0; j < OR_M; ++j)
                    if (j < m) it.X[r + (int64_t)j * it.ldx] = x[q][j];
        }
    }
    lap(7);
    if (tid == 0) {
        it.res[0] = status;
        it.res[1] = (double)rounds;
        it.res[2] = (double)nchol_last;
        it.res[3] = growth_last;
        clk[0] = clock64() - tc0;
        for (int q = 0; q < 8; ++q) it.res[4 + q] = (double)clk[q];
    }
}

EwItem ew_item(const BOp& o) {
    EwItem e;
    e.n = o.n;
    e.lda = o.lda;
    e.ldb = o.ldb;
    e.ldc = o.ldc;
    e.m = o.m;
    e.mode = o.mode;
    e.flags = o.flags;
    e.i0 = o.i0;
    e.A = o.A;
    e.B = o.B;
    e.W = o.W;
    e.W2 = o.W2;
    e.W3 = o.W3;
    e.C = o.C;
    e.D = o.D;
    e.E = o.E;
    e.F = o.F;
    e.s0 = o.s0;
    e.bytes = o.bytes;
    return e;
}

int set_big_lds(const void* kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

}  // namespace dftk_batch
using namespace dftk_batch;

int batch_exec_group(BatchCtx* ctx, hipStream_t stream, int type, std::vector<BOp*>& ops) {
    const int n_items = (int)ops.size();
    if (n_items == 0) return 0;
    if (n_items > 65535) return 1;
    if (type == BOP_APPLYH) return batch_exec_apply_H(ctx, stream, ops);
    if (type == BOP_DENSITY) return batch_exec_density(ctx, stream, ops);

    if (type == BOP_APPLYD) {
        std::vector<ApplyDItem> items(n_items);
        int64_t mx = 1;
        for (int i = 0; i < n_items; ++i) {
            const dftk_mi_kblock* kb = ops[i]->kb;
            items[i] = ApplyDItem{kb->n_p, ops[i]->m, kb->D_bw, kb->d_D, reinterpret_cast<const cd*>(ops[i]->A),
                                  reinterpret_cast<cd*>(ops[i]->C)};
            mx = std::max<int64_t>(mx, (int64_t)kb->n_p * ops[i]->m);
        }
        const ApplyDItem* d = reinterpret_cast<const ApplyDItem*>(batch_stage(ctx, items.data(), items.size() * sizeof(ApplyDItem)));
        if (!d) return DFTK_MI_EHIP;
        hipLaunchKernelGGL(k_b_apply_D, dim3((unsigned)((mx + 255) / 256), n_items), dim3(256), 0, stream, d);
        HIPCHK(hipGetLastError());
        return 0;
    }

    // ------------------------------------------------------------------ element-wise / column kernels
    if (type == BOP_COLRED || type == BOP_RESIDUAL || type == BOP_TPA || type == BOP_SCALE || type == BOP_COPY ||
        type == BOP_GATHER || type == BOP_FILL0 || type == BOP_SUBID || type == BOP_ADDDIAG || type == BOP_HERMIT ||
        type == BOP_CTRANS) {
        std::vector<EwItem> items(n_items);
        int64_t maxn = 1;
        int maxm = 1;
        size_t maxbytes = 0;
        for (int i = 0; i < n_items; ++i) {
            items[i] = ew_item(*ops[i]);
            maxn = std::max<int64_t>(maxn, ops[i]->n);
            maxm = std::max(maxm, ops[i]->m);
            maxbytes = std::max(maxbytes, ops[i]->bytes);
            if (type == BOP_FILL0 && (ops[i]->bytes % sizeof(double))) return 1;
        }
        const EwItem* d = reinterpret_cast<const EwItem*>(batch_stage(ctx, items.data(), items.size() * sizeof(EwItem)));
        if (!d) return DFTK_MI_EHIP;
        const unsigned rows = (unsigned)((maxn + 255) / 256);
        switch (type) {
            case BOP_COLRED: hipLaunchKernelGGL(k_b_colred, dim3(maxm, 1, n_items), dim3(256), 0, stream, d); break;
            case BOP_RESIDUAL: hipLaunchKernelGGL(k_b_residual, dim3(maxm, 1, n_items), dim3(256), 0, stream, d); break;
            case BOP_TPA: hipLaunchKernelGGL(k_b_tpa, dim3(maxm, 1, n_items), dim3(256), 0, stream, d); break;
            case BOP_SCALE: hipLaunchKernelGGL(k_b_rows, dim3(rows, maxm, n_items), dim3(256), 0, stream, d, 0); break;
            case BOP_COPY: hipLaunchKernelGGL(k_b_rows, dim3(rows, maxm, n_items), dim3(256), 0, stream, d, 1); break;
            case BOP_GATHER: hipLaunchKernelGGL(k_b_rows, dim3(rows, maxm, n_items), dim3(256), 0, stream, d, 2); break;
            case BOP_FILL0: {
                const unsigned g = (unsigned)std::min<size_t>(64, (maxbytes / sizeof(double) + 255) / 256 + 1);
                hipLaunchKernelGGL(k_b_rows, dim3(g, 1, n_items), dim3(256), 0, stream, d, 3);
                break;
            }
            case BOP_SUBID:
                hipLaunchKernelGGL(k_b_small, dim3((maxm + 255) / 256, 1, n_items), dim3(256), 0, stream, d, 0);
                break;
            case BOP_ADDDIAG:
                hipLaunchKernelGGL(k_b_small, dim3((maxm + 255) / 256, 1, n_items), dim3(256), 0, stream, d, 1);
                break;
            case BOP_HERMIT:
                hipLaunchKernelGGL(k_b_small, dim3((unsigned)(((size_t)maxm * maxm + 255) / 256), 1, n_items), dim3(256), 0,
                                   stream, d, 2);
                break;
            default:
                hipLaunchKernelGGL(k_b_small, dim3((unsigned)(((size_t)maxm * maxm + 255) / 256), 1, n_items), dim3(256), 0,
                                   stream, d, 3);
                break;
        }
        HIPCHK(hipGetLastError());
        return 0;
    }

    // ------------------------------------------------------------------ small products
    if (type == BOP_ZGEMM) {
        std::vector<GemmItem> cn, nn;
        int max_m_c = 1, max_n_c = 1, max_n_n = 1, max_k_c = 1;
        int64_t max_m_n = 1;
        for (BOp* o : ops) {
            if (o->flags & DFTK_MI_GEMM_REAL) return 1;
            if (o->gk <= 0) return 1;
            GemmItem g;
            g.m = (int)o->gm;
            g.n = (int)o->gn;
            g.k = (int)o->gk;
            g.flags = o->flags & 3;
            g.lda = o->lda;
            g.ldb = o->ldb;
            g.ldc = o->ldc;
            g.A = reinterpret_cast<const cd*>(o->A);
            g.B = reinterpret_cast<const cd*>(o->B);
            g.C = reinterpret_cast<cd*>(o->C);
            g.alpha = o->alpha;
            g.beta = o->beta;
            if (o->trans == 'C') {
                if (o->gm > 96 || o->gn > 96) return 1;
                cn.push_back(g);
                max_m_c = std::max(max_m_c, g.m);
                max_n_c = std::max(max_n_c, g.n);
                max_k_c = std::max(max_k_c, g.k);
            } else {
                if (o->gk > BN_KMAX || o->gn > 512) return 1;
                nn.push_back(g);
                max_m_n = std::max<int64_t>(max_m_n, g.m);
                max_n_n = std::max(max_n_n, g.n);
            }
        }
        if (!cn.empty()) {
            const GemmItem* d = reinterpret_cast<const GemmItem*>(batch_stage(ctx, cn.data(), cn.size() * sizeof(GemmItem)));
            if (!d) return DFTK_MI_EHIP;
            const int tm = (max_m_c + BG_T - 1) / BG_T, tn = (max_n_c + BG_T - 1) / BG_T;
            // long reductions are cut into chunks of ~4 LDS tiles (256 rows) so that a product is a few microseconds
            // of latency instead of tens; the partial tiles are summed in chunk order by a second small kernel
            int nsplit = std::min(32, (max_k_c + 4 * BG_KC - 1) / (4 * BG_KC));
            if (nsplit <= 1) {
                hipLaunchKernelGGL(k_b_gemm_c, dim3(tm * tn, 1, (unsigned)cn.size()), dim3(256), 0, stream, d, tn, 1, (cd*)nullptr,
                                   0, 0);
            } else {
                const int pm = tm * BG_T, pn = tn * BG_T;
                BatchScratchScope scratch_scope(ctx);
                cd* part = reinterpret_cast<cd*>(batch_scratch(ctx, cn.size() * (size_t)nsplit * pm * pn * sizeof(cd)));
                if (!part) return DFTK_MI_EHIP;
                hipLaunchKernelGGL(k_b_gemm_c, dim3(tm * tn, nsplit, (unsigned)cn.size()), dim3(256), 0, stream, d, tn, nsplit,
                                   part, pm, pn);
                hipLaunchKernelGGL(k_b_gemm_c_reduce, dim3((max_m_c * max_n_c + 255) / 256, (unsigned)cn.size()), dim3(256), 0,
                                   stream, d, nsplit, (const cd*)part, pm, pn);
            }
        }
        if (!nn.empty()) {
            const GemmItem* d = reinterpret_cast<const GemmItem*>(batch_stage(ctx, nn.data(), nn.size() * sizeof(GemmItem)));
            if (!d) return DFTK_MI_EHIP;
            hipLaunchKernelGGL(k_b_gemm_n, dim3((unsigned)((max_m_n + 255) / 256), (max_n_n + BN_C - 1) / BN_C, (unsigned)nn.size()),
                               dim3(256), 0, stream, d);
        }
        HIPCHK(hipGetLastError());
        return 0;
    }

    // ------------------------------------------------------------------ host <-> device copies
    if (type == BOP_H2D) {
        // all payloads packed behind one another in the ring (ONE copy), then scattered to their destinations
        size_t total = 0;
        for (BOp* o : ops) {
            if (o->payload.size() % 4) return 1;
            total += (o->payload.size() + 15) & ~(size_t)15;
        }
        std::vector<char> pack(total);
        std::vector<CopyItem> items(n_items);
        size_t off = 0;
        for (int i = 0; i < n_items; ++i) {
            memcpy(pack.data() + off, ops[i]->payload.data(), ops[i]->payload.size());
            items[i].src = reinterpret_cast<const void*>(off);   // patched below with the device base
            items[i].dst = ops[i]->C;
            items[i].words = (int64_t)(ops[i]->payload.size() / 4);
            off += (ops[i]->payload.size() + 15) & ~(size_t)15;
        }
        const char* dbase = reinterpret_cast<const char*>(batch_stage(ctx, pack.data(), pack.size()));
        if (!dbase) return DFTK_MI_EHIP;
        for (auto& it : items) it.src = dbase + reinterpret_cast<size_t>(it.src);
        const CopyItem* d = reinterpret_cast<const CopyItem*>(batch_stage(ctx, items.data(), items.size() * sizeof(CopyItem)));
        if (!d) return DFTK_MI_EHIP;
        hipLaunchKernelGGL(k_b_copy_words, dim3(4, n_items), dim3(256), 0, stream, d);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (type == BOP_D2H) {
        // validate the WHOLE group before the first slot / fix-up exists: a fall-back to the one-by-one path must not
        // leave fix-ups behind that would overwrite its (correct) host data after the round's synchronisation
        size_t slots = 0;
        for (int i = 0; i < n_items; ++i) {
            if (ops[i]->bytes % 4) return 1;
            slots += (ops[i]->bytes + 255) & ~(size_t)255;
        }
        if (slots > batch_result_room(ctx)) return 1;   // (the one-by-one copies need no slots)
        std::vector<CopyItem> items(n_items);
        for (int i = 0; i < n_items; ++i) {
            void* htwin = nullptr;
            void* dslot = batch_result_slot(ctx, ops[i]->bytes, &htwin);
            if (!dslot) return DFTK_MI_EHIP;
            items[i].src = ops[i]->A;
            items[i].dst = dslot;
            items[i].words = (int64_t)(ops[i]->bytes / 4);
            void* dst_h = ops[i]->host;
            const size_t nb = ops[i]->bytes;
            batch_add_fixup(ctx, [=]() { memcpy(dst_h, htwin, nb); });
        }
        const CopyItem* d = reinterpret_cast<const CopyItem*>(batch_stage(ctx, items.data(), items.size() * sizeof(CopyItem)));
        if (!d) return DFTK_MI_EHIP;
        hipLaunchKernelGGL(k_b_copy_words, dim3(4, n_items), dim3(256), 0, stream, d);
        HIPCHK(hipGetLastError());
        return 0;
    }

    // ------------------------------------------------------------------ fused small-block orthogonalisation
    if (type == BOP_ORTHO) {
        std::vector<OrthoItem> items(n_items);
        for (int i = 0; i < n_items; ++i) {
            BOp* o = ops[i];
            if (o->m < 1 || o->m > OR_M || o->k < 0 || o->k > OR_NY || o->n < 1) {
                dftk_set_error("fused ortho: block shape %lld x %d against %d columns is outside the kernel's range", (long long)o->n,
                               o->m, o->k);
                return DFTK_MI_EINVAL;
            }
            void* htwin = nullptr;
            double* dres = reinterpret_cast<double*>(batch_result_slot(ctx, 16 * sizeof(double), &htwin));   // 4 results + phase clocks
            if (!dres) return DFTK_MI_EHIP;
            items[i] = OrthoItem{o->n, o->ldc, o->lda, o->m, o->k, reinterpret_cast<cd*>(o->C), reinterpret_cast<const cd*>(o->A),
                                 reinterpret_cast<const double*>(o->W), o->s0, dres};
            const double* h = reinterpret_cast<const double*>(htwin);
            batch_add_fixup(ctx, [o, h]() {
                if (o->host) memcpy(o->host, h, 4 * sizeof(double));
                if (o->host2) memcpy(o->host2, h + 4, 8 * sizeof(double));      // DFTK_MI_ORTHO_CLOCKS (dftk_mi_ortho_small)
                o->status = 0;
            });
        }
        const OrthoItem* d = reinterpret_cast<const OrthoItem*>(batch_stage(ctx, items.data(), items.size() * sizeof(OrthoItem)));
        if (!d) return DFTK_MI_EHIP;
        // blocks of up to 4 x 512 rows stay in registers for the whole call; longer ones stream through L2
        int64_t nmax = 0;
        for (BOp* o : ops) nmax = std::max(nmax, o->n);
        static const bool no_reg = getenv("DFTK_MI_ORTHO_STREAM") != nullptr;
        if (no_reg || nmax > 4 * OR_T)
            hipLaunchKernelGGL(k_b_ortho, dim3(n_items), dim3(OR_T), 0, stream, d);
        else if (nmax <= OR_T)
            hipLaunchKernelGGL(k_b_ortho_reg<1>, dim3(n_items), dim3(OR_T), 0, stream, d);
        else if (nmax <= 2 * OR_T)
            hipLaunchKernelGGL(k_b_ortho_reg<2>, dim3(n_items), dim3(OR_T), 0, stream, d);
        else if (nmax <= 3 * OR_T)
            hipLaunchKernelGGL(k_b_ortho_reg<3>, dim3(n_items), dim3(OR_T), 0, stream, d);
        else
            hipLaunchKernelGGL(k_b_ortho_reg<4>, dim3(n_items), dim3(OR_T), 0, stream, d);
        HIPCHK(hipGetLastError());
        return 0;
    }

    // ------------------------------------------------------------------ small factorizations
    if (type == BOP_POTRF || type == BOP_HEEV) {
        int nmax = 1;
        for (BOp* o : ops) {
            if (o->m > 64 || o->m < 1) return 1;
            nmax = std::max(nmax, o->m);
        }
        std::vector<DenseItem> items(n_items);
        for (int i = 0; i < n_items; ++i) {
            BOp* o = ops[i];
            const size_t rb = (type == BOP_POTRF ? 8 : (size_t)o->m + 2) * sizeof(double);
            void* htwin = nullptr;
            double* dres = reinterpret_cast<double*>(batch_result_slot(ctx, rb, &htwin));
            if (!dres) return DFTK_MI_EHIP;
            items[i].n = o->m;
            items[i].lda = o->ldc;
            items[i].ldb = o->ldb;
            items[i].A = reinterpret_cast<cd*>(o->C);
            items[i].B = reinterpret_cast<cd*>(o->D);
            items[i].res = dres;
            items[i].ev = type == BOP_HEEV ? reinterpret_cast<double*>(o->E) : nullptr;
            const double* h = reinterpret_cast<const double*>(htwin);
            if (type == BOP_POTRF) {
                batch_add_fixup(ctx, [o, h]() {
                    double* out = reinterpret_cast<double*>(o->host);
                    if (h[0] != 0.0 || h[3] != 0.0 || h[6] != 0.0) {
                        o->status = DFTK_MI_NUM_CHOLESKY;
                    } else {
                        o->status = 0;
                        out[0] = h[1] + sqrt(h[2]);
                        out[1] = h[4] + sqrt(h[5]);
                    }
                });
            } else {
                batch_add_fixup(ctx, [o, h]() {
                    const int n = o->m;
                    if (h[n + 1] != 0.0) {
                        o->status = DFTK_MI_NUM_NONFINITE;
                    } else if (h[n] == 0.0) {
                        dftk_set_error("dense_heev (batched): Jacobi did not converge (n=%d)", n);
                        o->status = DFTK_MI_NUM_EIGEN;
                    } else {
                        o->status = 0;
                        memcpy(o->host, h, (size_t)n * sizeof(double));
                    }
                });
            }
        }
        const DenseItem* d = reinterpret_cast<const DenseItem*>(batch_stage(ctx, items.data(), items.size() * sizeof(DenseItem)));
        if (!d) return DFTK_MI_EHIP;
        const int pitch = nmax + 1 + (nmax & 1);
        const size_t lds = 2 * (size_t)pitch * pitch * sizeof(cd);
        // small matrices (the k-point configs: M = 6 .. 8, 3M <= 24) get ONE wave per matrix: the dozens of barriers per
        // Jacobi sweep / Cholesky column are then free, and these kernels are pure dependent-latency chains
        // POTRF: ONE wave per matrix up to order 32 (dozens of dependent barriers per column are then free).  HEEV: the fused
        // two-sided update is (n / 2)^2 + n^2 / 2 independent items per round -- one pass of 256 threads up to order 18
        const int threads = type == BOP_POTRF ? (nmax <= 32 ? 64 : 256) : (nmax <= 10 ? 64 : 256);
        if (type == BOP_POTRF) {
            CHK(set_big_lds(reinterpret_cast<const void*>(k_b_potrf), lds));
            hipLaunchKernelGGL(k_b_potrf, dim3(n_items), dim3(threads), lds, stream, d, pitch);
        } else {
            CHK(set_big_lds(reinterpret_cast<const void*>(k_b_heev), lds));
            hipLaunchKernelGGL(k_b_heev, dim3(n_items), dim3(threads), lds, stream, d, pitch);
        }
        HIPCHK(hipGetLastError());
        return 0;
    }
    return 1;
}

