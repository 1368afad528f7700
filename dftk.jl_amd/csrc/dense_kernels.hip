// dense_kernels.hip -- small dense factorizations and the column-wise helpers of LOBPCG (gfx950).
//
//   safe_cholesky: cholesky(O).U, inv(R)      src/eigen/lobpcg_hyper_impl.jl:190-210   (K13)
//   normest                                   :212
//   eigen(Hermitian(XAX))                     :146-171                                 (K11)
//   columnwise_norms / columnwise_dots        src/common/linalg.jl:2-15, src/gpu/linalg.jl:11-23
//   residuals, TPA preconditioner             :443-457, src/eigen/preconditioners.jl:50-77,
//                                             src/gpu/linalg.jl:25-36                  (K16-K19)
//   D * (P' psi)                              src/terms/operators.jl:127               (K8)
#include "common.h"
#include "batch.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cmath>
#include <numeric>
#include <cstring>
#include <vector>

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum for 256 threads; result valid in thread 0
__device__ __forceinline__ double block_sum256(double v, double* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    }
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------- column reductions
// The n_G-sized streaming kernels of one LOBPCG iteration.  They are HBM-bound, and what bounds them is the number of
// bytes in flight: one 16-byte load per thread and 2 x 256 threads per CU (the round 1-3 form) keeps 8 KB per CU in the
// air and reaches 1.2-1.5 TB/s (tools/ew_bench.py).  Here every thread issues EW_UNR independent loads per operand before
// it touches any of them, and a long column gets a workgroup of 1024 threads: 64 KB per operand and workgroup in flight.
// One workgroup per column; deterministic (fixed strides, fixed tree; the tree depends on the workgroup size, which is a
// function of n alone).
#define EW_UNR 4
#define EW_LONG 8192      // rows from which a column gets 1024 threads
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* sh) {   // sh[NT / 64]; result valid in thread 0
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += sh[i];
    }
    __syncthreads();
    return r;
}
// mode 0: out[c] = sqrt(sum |X|^2) ; 1: out[c] = Re sum conj(X) Y ; 2: out[c] = sum w |X|^2 ; 3: sum |X|^2 ; 4: Im sum conj(X) Y
template <int NT>
__global__ __launch_bounds__(NT) void k_col_reduce(int mode, int64_t n, const cd* __restrict__ X, int64_t ldx,
                                                   const cd* __restrict__ Y, int64_t ldy,
                                                   const double* __restrict__ w, double* __restrict__ out) {
    __shared__ double sh[NT / 64];
    const int c = blockIdx.x;
    const cd* x = X + (int64_t)c * ldx;
    const cd* y = Y ? Y + (int64_t)c * ldy : nullptr;
    const bool two = mode == 1 || mode == 4;
    double acc = 0.0;
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)NT * EW_UNR) {
        cd a[EW_UNR], bb[EW_UNR];
        double ww[EW_UNR];
#pragma unroll
        for (int u = 0; u < EW_UNR; ++u) {
            const int64_t i = i0 + (int64_t)u * NT;
            const bool in = i < n;
            a[u] = in ? x[i] : make_double2(0.0, 0.0);
            bb[u] = (in && two) ? y[i] : make_double2(0.0, 0.0);
            ww[u] = (in && mode == 2) ? w[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < EW_UNR; ++u) {
            if (mode == 1)
                acc += a[u].x * bb[u].x + a[u].y * bb[u].y;
            else if (mode == 4)      // Im conj(a) b
                acc += a[u].x * bb[u].y - a[u].y * bb[u].x;
            else if (mode == 2)
                acc += ww[u] * (a[u].x * a[u].x + a[u].y * a[u].y);
            else
                acc += a[u].x * a[u].x + a[u].y * a[u].y;
        }
    }
    const double r = block_sum<NT>(acc, sh);
    if (threadIdx.x == 0) out[c] = (mode == 0) ? sqrt(r) : r;
}

// R = AX - X * lam ; norms[c] = ||R[:,c]|| ; in the same pass over X (optional, kin != null / xx != null):
// mk[c] = sum kin |X|^2 (precondprep! of the TPA preconditioner) and xx[c] = sum |X|^2 (normalisation check)
template <int NT>
__global__ __launch_bounds__(NT) void k_residual(int64_t n, const cd* __restrict__ AX, int64_t lda,
                                                 const cd* __restrict__ X, int64_t ldx,
                                                 const double* __restrict__ lam, cd* __restrict__ R, int64_t ldr,
                                                 double* __restrict__ norms, const double* __restrict__ kin,
                                                 double* __restrict__ mk, double* __restrict__ xx) {
    __shared__ double sh[NT / 64];
    const int c = blockIdx.x;
    const double l = lam[c];
    const cd* ax = AX + (int64_t)c * lda;
    const cd* xc = X + (int64_t)c * ldx;
    cd* rc = R + (int64_t)c * ldr;
    double acc = 0.0, acck = 0.0, accx = 0.0;
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)NT * EW_UNR) {
        cd a[EW_UNR], x[EW_UNR];
        double kk[EW_UNR];
#pragma unroll
        for (int u = 0; u < EW_UNR; ++u) {
            const int64_t i = i0 + (int64_t)u * NT;
            const bool in = i < n;
            a[u] = in ? ax[i] : make_double2(0.0, 0.0);
            x[u] = in ? xc[i] : make_double2(0.0, 0.0);
            kk[u] = (in && kin) ? kin[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < EW_UNR; ++u) {
            const int64_t i = i0 + (int64_t)u * NT;
            const cd r = make_double2(a[u].x - l * x[u].x, a[u].y - l * x[u].y);
            if (i < n) rc[i] = r;
            acc += r.x * r.x + r.y * r.y;
            const double x2 = x[u].x * x[u].x + x[u].y * x[u].y;
            accx += x2;
            acck += kk[u] * x2;
        }
    }
    const double s = block_sum<NT>(acc, sh);
    const double sk = block_sum<NT>(acck, sh);
    const double sx = block_sum<NT>(accx, sh);
    if (threadIdx.x == 0) {
        norms[c] = sqrt(s);
        if (kin) mk[c] = sk;
        if (xx) xx[c] = sx;
    }
}

// ldiv!(precon, R) of the TPA preconditioner, out of place and with the column norms of the result:
//   dst[:,c] = src[:,c] * mean_kin[c] / (mean_kin[c] + kin) ; norms[c] = ||dst[:,c]||     (kin == null: plain copy)
// One workgroup per column (same reduction tree as k_col_reduce).
template <int NT>
__global__ __launch_bounds__(NT) void k_tpa(int64_t n, const cd* __restrict__ src, int64_t lds, cd* __restrict__ dst,
                                            int64_t ldd, const double* __restrict__ kin,
                                            const double* __restrict__ mean_kin, double* __restrict__ norms,
                                            double default_shift) {
    __shared__ double sh[NT / 64];
    const int c = blockIdx.x;
    const double mk = (kin && mean_kin) ? mean_kin[c] : 0.0;
    const cd* sc = src + (int64_t)c * lds;
    cd* dc = dst + (int64_t)c * ldd;
    double acc = 0.0;
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)NT * EW_UNR) {
        cd r[EW_UNR];
        double kk[EW_UNR];
#pragma unroll
        for (int u = 0; u < EW_UNR; ++u) {
            const int64_t i = i0 + (int64_t)u * NT;
            const bool in = i < n;
            r[u] = in ? sc[i] : make_double2(0.0, 0.0);
            kk[u] = (in && kin) ? kin[i] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < EW_UNR; ++u) {
            const int64_t i = i0 + (int64_t)u * NT;
            if (kin) {
                // mean_kin == null: precondprep! has not run yet -> ldiv!(Y, Diagonal(kin .+ default_shift), R)
                const double f = mean_kin ? mk / (mk + kk[u]) : 1.0 / (kk[u] + default_shift);
                r[u].x *= f;
                r[u].y *= f;
            }
            if (i < n) dc[i] = r[u];
            acc += r[u].x * r[u].x + r[u].y * r[u].y;
        }
    }
    const double s = block_sum<NT>(acc, sh);
    if (threadIdx.x == 0) norms[c] = sqrt(s);
}
// launch one workgroup per column with the workgroup size the column length asks for
#define EW_LAUNCH_COLS(kernel, n, m, stream, ...)                                                            \
    do {                                                                                                     \
        if ((n) >= EW_LONG)                                                                                  \
            hipLaunchKernelGGL(kernel<1024>, dim3(m), dim3(1024), 0, stream, __VA_ARGS__);                   \
        else                                                                                                 \
            hipLaunchKernelGGL(kernel<256>, dim3(m), dim3(256), 0, stream, __VA_ARGS__);                     \
    } while (0)

__global__ void k_conj_transpose(int n, const cd* __restrict__ A, int64_t lda, cd* __restrict__ B, int64_t ldb) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    const int j = (int)(idx / n), i = (int)(idx - (int64_t)j * n);
    const cd v = A[j + (int64_t)i * lda];
    B[i + (int64_t)j * ldb] = make_double2(v.x, -v.y);
}
__global__ void k_unary(int mode, double* __restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = mode == 0 ? d[i] * d[i] : sqrt(d[i]);
}

// the row-parallel forms: a workgroup of 256 threads takes EW_UNR * 256 consecutive rows of one column (blockIdx.y)
#define EW_ROWS (256 * EW_UNR)
__global__ __launch_bounds__(256) void k_scale_cols(int64_t n, int m, cd* __restrict__ X, int64_t ldx,
                                                    const double* __restrict__ s, int invert) {
    const int c = blockIdx.y;
    const double f = invert ? 1.0 / s[c] : s[c];
    cd* x = X + (int64_t)c * ldx;
    const int64_t i0 = (int64_t)blockIdx.x * EW_ROWS + threadIdx.x;
    cd v[EW_UNR];
#pragma unroll
    for (int u = 0; u < EW_UNR; ++u) {
        const int64_t i = i0 + u * 256;
        v[u] = i < n ? x[i] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < EW_UNR; ++u) {
        const int64_t i = i0 + u * 256;
        if (i < n) x[i] = make_double2(v[u].x * f, v[u].y * f);
    }
}

__global__ __launch_bounds__(256) void k_copy(int64_t n, int m, const cd* __restrict__ X, int64_t ldx, cd* __restrict__ Y,
                                              int64_t ldy) {
    const int c = blockIdx.y;
    const cd* x = X + (int64_t)c * ldx;
    cd* y = Y + (int64_t)c * ldy;
    const int64_t i0 = (int64_t)blockIdx.x * EW_ROWS + threadIdx.x;
    cd v[EW_UNR];
#pragma unroll
    for (int u = 0; u < EW_UNR; ++u) {
        const int64_t i = i0 + u * 256;
        v[u] = i < n ? x[i] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < EW_UNR; ++u) {
        const int64_t i = i0 + u * 256;
        if (i < n) y[i] = v[u];
    }
}

__global__ __launch_bounds__(256) void k_gather_cols(int64_t n, const cd* __restrict__ X, int64_t ldx,
                                                     const int* __restrict__ perm, cd* __restrict__ Y, int64_t ldy) {
    const int c = blockIdx.y;
    const cd* x = X + (int64_t)perm[c] * ldx;
    cd* y = Y + (int64_t)c * ldy;
    const int64_t i0 = (int64_t)blockIdx.x * EW_ROWS + threadIdx.x;
    cd v[EW_UNR];
#pragma unroll
    for (int u = 0; u < EW_UNR; ++u) {
        const int64_t i = i0 + u * 256;
        v[u] = i < n ? x[i] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < EW_UNR; ++u) {
        const int64_t i = i0 + u * 256;
        if (i < n) y[i] = v[u];
    }
}

// C[row0 + a, a] -= 1 for a in [0, cols)   (the "e" matrix of lobpcg_hyper_impl.jl:493-499)
__global__ void k_sub_identity_shifted(int rows, int cols, cd* __restrict__ C, int64_t ldc, int row0) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < cols && row0 + a < rows) C[(row0 + a) + (int64_t)a * ldc].x -= 1.0;
}

__global__ void k_add_diag(int n, cd* __restrict__ A, int64_t lda, double shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i + (int64_t)i * lda].x += shift;
}

// make A exactly Hermitian from its upper triangle: A[j,i] = conj(A[i,j]) (i<j), Im A[i,i] = 0
__global__ void k_hermitize_upper(int n, cd* __restrict__ A, int64_t lda) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    const int j = (int)(idx / n), i = (int)(idx - (int64_t)j * n);
    if (i == j) A[i + (int64_t)j * lda].y = 0.0;
    if (i < j) {
        const cd v = A[i + (int64_t)j * lda];
        A[j + (int64_t)i * lda] = make_double2(v.x, -v.y);
    }
}

// Y = D * X for the banded real D (n_p x n_p, half bandwidth bw), X is n_p x nb complex
__global__ void k_apply_D(int n_p, int nb, int bw, const double* __restrict__ D, const cd* __restrict__ X,
                          cd* __restrict__ Y) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_p * nb) return;
    const int c = (int)(idx / n_p), i = (int)(idx - (int64_t)c * n_p);
    const int j0 = max(0, i - bw), j1 = min(n_p - 1, i + bw);
    double sr = 0.0, si = 0.0;
    for (int j = j0; j <= j1; ++j) {
        const double d = D[i + (int64_t)j * n_p];
        const cd x = X[j + (int64_t)c * n_p];
        sr += d * x.x;
        si += d * x.y;
    }
    Y[idx] = make_double2(sr, si);
}

// ---------------------------------------------------------------------------- Cholesky + inverse
// Blocked right-looking upper Cholesky (A = R^H R), panel width PB = 32:
//   k_potrf_diag : one wave factors the PB x PB diagonal block in LDS
//   k_trsm_row   : R12 = R11^{-H} A12, one thread per column of the row panel (forward substitution)
//   trailing     : A22 -= R12^H R12 through the f64-MFMA zgemm ('C', alpha = -1, beta = 1)
// info[0] = 0 on success, else the 1-based failing column (non-positive / non-finite pivot).
#define PB 32
__global__ __launch_bounds__(256) void k_potrf_diag(int jb, int j0, cd* __restrict__ A, int64_t lda,
                                                    int* __restrict__ info) {
    // 32x32 diagonal block, upper Cholesky R'R = T.  One thread per 4 elements of the tile (kept in
    // registers); per pivot j only the scaled pivot row travels through LDS: 2 barriers per step.
    __shared__ cd row[PB];
    __shared__ double piv;
    __shared__ int fail;
    const int t = threadIdx.x;
    const int c = t & (PB - 1);          // column of this thread's elements
    const int r0 = t >> 5;               // rows r0, r0 + 8, r0 + 16, r0 + 24
    if (t == 0) fail = info[0];
    cd v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + 8 * q;
        v[q] = (r < jb && c < jb && r <= c) ? A[(j0 + r) + (int64_t)(j0 + c) * lda] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    if (fail != 0) return;   // an earlier panel already failed
    for (int j = 0; j < jb; ++j) {
        const int qj = j >> 3;           // which of the 4 registers holds row j (for threads with r0 == j % 8)
        const bool owns_row = (r0 == (j & 7));
        if (owns_row && c == j) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q == qj) piv = v[q].x;
        }
        __syncthreads();
        const double d = piv;
        if (!(d > 0.0) || !isfinite(d)) {
            if (t == 0) info[0] = j0 + j + 1;
            return;   // uniform: every thread read the same pivot
        }
        const double sd = sqrt(d), inv = 1.0 / sd;
        if (owns_row && c >= j) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q == qj) {
                    v[q] = (c == j) ? make_double2(sd, 0.0) : make_double2(v[q].x * inv, v[q].y * inv);
                    row[c] = v[q];
                }
        }
        __syncthreads();
        // trailing update: T[r][c] -= conj(R[j][r]) * R[j][c] for j < r <= c
        if (c > j) {
            const cd bb = row[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 8 * q;
                if (r > j && r <= c) {
                    const cd a = row[r];
                    v[q].x -= a.x * bb.x + a.y * bb.y;
                    v[q].y -= a.x * bb.y - a.y * bb.x;
                }
            }
        }
        // (the next step's first barrier orders these reads of row[] before it is rewritten)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + 8 * q;
        if (r < jb && c < jb && r <= c) A[(j0 + r) + (int64_t)(j0 + c) * lda] = v[q];
    }
}

// columns c in [j0+PB, n): solve R11^H y = A[j0:j0+PB, c] in place (R11 = A[j0:j0+PB, j0:j0+PB], upper)
__global__ __launch_bounds__(64) void k_trsm_row(int n, int j0, cd* __restrict__ A, int64_t lda,
                                                 const int* __restrict__ info) {
    __shared__ cd R[PB][PB + 1];
    for (int e = threadIdx.x; e < PB * PB; e += 64) {
        const int cc = e / PB, r = e - cc * PB;
        R[r][cc] = A[(j0 + r) + (int64_t)(j0 + cc) * lda];
    }
    __syncthreads();
    if (info[0] != 0) return;
    const int c = j0 + PB + blockIdx.x * 64 + threadIdx.x;
    if (c >= n) return;
    cd y[PB];
    cd* col = A + j0 + (int64_t)c * lda;
#pragma unroll
    for (int i = 0; i < PB; ++i) y[i] = col[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        double sr = y[i].x, si = y[i].y;
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const cd r = R[k][i];   // conj(R[k][i]) * y[k]
            sr -= r.x * y[k].x + r.y * y[k].y;
            si -= r.x * y[k].y - r.y * y[k].x;
        }
        const double inv = 1.0 / R[i][i].x;
        y[i] = make_double2(sr * inv, si * inv);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) col[i] = y[i];
}

// Z = inv(R) for upper-triangular R: one wave per column (back substitution, lane-parallel dots).
__global__ __launch_bounds__(64) void k_trtri_upper(int n, const cd* __restrict__ R, int64_t ldr,
                                                    cd* __restrict__ Z, int64_t ldz) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    cd* z = reinterpret_cast<cd*>(sm);
    const int j = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) z[i] = make_double2(0.0, 0.0);
    __syncthreads();
    for (int i = j; i >= 0; --i) {
        double sr = 0.0, si = 0.0;
        for (int k = i + 1 + lane; k <= j; k += 64) {
            const cd r = R[i + (int64_t)k * ldr];
            const cd zz = z[k];
            sr += r.x * zz.x - r.y * zz.y;
            si += r.x * zz.y + r.y * zz.x;
        }
        sr = wave_sum(sr);
        si = wave_sum(si);
        if (lane == 0) {
            const double d = R[i + (int64_t)i * ldr].x;
            const double rhs_r = (i == j ? 1.0 : 0.0) - sr;
            z[i] = make_double2(rhs_r / d, -si / d);
        }
        __syncthreads();
    }
    for (int i = lane; i < n; i += 64) Z[i + (int64_t)j * ldz] = z[i];
}

// Blocked inverse of an upper-triangular matrix: workgroup b owns TRI_TC columns J of Z = R^{-1}
// (kept in LDS) and back-substitutes them in 32-row blocks, last block first:
//   Z[bi, J] = R[bi,bi]^{-1} ( I[bi, J] - sum_{bk > bi} R[bi,bk] Z[bk, J] ).
// The off-diagonal products read 32x32 tiles of R through LDS (coalesced); only the 32 steps of each
// diagonal block are sequential.  Dynamic LDS: (32*nblk*TRI_TC + 32*33) complex numbers.
#define TRI_TC 8
__global__ __launch_bounds__(256) void k_trtri_upper_blk(int n, const cd* __restrict__ R, int64_t ldr,
                                                         cd* __restrict__ Z, int64_t ldz) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int jc0 = blockIdx.x * TRI_TC;
    const int jmax = min(n, jc0 + TRI_TC) - 1;
    const int nblk = jmax / 32 + 1;
    cd* Zs = reinterpret_cast<cd*>(sm);            // [32 * nblk][TRI_TC]
    cd* Rt = Zs + (size_t)32 * nblk * TRI_TC;      // [32][33]
    const int t = threadIdx.x;
    const int cc = t & (TRI_TC - 1), rr = t / TRI_TC;   // rr in [0, 32): one (row, column) pair per thread
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int i0 = bi * 32;
        cd acc = make_double2((i0 + rr == jc0 + cc) ? 1.0 : 0.0, 0.0);
        for (int bk = nblk - 1; bk > bi; --bk) {
            const int k0 = bk * 32;
            __syncthreads();
            for (int e = t; e < 1024; e += 256) {
                const int k = e >> 5, r = e & 31;
                const int gi = i0 + r, gk = k0 + k;
                Rt[r * 33 + k] = (gi < n && gk < n) ? R[gi + (int64_t)gk * ldr] : make_double2(0.0, 0.0);
            }
            __syncthreads();
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const cd zk = Zs[(k0 + k) * TRI_TC + cc];
                const cd a = Rt[rr * 33 + k];
                acc.x -= a.x * zk.x - a.y * zk.y;
                acc.y -= a.x * zk.y + a.y * zk.x;
            }
        }
        __syncthreads();
        for (int e = t; e < 1024; e += 256) {
            const int k = e >> 5, r = e & 31;
            const int gi = i0 + r, gk = i0 + k;
            Rt[r * 33 + k] = (gi < n && gk < n && r <= k) ? R[gi + (int64_t)gk * ldr] : make_double2(0.0, 0.0);
        }
        __syncthreads();
        for (int k = 31; k >= 0; --k) {
            const int gk = i0 + k;
            if (rr == k) {
                const double d = Rt[k * 33 + k].x;
                Zs[gk * TRI_TC + cc] = (gk < n) ? make_double2(acc.x / d, acc.y / d) : make_double2(0.0, 0.0);
            }
            __syncthreads();
            if (rr < k) {
                const cd zk = Zs[gk * TRI_TC + cc];
                const cd a = Rt[rr * 33 + k];
                acc.x -= a.x * zk.x - a.y * zk.y;
                acc.y -= a.x * zk.y + a.y * zk.x;
            }
        }
    }
    __syncthreads();
    const int ncol = jmax - jc0 + 1;
    for (int e = t; e < n * ncol; e += 256) {
        const int cj = e / n, i = e - cj * n;
        Z[i + (int64_t)(jc0 + cj) * ldz] = (i < 32 * nblk) ? Zs[i * TRI_TC + cj] : make_double2(0.0, 0.0);
    }
}

// ---- cooperative Cholesky + inverse for n <= 512 (round 4) --------------------------------------------------------
// The blocked factorisation above is a chain of 3 launches per 32-wide panel (diagonal block, row panel, trailing GEMM):
// 16 panels x ~75 us = 1.2 ms for the 503^2 Gram matrices of ortho! -- all latency, the flops are nothing.  Here ONE launch of
// nb = ceil(n / 16) workgroups runs the whole factorisation as a dataflow: workgroup l owns block column l of the LOWER
// factor L (A = L L^H, R = L^H is what the caller gets), keeps its rows in REGISTERS (two rows of 16 per thread), consumes
// the panels j < l as their owners publish them (global buffer + release / acquire flag at agent scope: the workgroups sit
// on different XCDs), then factors its own diagonal block, scales its column panel and publishes it.  The critical path per
// panel is: read 16 columns from L2, 256 complex multiply-adds per row, a 16-step Cholesky of the diagonal block in LDS, a
// 16-step substitution per row, publish: ~10 us.  The inverse follows in the same launch: workgroup l forward-substitutes
// block column l of X = L^{-1} (X_kl = -W_k sum_j L_kj X_jl with the published inverse diagonal blocks W_k), and
// R^{-1} = X^H.  All nb <= 32 workgroups are resident at once (1 per CU), which is what makes the flag waits safe.
#define CCB 16
#define CCT 512                                          // threads per workgroup: one row of the block column each
// element type of the factorisation: cd, or double when the caller vouches for a real symmetric matrix (the Gram matrices of
// the Gamma-real LOBPCG: every imaginary part exactly zero) -- a quarter of the multiply-adds, half the bytes
__device__ __forceinline__ cd e_mul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cd e_mulc(cd a, cd b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a conj(b)
__device__ __forceinline__ double e_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double e_mulc(double a, double b) { return a * b; }
__device__ __forceinline__ void e_sub(cd& acc, cd z) { acc.x -= z.x; acc.y -= z.y; }
__device__ __forceinline__ void e_sub(double& acc, double z) { acc -= z; }
__device__ __forceinline__ void e_add(cd& acc, cd z) { acc.x += z.x; acc.y += z.y; }
__device__ __forceinline__ void e_add(double& acc, double z) { acc += z; }
__device__ __forceinline__ cd e_scale(cd a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ double e_scale(double a, double s) { return a * s; }
__device__ __forceinline__ double e_re(cd a) { return a.x; }
__device__ __forceinline__ double e_re(double a) { return a; }
__device__ __forceinline__ void e_set(cd& dst, double re) { dst = make_double2(re, 0.0); }
__device__ __forceinline__ void e_set(double& dst, double re) { dst = re; }
__device__ __forceinline__ void e_load_conj(cd& dst, cd a, bool diag) { dst = make_double2(a.x, diag ? 0.0 : -a.y); }
__device__ __forceinline__ void e_load_conj(double& dst, cd a, bool) { dst = a.x; }
__device__ __forceinline__ cd e_out_conj(cd a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cd e_out_conj(double a) { return make_double2(a, 0.0); }
__device__ __forceinline__ void coop_wait(const int* flag) {
    if (threadIdx.x == 0)
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    __threadfence();
}
__device__ __forceinline__ void coop_signal(int* flag) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__global__ __launch_bounds__(CCT, 1) void k_potrf_trtri_coop(int n, cd* __restrict__ A, int64_t lda, cd* __restrict__ Z,
                                                             int64_t ldz, T* __restrict__ Lbuf, T* __restrict__ Wbuf,
                                                             int* __restrict__ flags, int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    T* Xs = reinterpret_cast<T*>(sm);                   // phase 2: [rows][16] block column of X = L^{-1}
    __shared__ T Dm[CCB][CCB + 1];                       // diagonal block (lower), then scratch
    __shared__ T Wm[CCB][CCB + 1];                       // its inverse
    __shared__ T Lj[CCB][CCB + 1];                       // L_{l,j} of the panel being consumed
    const int l = blockIdx.x, nb = gridDim.x, tid = threadIdx.x;
    const int r0 = CCB * l;                              // first global row / column of this block
    const int np = CCB * nb;                             // padded order
    const int nrows = np - r0;                           // rows of this block column (diagonal block first)
    int* flagsW = flags + 32;                            // second set: inverse diagonal blocks published
    // panel j lives at Lbuf + j * np * CCB as [row - 16 j][16] (row-major: a row is contiguous)
    auto panel = [&](int j) { return Lbuf + (size_t)j * np * CCB; };
    // ---- load my row of the block column of the lower triangle: L(r, c) source = conj(A_upper(c, r)); identity in the padding
    const int i = tid, gr = r0 + i;
    const bool have_row = i < nrows;
    T row[CCB];
#pragma unroll
    for (int c = 0; c < CCB; ++c) {
        const int gc = r0 + c;
        e_set(row[c], 0.0);
        if (have_row) {
            if (gr < n && gc < n) {
                if (gr >= gc) e_load_conj(row[c], A[gc + (int64_t)gr * lda], gr == gc);
            } else if (gr == gc) {
                e_set(row[c], 1.0);
            }
        }
    }
    // ---- consume the panels of the block columns to the left
    for (int j = 0; j < l; ++j) {
        coop_wait(flags + j);
        const T* P = panel(j) + (size_t)(r0 - CCB * j) * CCB;      // my first row inside panel j
        if (tid < CCB * CCB) Lj[tid >> 4][tid & 15] = P[tid];      // rows r0 .. r0+15 of the panel = L_{l,j}
        T p[CCB];
        if (have_row) {
#pragma unroll
            for (int t = 0; t < CCB; ++t) p[t] = P[(size_t)i * CCB + t];
        }
        __syncthreads();
        if (have_row) {
#pragma unroll
            for (int c = 0; c < CCB; ++c) {
                T s;
                e_set(s, 0.0);
#pragma unroll
                for (int t = 0; t < CCB; ++t) e_add(s, e_mulc(p[t], Lj[c][t]));
                e_sub(row[c], s);
            }
        }
        __syncthreads();   // Lj is rewritten by the next panel
    }
    // ---- factor the diagonal block (threads 0..15 hold its rows): wave 0 alone, LDS, no workgroup barriers
    if (tid < CCB) {
#pragma unroll
        for (int c = 0; c < CCB; ++c) Dm[tid][c] = row[c];
    }
    __syncthreads();
    if (tid < 64) {
        // lane owns elements (k, m) = (4 q + (lane >> 4), lane & 15), q = 0..3
        const int m = tid & 15, kb = tid >> 4;
        for (int c = 0; c < CCB; ++c) {
            const double d = e_re(Dm[c][c]);
            const bool ok = d > 0.0 && isfinite(d);
            if (!ok && tid == 0) atomicCAS(info, 0, r0 + c + 1);
            const double sd = ok ? sqrt(d) : 1.0, inv = 1.0 / sd;
            __builtin_amdgcn_wave_barrier();
            if (m == c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 4 * q + kb;
                    if (k == c)
                        e_set(Dm[k][c], sd);
                    else if (k > c)
                        Dm[k][c] = e_scale(Dm[k][c], inv);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (m > c) {
                const T lm = Dm[m][c];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 4 * q + kb;
                    if (k >= m) e_sub(Dm[k][m], e_mulc(Dm[k][c], lm));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + kb;
            if (m > k) e_set(Dm[k][m], 0.0);
        }
    }
    __syncthreads();
    // ---- column panel below the diagonal block: y D^H = x  per row;  the diagonal block rows take D itself
    if (have_row) {
        if (i < CCB) {
#pragma unroll
            for (int c = 0; c < CCB; ++c) row[c] = Dm[i][c];
        } else {
#pragma unroll
            for (int c = 0; c < CCB; ++c) {
                T acc = row[c];
#pragma unroll
                for (int t = 0; t < CCB; ++t)
                    if (t < c) e_sub(acc, e_mulc(row[t], Dm[c][t]));
                row[c] = e_scale(acc, 1.0 / e_re(Dm[c][c]));
            }
        }
        // ---- publish my row of panel l
        T* P = panel(l);
#pragma unroll
        for (int c = 0; c < CCB; ++c) P[(size_t)i * CCB + c] = row[c];
    }
    coop_signal(flags + l);
    // ---- off the critical path: R = L^H into the caller's upper triangle, the inverse diagonal block
    if (have_row) {
#pragma unroll
        for (int c = 0; c < CCB; ++c) {
            const int gc = r0 + c;
            if (gr < n && gc < n && gr >= gc) A[gc + (int64_t)gr * lda] = e_out_conj(row[c]);
        }
    }
    if (tid < CCB) {   // column m of W = D^{-1} by forward substitution
        const int m = tid;
        T w[CCB];
#pragma unroll
        for (int k = 0; k < CCB; ++k) {
            T acc;
            e_set(acc, k == m ? 1.0 : 0.0);
#pragma unroll
            for (int t = 0; t < CCB; ++t)
                if (t < k) e_sub(acc, e_mul(Dm[k][t], w[t]));
            w[k] = e_scale(acc, 1.0 / e_re(Dm[k][k]));
            if (k < m) e_set(w[k], 0.0);
        }
#pragma unroll
        for (int k = 0; k < CCB; ++k) Wm[k][m] = w[k];
    }
    __syncthreads();
    if (tid < CCB * CCB) Wbuf[(size_t)l * CCB * CCB + tid] = Wm[tid >> 4][tid & 15];
    coop_signal(flagsW + l);
    // ---- phase 2: block column l of X = L^{-1}:  X_ll = W_l,  X_kl = -W_k sum_{j=l}^{k-1} L_kj X_jl
    // threads (half, a, b): the j range of the sum is split over the two halves of the workgroup
    const int a = (tid >> 4) & 15, b = tid & 15, half = tid >> 8;
    if (tid < CCB * CCB) Xs[a * CCB + b] = Wm[a][b];
    __syncthreads();
    for (int k = l + 1; k < nb; ++k) {
        coop_wait(flagsW + k);
        T s;
        e_set(s, 0.0);
        for (int j = l + half; j < k; j += 2) {
            const T* Lkj = panel(j) + (size_t)(CCB * (k - j)) * CCB + a * CCB;   // row a of L_{k,j}
            const T* Xj = Xs + (size_t)(CCB * (j - l)) * CCB + b;
#pragma unroll
            for (int t = 0; t < CCB; ++t) e_add(s, e_mul(Lkj[t], Xj[t * CCB]));
        }
        if (half == 1) Wm[a][b] = s;
        __syncthreads();
        if (half == 0) {
            e_add(s, Wm[a][b]);
            Lj[a][b] = s;                                              // S
            Dm[a][b] = Wbuf[(size_t)k * CCB * CCB + (tid & 255)];        // W_k
        }
        __syncthreads();
        if (half == 0) {
            T acc;
            e_set(acc, 0.0);
#pragma unroll
            for (int t = 0; t < CCB; ++t) e_sub(acc, e_mul(Dm[a][t], Lj[t][b]));
            Xs[(size_t)(CCB * (k - l) + a) * CCB + b] = acc;
        }
        __syncthreads();
    }
    // ---- Z = R^{-1} = X^H: my block column of X is the block ROW l of Z; zeros below the diagonal
    for (int e = tid; e < CCB * np; e += CCT) {
        const int bb = e / np, gcol = e - bb * np;        // Z(r0 + bb, gcol)
        const int grow = r0 + bb;
        if (grow < n && gcol < n) {
            cd v = make_double2(0.0, 0.0);
            if (gcol >= grow) v = e_out_conj(Xs[(size_t)(gcol - r0) * CCB + bb]);   // conj X(gcol, grow)
            Z[grow + (int64_t)gcol * ldz] = v;
        }
    }
}

// Partial results of column chunk blockIdx.x of matrix blockIdx.y -- (M, out) or (M2, out2), both estimates in
// one launch: out[3 chunk + 0] = max |diag|, [+1] = sum of |offdiag|^2 over the upper triangle, [+2] = non-finite
// flag.  The host combines the NORMEST_CHUNKS partials in fixed order.  A wave owns a column (rows contiguous:
// coalesced, no index division); columns are dealt round-robin so that the triangle is balanced.
#define NORMEST_CHUNKS 8
__global__ __launch_bounds__(256) void k_normest_upper(int n, const cd* __restrict__ M, int64_t ldm,
                                                       double* __restrict__ out, const cd* __restrict__ M2,
                                                       int64_t ldm2, double* __restrict__ out2) {
    if (blockIdx.y == 1) {
        M = M2;
        ldm = ldm2;
        out = out2;
    }
    __shared__ double sh[4];
    __shared__ double smax[256];
    double off = 0.0, mx = 0.0, bad = 0.0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = blockIdx.x * 4 + wave; c < n; c += 4 * NORMEST_CHUNKS) {
        const cd* col = M + (int64_t)c * ldm;
        for (int r = lane; r <= c; r += 64) {
            const cd v = col[r];
            if (!(isfinite(v.x) && isfinite(v.y))) bad = 1.0;
            const double a2 = v.x * v.x + v.y * v.y;
            if (r == c)
                mx = fmax(mx, sqrt(a2));
            else
                off += a2;
        }
    }
    smax[threadIdx.x] = mx;
    __syncthreads();
    const double o = block_sum256(off, sh);
    const double bsum = block_sum256(bad, sh);
    if (threadIdx.x == 0) {
        double m2 = 0.0;
        for (int i = 0; i < 256; ++i) m2 = fmax(m2, smax[i]);
        out[3 * blockIdx.x + 0] = m2;
        out[3 * blockIdx.x + 1] = o;
        out[3 * blockIdx.x + 2] = bsum;
    }
}

// ---------------------------------------------------------------------------- Hermitian eigensolver
// Blocked two-sided Jacobi with round-robin (tournament) ordering.  Block size JB; per round the
// n/(2 JB) disjoint block pairs are diagonalised approximately in LDS by one cyclic Jacobi sweep (the
// "pair" part), and the accumulated 32 x 32 rotations U are applied two-sided to A and to the columns of V
// (the "update" part, f64 MFMA).
// Both parts of consecutive rounds run in ONE launch (k_jacobi_round): the pair workgroups of round r do
// not wait for the update of round r-1 -- they apply U(r-1) themselves to the three 32 x 32 tiles their
// 2 x 2 block problem lives in (reading the matrix as it was BEFORE round r-1, hence the update writes out of
// place into a second copy) while the update workgroups of round r-1 fill the rest of the chip.  The
// latency-bound pair solve is the whole critical path: one launch of ~22 us per round instead of two
// (18 + 13 us).
typedef double v4d_t __attribute__((ext_vector_type(4)));
// storage type of the work matrices W, V and the rotation blocks: complex, or plain doubles for a real symmetric
// input (REAL: half the bytes per round -- at n = 1509 a round streams all of W and V through the memory side)
template <bool REAL>
struct JacEl {
    typedef cd T;
};
template <>
struct JacEl<true> {
    typedef double T;
};
__device__ __forceinline__ cd jac_ld(const cd& v) { return v; }
__device__ __forceinline__ cd jac_ld(const double& v) { return make_double2(v, 0.0); }
__device__ __forceinline__ void jac_st(cd& dst, double re, double im) { dst = make_double2(re, im); }
__device__ __forceinline__ void jac_st(double& dst, double re, double) { dst = re; }
#define JB 16
#define J2B (2 * JB)
// DFTK_MI_HEEV_CLOCK=1 (flags bit 2 of k_jacobi_round): wall ticks (100 MHz) of the phases of pair workgroup 0 and of the first
// update workgroup, summed over the launches of one dense_heev call: [0] launches, [1] pair: entry -> block problem in LDS,
// [2] the inner rounds, [3] U written, [4] update workgroup entry -> exit, [5] shader cycles of the inner rounds
__device__ unsigned long long g_jac_clk[8];
// same switch: [4 i] = first workgroup entry, [4 i + 1] / [4 i + 2] = last exit of a pair / update workgroup (absolute wall
// ticks) of launch i < 1024; sampled: every pair workgroup, every 16th update workgroup
__device__ unsigned long long g_jac_span[4096];
// entry / exit ticks of pair workgroup p < 64 of launch i < 1024: [(i * 64 + p) * 2 + {0, 1}]
__device__ unsigned long long g_jac_pair[1024 * 64 * 2];

__host__ __device__ __forceinline__ void tournament_pair(int nb, int round, int k, int& p, int& q) {
    // nb even, round in [0, nb-1), k in [0, nb/2): pair k of the round; round < 0: the fixed pairing (2k, 2k+1)
    if (round < 0) {
        p = 2 * k;
        q = 2 * k + 1;
        return;
    }
    int a, b2;
    if (k == 0) {
        a = nb - 1;
        b2 = round;
    } else {
        a = (round + k) % (nb - 1);
        b2 = (round - k + (nb - 1)) % (nb - 1);
    }
    p = a < b2 ? a : b2;
    q = a < b2 ? b2 : a;
}

// inverse of tournament_pair: block `blk` is member h (0: the smaller index) of pair k in `round`
__host__ __device__ __forceinline__ void tournament_find(int nb, int round, int blk, int& k, int& h) {
    if (round < 0) {
        k = blk >> 1;
        h = blk & 1;
        return;
    }
    const int m = nb - 1;
    if (blk == m || blk == round) {
        k = 0;
    } else {
        const int d = (blk - round + m) % m;
        k = (d <= nb / 2 - 1) ? d : m - d;
    }
    int p, q;
    tournament_pair(nb, round, k, p, q);
    h = (blk == p) ? 0 : 1;
}

// mode 0 ("cross", round in [0, nb-1)): tournament block pair (bp, bq); rotates the JB*JB cross pairs
//        (i, JB + (i + r) % JB), r = 0..JB-1 -- every (i, j) with i in block p and j in block q once.
// mode 1 ("diag", one launch per sweep): block pair (2k, 2k+1); rotates the within-block pairs of both
//        blocks (tournament of JB) -- together with the cross rounds each index pair is met once per sweep.
// One inner round = ONE barrier: S is double-buffered (read S[cur], write S[cur ^ 1]); thread (k1, k2)
// recomputes the rotations of pairs k1 and k2 itself from S[cur] (no parameter exchange through LDS, no
// serial 16-thread phase), applies them two-sided to its 2x2 block and column-wise to two rows of U
// (in place: nobody else touches those entries in this round).
// Look-ahead (have_prev): A is the matrix BEFORE round prev_round and Uprev that round's rotations; the
// 2 x 2 block problem of this round is  S[r][c] = u_r^H A[rows(P(r)), cols(P(c))] u_c  with P(.) the
// prev_round pair a block belonged to and u the matching 16 columns of that pair's U.
// REAL: the matrix is real symmetric stored as complex with zero imaginary parts (Rayleigh-Ritz of the Gamma-real
// LOBPCG): one real MFMA per complex quadruple, real rotations; imaginary parts are written as exact zeros.
template <bool REAL>
__device__ __forceinline__ void jacobi_pair_part(int pair_id, int nb, int round, int prev_round, bool have_prev,
                                                 const typename JacEl<REAL>::T* __restrict__ A, int64_t lda,
                                                 const typename JacEl<REAL>::T* __restrict__ Uprev,
                                                 typename JacEl<REAL>::T* __restrict__ Ubuf,
                                                 typename JacEl<REAL>::T (*S)[J2B][J2B + 1],
                                                 typename JacEl<REAL>::T (*U)[J2B + 1], bool clk) {
    typedef typename JacEl<REAL>::T ET;
    const bool stamp = clk && pair_id == 0 && threadIdx.x == 0;
    long long w0 = 0, w1 = 0, c1 = 0;
    if (stamp) w0 = wall_clock64();
    const int mode = round < 0 ? 1 : 0;
    int bp, bq;
    tournament_pair(nb, round, pair_id, bp, bq);
    const int tid = threadIdx.x;
    if (!have_prev) {
        // load the 2x2 block sub-matrix (global index of local i)
        for (int e = tid; e < J2B * J2B; e += 256) {
            const int c = e / J2B, r = e - c * J2B;
            const int gr = (r < JB ? bp * JB + r : bq * JB + (r - JB));
            const int gc = (c < JB ? bp * JB + c : bq * JB + (c - JB));
            S[0][r][c] = A[gr + (int64_t)gc * lda];
        }
    } else {
        int ka, ha, kb, hb, pa, qa, pb, qb;
        tournament_find(nb, prev_round, bp, ka, ha);
        tournament_find(nb, prev_round, bq, kb, hb);
        tournament_pair(nb, prev_round, ka, pa, qa);
        tournament_pair(nb, prev_round, kb, pb, qb);
        // waves 0..2 take the three tiles (row side, column side) = (a,a), (a,b), (b,b); only the 16 x 16
        // quadrant (h_row, h_col) of  U_row^H T U_col  is needed.  f64 MFMA straight from global operands: the
        // accumulator layout of  Y = T U_col[:, h_col]  (value r = row lk + 4r) is exactly the B-operand layout
        // of the second product, so Y never leaves the registers (96 MFMAs per wave).
        const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
        if (wave < 3) {
            const bool rb = wave == 2, cb = wave >= 1;
            const int rp = rb ? pb : pa, rq = rb ? qb : qa, cp = cb ? pb : pa, cq = cb ? qb : qa;
            const ET* UR = Uprev + (int64_t)(rb ? kb : ka) * J2B * J2B + (int64_t)(JB * (rb ? hb : ha)) * J2B;
            const ET* UC = Uprev + (int64_t)(cb ? kb : ka) * J2B * J2B + (int64_t)(JB * (cb ? hb : ha)) * J2B;
            cd fa[2][8], uc[8], ur[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = 4 * t + lk;
                const int64_t gc = (k < JB ? cp * JB + k : cq * JB + (k - JB));
                fa[0][t] = jac_ld(A[(rp * JB + li) + gc * lda]);
                fa[1][t] = jac_ld(A[(rq * JB + li) + gc * lda]);
                uc[t] = jac_ld(UC[k + li * J2B]);
                ur[t] = jac_ld(UR[k + li * J2B]);   // A operand of the second product: conj(U_row[k][row li])
            }
            v4d_t yR[2], yI[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                yR[h] = yI[h] = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const double ar = fa[h][t].x, ai = fa[h][t].y, nai = -ai;
                    yR[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, uc[t].x, yR[h], 0, 0, 0);
                    if (!REAL) {
                        yI[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, uc[t].y, yI[h], 0, 0, 0);
                        yR[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, uc[t].y, yR[h], 0, 0, 0);
                        yI[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, uc[t].x, yI[h], 0, 0, 0);
                    }
                }
            }
            v4d_t zR = (v4d_t){0.0, 0.0, 0.0, 0.0}, zI = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // k-step (h, r): k = 16 h + 4 r + lk
                    const cd u = ur[4 * h + r];
                    const double br = yR[h][r], bi = yI[h][r], nui = -u.y;
                    zR = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, br, zR, 0, 0, 0);
                    if (!REAL) {
                        zI = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, bi, zI, 0, 0, 0);
                        zR = __builtin_amdgcn_mfma_f64_16x16x4f64(u.y, bi, zR, 0, 0, 0);
                        zI = __builtin_amdgcn_mfma_f64_16x16x4f64(nui, br, zI, 0, 0, 0);
                    }
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = lk + 4 * r;
                jac_st(S[0][JB * (rb ? 1 : 0) + i][JB * (cb ? 1 : 0) + li], zR[r], zI[r]);
                if (wave == 1) jac_st(S[0][JB + li][i], zR[r], -zI[r]);
            }
        }
    }
    for (int e = tid; e < J2B * J2B; e += 256) {
        const int c = e / J2B, r = e - c * J2B;
        jac_st(U[r][c], r == c ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    if (stamp) {
        w1 = wall_clock64();
        c1 = clock64();
    }
    const int k1 = tid >> 4, k2 = tid & 15;
    const int nrounds = mode == 0 ? JB : JB - 1;
    int cur = 0;
    for (int rd = 0; rd < nrounds; ++rd) {
        const ET(*Sc)[J2B + 1] = S[cur];
        ET(*Sn)[J2B + 1] = S[cur ^ 1];
        // index pair (p, q) and Jacobi rotation (c, s) of rotation slot k in this round
        auto pair_of = [&](int k, int& p, int& q) {
            if (mode == 0) {
                p = k;
                q = JB + ((k + rd) & (JB - 1));
            } else {
                tournament_pair(JB, rd, k & (JB / 2 - 1), p, q);
                if (k >= JB / 2) {
                    p += JB;
                    q += JB;
                }
            }
        };
        // Jacobi rotation annihilating S[p][q]:  t = sgn(d) |b| / (|d| + sqrt(d^2 + |b|^2)), d = (g - a)/2,
        // c = 1/sqrt(1 + t^2), s = t c b/|b| = sgn(d) c b / (|d| + sqrt(d^2 + |b|^2))  -- no |b| needed.
        // Hardware rsq/rcp seeds + Newton steps instead of the library sqrt/div sequences: every thread
        // evaluates two of these per round, so they must be short; c gets two steps (c^2 + |s|^2 = 1 to
        // round-off keeps U unitary), the angle itself needs far less than full precision.
        // The round is a pure latency chain (one wave per SIMD, 16 rounds per launch, DFTK_MI_HEEV_CLOCK): ALL LDS operands
        // of the round are fetched in one batch, and the rotations are evaluated without branches (a skipped rotation is
        // a select at the end; its intermediate inf / NaN never leaves the registers) so that the two dependent chains of
        // a thread interleave instead of running one after the other behind their own LDS waits.
        // both rotations of the thread side by side (A = slot k1, B = slot k2)
        auto rotation2 = [&](cd beA, double alA, double gaA, cd beB, double alB, double gaB, double& cA, cd& sA, double& cB,
                             cd& sB) {
            const double b2A = REAL ? beA.x * beA.x : beA.x * beA.x + beA.y * beA.y;
            const double b2B = REAL ? beB.x * beB.x : beB.x * beB.x + beB.y * beB.y;
            const bool actA = b2A > 1e-300 && b2A > 1e-36 * (fabs(alA * gaA) + 1e-300);
            const bool actB = b2B > 1e-300 && b2B > 1e-36 * (fabs(alB * gaB) + 1e-300);
            const double dA = 0.5 * (gaA - alA), dB = 0.5 * (gaB - alB);
            const double xA = fma(dA, dA, b2A), xB = fma(dB, dB, b2B);
            double rA = __builtin_amdgcn_rsq(xA), rB = __builtin_amdgcn_rsq(xB);
            const double hA = -0.5 * xA * rA, hB = -0.5 * xB * rB;
            rA = rA * fma(hA, rA, 1.5);
            rB = rB * fma(hB, rB, 1.5);
            const double denA = fabs(dA) + xA * rA, denB = fabs(dB) + xB * rB;
            double uA = __builtin_amdgcn_rcp(denA), uB = __builtin_amdgcn_rcp(denB);
            uA = uA * fma(-denA, uA, 2.0);
            uB = uB * fma(-denB, uB, 2.0);
            const double yA = fma(b2A * uA, uA, 1.0), yB = fma(b2B * uB, uB, 1.0);
            double ccA = __builtin_amdgcn_rsq(yA), ccB = __builtin_amdgcn_rsq(yB);
            const double gA = -0.5 * yA, gB = -0.5 * yB;
            ccA = ccA * fma(gA * ccA, ccA, 1.5);
            ccB = ccB * fma(gB * ccB, ccB, 1.5);
            ccA = ccA * fma(gA * ccA, ccA, 1.5);
            ccB = ccB * fma(gB * ccB, ccB, 1.5);
            const double fA = (dA >= 0.0 ? ccA : -ccA) * uA, fB = (dB >= 0.0 ? ccB : -ccB) * uB;
            cA = actA ? ccA : 1.0;
            cB = actB ? ccB : 1.0;
            sA = make_double2(actA ? fA * beA.x : 0.0, (REAL || !actA) ? 0.0 : fA * beA.y);
            sB = make_double2(actB ? fB * beB.x : 0.0, (REAL || !actB) ? 0.0 : fB * beB.y);
        };
        int p1, q1, p2, q2;
        pair_of(k1, p1, q1);
        pair_of(k2, p2, q2);
        const cd be1 = jac_ld(Sc[p1][q1]), be2 = jac_ld(Sc[p2][q2]);
        const double al1 = jac_ld(Sc[p1][p1]).x, ga1 = jac_ld(Sc[q1][q1]).x;
        const double al2 = jac_ld(Sc[p2][p2]).x, ga2 = jac_ld(Sc[q2][q2]).x;
        const ET a = Sc[p1][p2], b2 = Sc[p1][q2], c3 = Sc[q1][p2], d = Sc[q1][q2];
        const ET x0p = U[k2][p1], x0q = U[k2][q1], x1p = U[k2 + JB][p1], x1q = U[k2 + JB][q1];
        __builtin_amdgcn_sched_barrier(0);   // the 14 LDS reads above are issued before any of the arithmetic below
        double c1, c2;
        cd s1, s2;
        rotation2(be1, al1, ga1, be2, al2, ga2, c1, s1, c2, s2);   // (k1 == k2: the same inputs, the same rotation twice)
        if constexpr (REAL) {
            const double ra = c1 * a - s1.x * c3, rb = c1 * b2 - s1.x * d;
            const double rc = s1.x * a + c1 * c3, rdd = s1.x * b2 + c1 * d;
            Sn[p1][p2] = c2 * ra - s2.x * rb;
            Sn[p1][q2] = s2.x * ra + c2 * rb;
            Sn[q1][p2] = c2 * rc - s2.x * rdd;
            Sn[q1][q2] = s2.x * rc + c2 * rdd;
            U[k2][p1] = c1 * x0p - s1.x * x0q;
            U[k2][q1] = s1.x * x0p + c1 * x0q;
            U[k2 + JB][p1] = c1 * x1p - s1.x * x1q;
            U[k2 + JB][q1] = s1.x * x1p + c1 * x1q;
        } else {
            // rows: (row_p, row_q) <- (c row_p - s row_q, conj(s) row_p + c row_q)
            const cd ra = make_double2(c1 * a.x - (s1.x * c3.x - s1.y * c3.y), c1 * a.y - (s1.x * c3.y + s1.y * c3.x));
            const cd rb = make_double2(c1 * b2.x - (s1.x * d.x - s1.y * d.y), c1 * b2.y - (s1.x * d.y + s1.y * d.x));
            const cd rc = make_double2(s1.x * a.x + s1.y * a.y + c1 * c3.x, s1.x * a.y - s1.y * a.x + c1 * c3.y);
            const cd rdd = make_double2(s1.x * b2.x + s1.y * b2.y + c1 * d.x, s1.x * b2.y - s1.y * b2.x + c1 * d.y);
            // cols: (x_p, x_q) <- (c x_p - conj(s) x_q, s x_p + c x_q)
            Sn[p1][p2] = make_double2(c2 * ra.x - (s2.x * rb.x + s2.y * rb.y), c2 * ra.y - (s2.x * rb.y - s2.y * rb.x));
            Sn[p1][q2] = make_double2(s2.x * ra.x - s2.y * ra.y + c2 * rb.x, s2.x * ra.y + s2.y * ra.x + c2 * rb.y);
            Sn[q1][p2] = make_double2(c2 * rc.x - (s2.x * rdd.x + s2.y * rdd.y), c2 * rc.y - (s2.x * rdd.y - s2.y * rdd.x));
            Sn[q1][q2] = make_double2(s2.x * rc.x - s2.y * rc.y + c2 * rdd.x, s2.x * rc.y + s2.y * rc.x + c2 * rdd.y);
            // column rotations of U: rows k2 and k2 + 16, rotation k1
            U[k2][p1] = make_double2(c1 * x0p.x - (s1.x * x0q.x + s1.y * x0q.y), c1 * x0p.y - (s1.x * x0q.y - s1.y * x0q.x));
            U[k2][q1] = make_double2(s1.x * x0p.x - s1.y * x0p.y + c1 * x0q.x, s1.x * x0p.y + s1.y * x0p.x + c1 * x0q.y);
            U[k2 + JB][p1] = make_double2(c1 * x1p.x - (s1.x * x1q.x + s1.y * x1q.y), c1 * x1p.y - (s1.x * x1q.y - s1.y * x1q.x));
            U[k2 + JB][q1] = make_double2(s1.x * x1p.x - s1.y * x1p.y + c1 * x1q.x, s1.x * x1p.y + s1.y * x1p.x + c1 * x1q.y);
        }
        __syncthreads();
        cur ^= 1;
    }
    long long w2 = 0, c2 = 0;
    if (stamp) {
        w2 = wall_clock64();
        c2 = clock64();
    }
    ET* Uo = Ubuf + (int64_t)pair_id * J2B * J2B;
    for (int e = tid; e < J2B * J2B; e += 256) {
        const int c = e / J2B, r = e - c * J2B;
        Uo[e] = U[r][c];   // column-major 2b x 2b
    }
    if (stamp) {
        const long long w3 = wall_clock64();
        atomicAdd(&g_jac_clk[0], 1ull);
        atomicAdd(&g_jac_clk[1], (unsigned long long)(w1 - w0));
        atomicAdd(&g_jac_clk[2], (unsigned long long)(w2 - w1));
        atomicAdd(&g_jac_clk[3], (unsigned long long)(w3 - w2));
        atomicAdd(&g_jac_clk[5], (unsigned long long)(c2 - c1));
    }
}

__device__ __forceinline__ int pair_index(int bp, int bq, int kk) {
    return kk < JB ? bp * JB + kk : bq * JB + (kk - JB);
}

// Two-sided update of one round (half the work on A thanks to the Hermitian symmetry; f64 MFMA), out of place
// A -> Aout (every tile is rewritten: each block is in exactly one pair):
//   ids [0, ntiles):  tile (i <= j) of the pair partition, Aout_ij = U_i^H A_ij U_j (32x32), the mirror
//                     tile Aout_ji = Aout_ij^H is written along with it;  wave w owns the output quadrant
//                     (w >> 1, w & 1): 32 MFMAs for T = A_ij U_j (through LDS), 32 for U_i^H T;
//   ids [ntiles, ..): V[:, cols(pair)] <- V[:, cols(pair)] U_pair (in place), one wave per 16-row strip.
template <bool REAL>
__device__ __forceinline__ void jacobi_update_part(int id, int n, int nb, int round,
                                                   const typename JacEl<REAL>::T* __restrict__ A,
                                                   typename JacEl<REAL>::T* __restrict__ Aout, int64_t lda,
                                                   typename JacEl<REAL>::T* __restrict__ V, int64_t ldv,
                                                   const typename JacEl<REAL>::T* __restrict__ Ubuf, int ntiles,
                                                   int vblocks, typename JacEl<REAL>::T (*Ts)[J2B + 1]) {
    typedef typename JacEl<REAL>::T ET;
    const int npairs = nb / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    if (id >= ntiles) {
        // ---- eigenvector columns
        const int vb = id - ntiles;
        const int pj = vb / vblocks;
        const int r0 = ((vb - pj * vblocks) * 4 + wave) * 16;
        if (r0 >= n) return;
        int bp, bq;
        tournament_pair(nb, round, pj, bp, bq);
        const ET* U = Ubuf + (int64_t)pj * J2B * J2B;
        v4d_t accR[2], accI[2];
        accR[0] = accR[1] = accI[0] = accI[1] = (v4d_t){0.0, 0.0, 0.0, 0.0};
        cd fa[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) fa[t] = jac_ld(V[r0 + li + (int64_t)pair_index(bp, bq, 4 * t + lk) * ldv]);
#pragma unroll
        // the TRANSPOSED product (V U)^T = U^T V^T: same registers with the MFMA operands swapped; its accumulator
        // layout (value r = column 16c + lk + 4r of the pair, lane li = row r0 + li) makes the stores below 16
        // consecutive rows per column instead of 16 elements a whole column apart
        for (int t = 0; t < 8; ++t) {
            const double ar = fa[t].x, ai = fa[t].y;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const cd u = jac_ld(U[(4 * t + lk) + (16 * c + li) * J2B]);
                accR[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, ar, accR[c], 0, 0, 0);
                if (!REAL) {
                    const double nuy = -u.y;
                    accI[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, ai, accI[c], 0, 0, 0);
                    accR[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(nuy, ai, accR[c], 0, 0, 0);
                    accI[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(u.y, ar, accI[c], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gc = pair_index(bp, bq, 16 * c + lk + 4 * r);
                jac_st(V[r0 + li + gc * ldv], accR[c][r], accI[c][r]);
            }
        return;
    }
    // ---- tile (pi <= pj): linear index -> (pi, pj) of the upper triangle (row by row)
    int pi = 0, rem = id;
    while (rem >= npairs - pi) {
        rem -= npairs - pi;
        ++pi;
    }
    const int pj = pi + rem;
    int bpi, bqi, bpj, bqj;
    tournament_pair(nb, round, pi, bpi, bqi);
    tournament_pair(nb, round, pj, bpj, bqj);
    const ET* Ui = Ubuf + (int64_t)pi * J2B * J2B;
    const ET* Uj = Ubuf + (int64_t)pj * J2B * J2B;
    const int qi = wave >> 1, qj = wave & 1;
    // stage 1: T[qi, qj] = A_ij[qi rows, :] * U_j[:, qj cols]
    {
        v4d_t tR = (v4d_t){0.0, 0.0, 0.0, 0.0}, tI = (v4d_t){0.0, 0.0, 0.0, 0.0};
        const int gr = pair_index(bpi, bqi, 16 * qi + li);
        cd fa[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) fa[t] = jac_ld(A[gr + (int64_t)pair_index(bpj, bqj, 4 * t + lk) * lda]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const double ar = fa[t].x, ai = fa[t].y, nai = -ai;
            const cd u = jac_ld(Uj[(4 * t + lk) + (16 * qj + li) * J2B]);
            tR = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, u.x, tR, 0, 0, 0);
            if (!REAL) {
                tI = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, u.y, tI, 0, 0, 0);
                tR = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, u.y, tR, 0, 0, 0);
                tI = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, u.x, tI, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) jac_st(Ts[16 * qi + lk + 4 * r][16 * qj + li], tR[r], tI[r]);
    }
    __syncthreads();
    // stage 2: R[qi, qj] = U_i[:, qi cols]^H * T[:, qj cols]
    v4d_t rR = (v4d_t){0.0, 0.0, 0.0, 0.0}, rI = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const cd u = jac_ld(Ui[(4 * t + lk) + (16 * qi + li) * J2B]);   // A operand: conj(U_i[k][row])
        const cd b = jac_ld(Ts[4 * t + lk][16 * qj + li]);
        const double nui = -u.y;
        rR = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, b.x, rR, 0, 0, 0);
        if (!REAL) {
            rI = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, b.y, rI, 0, 0, 0);
            rR = __builtin_amdgcn_mfma_f64_16x16x4f64(u.y, b.y, rR, 0, 0, 0);
            rI = __builtin_amdgcn_mfma_f64_16x16x4f64(nui, b.x, rI, 0, 0, 0);
        }
    }
    const int64_t gc = pair_index(bpj, bqj, 16 * qj + li);
    if constexpr (REAL) {
        // R^T = T^T U_i from the same registers with swapped operands (8 cheap MFMAs): value r = column
        // 16 qj + lk + 4r, lane li = row 16 qi + li -> the direct tile is stored as runs of 16 consecutive rows, like
        // the mirror tile (the accumulator layout of R itself would scatter 16 lanes over 16 columns)
        v4d_t rT = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 8; ++t)
            rT = __builtin_amdgcn_mfma_f64_16x16x4f64(Ts[4 * t + lk][16 * qj + li], Ui[(4 * t + lk) + (16 * qi + li) * J2B],
                                                      rT, 0, 0, 0);
        const int64_t grl = pair_index(bpi, bqi, 16 * qi + li);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Aout[grl + (int64_t)pair_index(bpj, bqj, 16 * qj + lk + 4 * r) * lda] = rT[r];
            if (pi != pj) Aout[gc + (int64_t)pair_index(bpi, bqi, 16 * qi + lk + 4 * r) * lda] = rR[r];   // mirror
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gr = pair_index(bpi, bqi, 16 * qi + lk + 4 * r);
            jac_st(Aout[gr + gc * lda], rR[r], rI[r]);
            if (pi != pj) jac_st(Aout[gc + gr * lda], rR[r], -rI[r]);   // mirror tile A_ji = A_ij^H
        }
    }
}

// workgroups per CU the round kernel is compiled for (REAL: 83 VGPRs -> 80: one more resident workgroup per CU for the
// update part, which is the longer one from ~1300 x 1300 on; the complex kernel needs its 130)
#ifndef JAC_MIN_BLOCKS
#define JAC_MIN_BLOCKS(REAL) ((REAL) ? 6 : 1)
#endif
// One launch per round: pair workgroups of `round` first (they are the critical path), then the update
// workgroups of `prev_round`.  flags bit 0: pair part present, bit 1: a previous round is pending.
template <bool REAL>
__global__ __launch_bounds__(256, JAC_MIN_BLOCKS(REAL)) void k_jacobi_round(int n, int nb, int round, int prev_round, int flags,
                                                      const typename JacEl<REAL>::T* __restrict__ Win,
                                                      typename JacEl<REAL>::T* __restrict__ Wout, int64_t lda,
                                                      typename JacEl<REAL>::T* __restrict__ V, int64_t ldv,
                                                      const typename JacEl<REAL>::T* __restrict__ Uprev,
                                                      typename JacEl<REAL>::T* __restrict__ Uout, int ntiles,
                                                      int vblocks) {
    __shared__ typename JacEl<REAL>::T S[2][J2B][J2B + 1];
    __shared__ typename JacEl<REAL>::T U[J2B][J2B + 1];
    const int npair_wg = (flags & 1) ? nb / 2 : 0;
    const int seq = flags >> 8;
    const bool is_pair = (int)blockIdx.x < npair_wg;
    const bool sampled = (flags & 4) && threadIdx.x == 0 && seq < 1024 && (is_pair || ((blockIdx.x - npair_wg) & 15) == 0);
    if (sampled) {
        const unsigned long long t = wall_clock64();
        atomicMin(&g_jac_span[4 * seq], t);
        if (is_pair && blockIdx.x < 64) g_jac_pair[((size_t)seq * 64 + blockIdx.x) * 2] = t;
    }
    struct SpanEnd {
        int cell, pcell;
        bool on;
        __device__ ~SpanEnd() {
            if (on) {
                const unsigned long long t = wall_clock64();
                atomicMax(&g_jac_span[cell], t);
                if (pcell >= 0) g_jac_pair[pcell] = t;
            }
        }
    } span_end{4 * seq + (is_pair ? 1 : 2), (is_pair && blockIdx.x < 64) ? (int)(((size_t)seq * 64 + blockIdx.x) * 2 + 1) : -1, sampled};
    if ((int)blockIdx.x < npair_wg) {
        jacobi_pair_part<REAL>(blockIdx.x, nb, round, prev_round, (flags & 2) != 0, Win, lda, Uprev, Uout, S, U, (flags & 4) != 0);
    } else {
        const bool stamp = (flags & 4) && (int)blockIdx.x == npair_wg && threadIdx.x == 0;
        long long w0 = 0;
        if (stamp) w0 = wall_clock64();
        jacobi_update_part<REAL>(blockIdx.x - npair_wg, n, nb, prev_round, Win, Wout, lda, V, ldv, Uprev, ntiles, vblocks,
                           S[0]);
        if (stamp) atomicAdd(&g_jac_clk[4], (unsigned long long)(wall_clock64() - w0));
    }
}

// out[0] = sum |offdiag|^2, out[1] = sum |diag|^2   (whole matrix); out[2 nblocks + block] = sum Im^2
template <typename ET>
__global__ __launch_bounds__(256) void k_offdiag_norm(int n, const ET* __restrict__ A, int64_t lda,
                                                      double* __restrict__ out) {
    __shared__ double sh[4];
    double off = 0.0, dg = 0.0, im2 = 0.0;
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e / n), r = (int)(e - (int64_t)c * n);
        const cd v = jac_ld(A[r + (int64_t)c * lda]);
        const double a2 = v.x * v.x + v.y * v.y;
        im2 += v.y * v.y;
        if (r == c)
            dg += a2;
        else
            off += a2;
    }
    const double o = block_sum256(off, sh);
    const double d = block_sum256(dg, sh);
    const double i2 = block_sum256(im2, sh);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = o;
        out[2 * blockIdx.x + 1] = d;
        out[2 * gridDim.x + blockIdx.x] = i2;
    }
}

template <typename ET>
__global__ void k_set_identity(int n, ET* __restrict__ V, int64_t ldv) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    const int j = (int)(idx / n), i = (int)(idx - (int64_t)j * n);
    jac_st(V[i + (int64_t)j * ldv], i == j ? 1.0 : 0.0, 0.0);
}

// copy A (n x n) into the padded work matrix W (np x np), padding diagonal with `big` values
template <typename ET>
__global__ void k_pad_matrix(int n, int np, const cd* __restrict__ A, int64_t lda, ET* __restrict__ W, double big) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)np * np) return;
    const int j = (int)(idx / np), i = (int)(idx - (int64_t)j * np);
    cd v = make_double2(0.0, 0.0);
    if (i < n && j < n)
        v = A[i + (int64_t)j * lda];
    else if (i == j)
        v = make_double2(big * (1.0 + 1e-3 * (i - n)), 0.0);
    jac_st(W[idx], v.x, v.y);
}

template <typename ET>
__global__ void k_extract_diag(int n, const ET* __restrict__ W, int64_t ldw, double* __restrict__ d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = jac_ld(W[i + (int64_t)i * ldw]).x;
}
// eigenvector columns of the real path back into the caller's complex array, sorted
__global__ void k_gather_cols_real(int64_t n, const double* __restrict__ X, int64_t ldx, const int* __restrict__ perm,
                                   cd* __restrict__ Y, int64_t ldy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (i < n) Y[(int64_t)c * ldy + i] = make_double2(X[(int64_t)perm[c] * ldx + i], 0.0);
}

// ======================================================================================== host side
static int dws_ensure(dftk_mi_basis* b, void** buf, size_t* cur, size_t bytes) {
    if (bytes <= *cur) return 0;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (*buf) HIPCHK(hipFree(*buf));
    *buf = nullptr;
    *cur = 0;
    HIPCHK(dftk_scratch_malloc(buf, bytes));
    *cur = bytes;
    return 0;
}

// device -> pinned host words (host_fetch): the destination is host memory mapped into the device's address space
__global__ void k_fetch_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
int host_fetch(dftk_mi_basis* b, void* dst_h, const void* src_d, size_t bytes) {
    if (bytes == 0) return host_wait(b);
    // (A/B against the runtime's blit copy: profiles/r04_late_step_cfg5_{zero_copy,blit}_fetch.txt)
    if (bytes > HOST_FETCH_BYTES || (bytes & 3) || (reinterpret_cast<uintptr_t>(src_d) & 3) || !b->h_fetch) {
        HIPCHK(hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, b->stream));
        return host_wait(b);
    }
    const size_t words = bytes / 4;
    const unsigned blocks = (unsigned)std::min<size_t>(32, (words + 255) / 256);
    hipLaunchKernelGGL(k_fetch_words, dim3(blocks), dim3(256), 0, b->stream, reinterpret_cast<const uint32_t*>(src_d),
                       reinterpret_cast<uint32_t*>(b->h_fetch), words);
    HIPCHK(hipGetLastError());
    CHK(host_wait(b));
    std::memcpy(dst_h, b->h_fetch, bytes);
    return 0;
}

static int fetch_scalars(dftk_mi_basis* b, int count) {
    return host_fetch(b, b->h_scalars, b->d_scalars, count * sizeof(double));
}

int dense_potrf_trtri(dftk_mi_basis* b, int n, cd* A, int64_t lda, cd* invR, int64_t ldi, double* normest_R,
                      double* normest_invR, bool real_input) {
    if (batching()) {
        double out[2] = {0.0, 0.0};      // on the fiber's stack: filled by the round's executor
        BOp o;
        o.b = b;
        o.type = BOP_POTRF; o.m = n; o.C = A; o.ldc = lda; o.D = invR; o.ldb = ldi; o.host = out;
        const int st = batch_record_sync(std::move(o));
        if (normest_R) *normest_R = out[0];
        if (normest_invR) *normest_invR = out[1];
        return st;
    }
    const int ps = prof_begin(b, PROF_CHOL, (double)n);
    int* d_info = reinterpret_cast<int*>(b->d_scalars + 200);
    HIPCHK(hipMemsetAsync(d_info, 0, sizeof(int), b->stream));
    // n <= 512 (the Gram matrices of ortho!): ONE cooperative launch instead of 3 launches per 32-wide panel
    static int coop_ok = -1;
    if (coop_ok < 0)
        coop_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_trtri_coop<cd>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) == hipSuccess &&
                  hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_trtri_coop<double>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024) == hipSuccess;
    bool coop = coop_ok == 1 && n <= 512 && n > 32;
    // Every workgroup of the cooperative kernel needs a CU to itself (LDS) and spins until the other block columns of ITS
    // launch have published, so all nb <= 32 of them must be resident together.  Two guards:
    //  * the launch is a COOPERATIVE launch (hipLaunchCooperativeKernel): the runtime refuses a grid that cannot be
    //    co-resident on the device as it is (partitioned / CU-masked devices) and schedules the grid as a whole, also
    //    against the cooperative grids of OTHER processes sharing the device (two ranks on one GPU in the test-suite);
    //    a refused launch falls back to the blocked path below;
    //  * within this process the workgroups in flight over all streams (lanes) are counted -- the call is synchronous,
    //    it ends with a host fetch -- and kept below half the device's compute units.
    // DFTK_MI_POTRF_COOP_LAUNCH=0: plain launch (the round-4 form) under the per-process count only.
    static const bool coop_launch = !(getenv("DFTK_MI_POTRF_COOP_LAUNCH") && atoi(getenv("DFTK_MI_POTRF_COOP_LAUNCH")) == 0);
    // budget and in-flight count PER DEVICE, initialised once under a lock (several host lane threads call this routine;
    // a process may drive several devices)
    const int MAXDEV = 64;
    static std::once_flag coop_once[MAXDEV];
    static int coop_budget_by_dev[MAXDEV];
    static std::atomic<int> coop_in_flight_by_dev[MAXDEV];
    const int dev = (b->device >= 0 && b->device < MAXDEV) ? b->device : 0;
    std::call_once(coop_once[dev], [&]() {
        hipDeviceProp_t prop;
        int coop_attr = 0, budget = 0;
        if (hipGetDeviceProperties(&prop, b->device) == hipSuccess && prop.multiProcessorCount > 0)
            budget = prop.multiProcessorCount / 2;
        if (coop_launch && (hipDeviceGetAttribute(&coop_attr, hipDeviceAttributeCooperativeLaunch, b->device) != hipSuccess ||
                            !coop_attr))
            budget = 0;      // no cooperative launches on this device: blocked path
        coop_budget_by_dev[dev] = budget;
        coop_in_flight_by_dev[dev].store(0);
    });
    const int coop_budget_wgs = coop_budget_by_dev[dev];
    std::atomic<int>& coop_wgs_in_flight = coop_in_flight_by_dev[dev];
    struct CoopBudget {
        std::atomic<int>& c;
        int n = 0;
        ~CoopBudget() { c.fetch_sub(n); }
    } coop_budget{coop_wgs_in_flight};
    if (coop) {
        const int nb = (n + CCB - 1) / CCB;
        if (coop_wgs_in_flight.fetch_add(nb) + nb > coop_budget_wgs) {
            coop_wgs_in_flight.fetch_sub(nb);
            coop = false;
        } else {
            coop_budget.n = nb;
        }
    }
    if (coop) {
        const int nb = (n + CCB - 1) / CCB, np = nb * CCB;
        const size_t need = ((size_t)nb * np * CCB + (size_t)nb * CCB * CCB) * sizeof(cd) + 64 * sizeof(int);
        CHK(dws_ensure(b, &b->dense_ws, &b->dense_ws_bytes, need));
        cd* Lbuf = reinterpret_cast<cd*>(b->dense_ws);
        cd* Wbuf = Lbuf + (size_t)nb * np * CCB;
        int* flags = reinterpret_cast<int*>(Wbuf + (size_t)nb * CCB * CCB);
        HIPCHK(hipMemsetAsync(flags, 0, 64 * sizeof(int), b->stream));
        if (!coop_launch) {
            if (real_input)   // (the buffers are sized for complex elements: the real factorisation uses half of each)
                hipLaunchKernelGGL(k_potrf_trtri_coop<double>, dim3(nb), dim3(CCT), (size_t)np * CCB * sizeof(double), b->stream,
                                   n, A, lda, invR, ldi, reinterpret_cast<double*>(Lbuf), reinterpret_cast<double*>(Wbuf), flags,
                                   d_info);
            else
                hipLaunchKernelGGL(k_potrf_trtri_coop<cd>, dim3(nb), dim3(CCT), (size_t)np * CCB * sizeof(cd), b->stream, n, A,
                                   lda, invR, ldi, Lbuf, Wbuf, flags, d_info);
            HIPCHK(hipGetLastError());
        } else {
            double* Ld = reinterpret_cast<double*>(Lbuf);
            double* Wd = reinterpret_cast<double*>(Wbuf);
            int n_arg = n;
            void* args_r[] = {&n_arg, &A, &lda, &invR, &ldi, &Ld, &Wd, &flags, &d_info};
            void* args_c[] = {&n_arg, &A, &lda, &invR, &ldi, &Lbuf, &Wbuf, &flags, &d_info};
            const hipError_t le =
                real_input ? hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_potrf_trtri_coop<double>), dim3(nb),
                                                        dim3(CCT), args_r, (size_t)np * CCB * sizeof(double), b->stream)
                           : hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_potrf_trtri_coop<cd>), dim3(nb), dim3(CCT),
                                                        args_c, (size_t)np * CCB * sizeof(cd), b->stream);
            if (le != hipSuccess) {       // the grid cannot be co-resident here: the blocked path takes over
                (void)hipGetLastError();
                coop_wgs_in_flight.fetch_sub(coop_budget.n);
                coop_budget.n = 0;
                coop = false;
            }
        }
    }
    for (int j0 = 0; j0 < n && !coop; j0 += PB) {
        const int jb = (n - j0) < PB ? (n - j0) : PB;
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), 0, b->stream, jb, j0, A, lda, d_info);
        const int n2 = n - j0 - jb;
        if (n2 > 0) {
            hipLaunchKernelGGL(k_trsm_row, dim3((n2 + 63) / 64), dim3(64), 0, b->stream, n, j0, A, lda, d_info);
            cd* R12 = A + j0 + (int64_t)(j0 + jb) * lda;
            cd* A22 = A + (j0 + jb) + (int64_t)(j0 + jb) * lda;
            const cd mone = make_double2(-1.0, 0.0), one = make_double2(1.0, 0.0);
            CHK(zgemm(b, 'C', n2, n2, jb, mone, R12, lda, R12, lda, one, A22, lda, /*upper=*/1));   // only the upper triangle is ever read
        }
    }
    if (!coop) {
        const size_t lds = ((size_t)32 * ((n + 31) / 32) * TRI_TC + 32 * 33) * sizeof(cd);
        static int big_lds_ok = -1;   // > 64 KiB of dynamic LDS needs an opt-in
        if (big_lds_ok < 0)
            big_lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_trtri_upper_blk),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
        if (lds <= 64 * 1024 || (big_lds_ok && lds <= 150 * 1024))
            hipLaunchKernelGGL(k_trtri_upper_blk, dim3((n + TRI_TC - 1) / TRI_TC), dim3(256), lds, b->stream, n, A, lda,
                               invR, ldi);
        else
            hipLaunchKernelGGL(k_trtri_upper, dim3(n), dim3(64), (size_t)n * sizeof(cd), b->stream, n, A, lda, invR, ldi);
    }
    hipLaunchKernelGGL(k_normest_upper, dim3(NORMEST_CHUNKS, 2), dim3(256), 0, b->stream, n, A, lda, b->d_scalars, invR,
                       ldi, b->d_scalars + 3 * NORMEST_CHUNKS);
    prof_end(b, ps);
    HIPCHK(hipGetLastError());
    CHK(fetch_scalars(b, 208));
    int info;
    std::memcpy(&info, (const void*)(b->h_scalars + 200), sizeof(int));
    double est[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};   // max |diag|, sum |offdiag|^2, non-finite flag
    for (int mtx = 0; mtx < 2; ++mtx)
        for (int ch = 0; ch < NORMEST_CHUNKS; ++ch) {
            const double* h = b->h_scalars + 3 * (NORMEST_CHUNKS * mtx + ch);
            est[mtx][0] = std::max(est[mtx][0], h[0]);
            est[mtx][1] += h[1];
            est[mtx][2] += h[2];
        }
    if (info != 0 || est[0][2] != 0.0 || est[1][2] != 0.0) return DFTK_MI_NUM_CHOLESKY;
    if (normest_R) *normest_R = est[0][0] + sqrt(est[0][1]);
    if (normest_invR) *normest_invR = est[1][0] + sqrt(est[1][1]);
    return 0;
}

// Host-only view of the Jacobi schedule (CPU test-suite): number of 16-wide blocks for an n x n matrix and, for
// `round` in [-1, nb - 2], the block pairs (p_k, q_k) of that round plus, for every block, its (pair, member)
// as the look-ahead pair solve finds them.
int jacobi_schedule_host(int n, int round, int* nb_out, int* pairs, int* where) {
    if (n <= 0 || !nb_out) return DFTK_MI_EINVAL;
    int nb = (n + JB - 1) / JB;
    if (nb % 2) nb += 1;
    if (nb < 2) nb = 2;
    *nb_out = nb;
    if (!pairs && !where) return 0;
    if (round < -1 || round > nb - 2) return DFTK_MI_EINVAL;
    if (pairs)
        for (int k = 0; k < nb / 2; ++k) tournament_pair(nb, round, k, pairs[2 * k], pairs[2 * k + 1]);
    if (where)
        for (int blk = 0; blk < nb; ++blk) tournament_find(nb, round, blk, where[2 * blk], where[2 * blk + 1]);
    return 0;
}

// REAL = true: real symmetric input (every imaginary part exactly zero); the work matrices are plain doubles
template <bool REAL>
static int heev_impl(dftk_mi_basis* b, int n, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv, double off2,
                     double dg2) {
    typedef typename JacEl<REAL>::T ET;
    int nb = (n + JB - 1) / JB;
    if (nb % 2) nb += 1;
    if (nb < 2) nb = 2;
    const int np = nb * JB;
    const int npairs = nb / 2;
    // workspace: W, W2 (np x np, ping-pong), Vw (np x np), 2 x Ubuf (npairs x 2b x 2b), diag (np doubles), perm (np ints)
    // (sized for complex elements; the real path uses half of each array)
    const size_t szW = (size_t)np * np * sizeof(cd);
    const size_t szU = (size_t)npairs * J2B * J2B * sizeof(cd);
    const size_t total = 3 * szW + 2 * szU + (size_t)np * (sizeof(double) + sizeof(int)) + 4096 * sizeof(double);
    CHK(dws_ensure(b, &b->dense_ws, &b->dense_ws_bytes, total));   // per basis: one stream, one device
    char* base = reinterpret_cast<char*>(b->dense_ws);
    ET* W = reinterpret_cast<ET*>(base);
    ET* Vw = reinterpret_cast<ET*>(base + szW);
    ET* Wb[2] = {W, reinterpret_cast<ET*>(base + 2 * szW)};
    ET* Ub[2] = {reinterpret_cast<ET*>(base + 3 * szW), reinterpret_cast<ET*>(base + 3 * szW + szU)};
    double* d_diag = reinterpret_cast<double*>(base + 3 * szW + 2 * szU);
    int* d_perm = reinterpret_cast<int*>(d_diag + np);
    double* d_red = reinterpret_cast<double*>(d_perm + np + (np & 1));
    const int redblocks = 64;
    std::vector<double> hred(3 * redblocks);

    // scale for the padding: Gershgorin-like bound from the Frobenius norm
    const double fro = sqrt(off2 + dg2);
    const double big = 2.0 * fro + 1.0;
    hipLaunchKernelGGL(k_pad_matrix<ET>, dim3((unsigned)(((size_t)np * np + 255) / 256)), dim3(256), 0, b->stream, n, np,
                       A, lda, W, big);
    hipLaunchKernelGGL(k_set_identity<ET>, dim3((unsigned)(((size_t)np * np + 255) / 256)), dim3(256), 0, b->stream, np,
                       Vw, (int64_t)np);
    // off-diagonal Frobenius norm relative to ||A||_F; the round-off floor of the blocked sweeps
    // grows like eps*sqrt(n), so accept 1e-14 outright or a stagnated sweep below 1e-12.  Above n = 90 the tolerance is
    // n eps / 2 (1.1e-13 at n = 1006): the backward error LAPACK itself guarantees is O(n eps ||A||), the sweeps
    // converge quadratically here (3e-13 -> 1e-16 in the next one), and a sweep of a 1509^2 matrix is 2.5 ms
    const double tol = std::max(1e-14, 0.5 * n * 2.220446049250313e-16);
    double prev_off = -1.0;
    int sweep = 0;
    const int maxsweeps = 40;
    bool done = (off2 <= tol * tol * (dg2 + off2)) && off2 == 0.0;
    bool skip_next_check = off2 > 1e-4 * (dg2 + off2);   // the initial matrix is already measured
    // software pipeline over rounds: launch i = pair part of round i + update part of round i-1
    const int ntiles = npairs * (npairs + 1) / 2, vblocks = (np / 16 + 3) / 4;
    int cur = 0, uw = 0, pending_round = 0;
    bool pending = false;
    static const bool clk = getenv("DFTK_MI_HEEV_CLOCK") != nullptr;
    int clk_seq = 0;
    if (clk) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<unsigned long long> sp(4096);
        for (int i = 0; i < 1024; ++i) {
            sp[4 * i] = ~0ull;
            sp[4 * i + 1] = sp[4 * i + 2] = sp[4 * i + 3] = 0ull;
        }
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_jac_clk), z, sizeof(z)));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_jac_span), sp.data(), sp.size() * sizeof(unsigned long long)));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    auto launch_round = [&](bool do_pair, int round) {
        const int flags = (do_pair ? 1 : 0) | (pending ? 2 : 0) | (clk ? 4 | (clk_seq++ << 8) : 0);
        const int grid = (do_pair ? npairs : 0) + (pending ? ntiles + npairs * vblocks : 0);
        if (grid == 0) return;
        hipLaunchKernelGGL(k_jacobi_round<REAL>, dim3(grid), dim3(256), 0, b->stream, np, nb, round, pending_round, flags,
                           Wb[cur], Wb[cur ^ 1], (int64_t)np, Vw, (int64_t)np, Ub[uw ^ 1], Ub[uw], ntiles, vblocks);
        if (pending) cur ^= 1;      // the update of the pending round has been written to the other copy
        pending = do_pair;
        if (do_pair) {
            pending_round = round;
            uw ^= 1;
        }
    };
    for (; sweep < maxsweeps && !done; ++sweep) {
        // round -1: within-block rotations on the block pairs (2k, 2k+1); rounds 0 .. nb-2: the tournament
        for (int round = -1; round < nb - 1; ++round) launch_round(true, round);
        // the sweeps converge linearly down to ~1e-3 and quadratically from there: after a check that saw more
        // than 1e-2 the next sweep cannot reach 1e-14, so its check (a host synchronisation) is skipped
        if (skip_next_check && sweep + 1 < maxsweeps) {
            skip_next_check = false;
            continue;
        }
        launch_round(false, 0);   // drain the pipeline: the check needs the matrix after the last round
        hipLaunchKernelGGL(k_offdiag_norm<ET>, dim3(redblocks), dim3(256), 0, b->stream, np, Wb[cur], (int64_t)np, d_red);
        CHK(host_fetch(b, hred.data(), d_red, hred.size() * sizeof(double)));
        double o2 = 0.0;
        for (int i = 0; i < redblocks; ++i) o2 += hred[2 * i];
        if (!std::isfinite(o2)) return DFTK_MI_NUM_NONFINITE;
        const double off = sqrt(o2);
        static const bool trace = getenv("DFTK_MI_HEEV_TRACE") != nullptr;   // convergence history per sweep
        if (trace) fprintf(stderr, "[heev%s] n=%d sweep %d off/fro=%.3e\n", REAL ? " real" : "", n, sweep + 1, off / fro);
        if (off <= tol * fro) done = true;
        if (!done && prev_off >= 0.0 && off > 0.5 * prev_off && off <= 1e-12 * fro) done = true;
        prev_off = off;
        skip_next_check = off > 1e-2 * fro;
    }
    HIPCHK(hipGetLastError());
    if (clk) {
        unsigned long long h[8];
        HIPCHK(hipStreamSynchronize(b->stream));
        HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_jac_clk), sizeof(h)));
        const double nl = h[0] ? (double)h[0] : 1.0;
        fprintf(stderr, "[heev clock] n=%d sweeps=%d launches=%llu  per launch (us): setup %.2f  inner rounds %.2f (%.0f shader "
                        "cycles)  U store %.2f | update workgroup %.2f\n", n, sweep, h[0], h[1] / nl * 0.01, h[2] / nl * 0.01,
                h[5] / nl, h[3] / nl * 0.01, h[4] / nl * 0.01);
        std::vector<unsigned long long> sp(4096);
        HIPCHK(hipMemcpyFromSymbol(sp.data(), HIP_SYMBOL(g_jac_span), sp.size() * sizeof(unsigned long long)));
        const int nsp = clk_seq < 1024 ? clk_seq : 1024;
        double pe = 0.0, ue = 0.0, gap = 0.0;
        int ng = 0, np_ = 0, nu = 0;
        for (int i = 0; i < nsp; ++i) {
            const unsigned long long t0 = sp[4 * i], tp = sp[4 * i + 1], tu = sp[4 * i + 2];
            if (tp) {
                pe += (double)(tp - t0);
                ++np_;
            }
            if (tu) {
                ue += (double)(tu - t0);
                ++nu;
            }
            const unsigned long long prev_end = i > 0 ? std::max(sp[4 * i - 3], sp[4 * i - 2]) : 0ull;
            if (i > 0 && t0 > prev_end) {
                gap += (double)(t0 - prev_end);
                ++ng;
            }
        }
        if (nsp > 1 && nb / 2 <= 64) {
            std::vector<unsigned long long> pr((size_t)1024 * 64 * 2);
            HIPCHK(hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_jac_pair), pr.size() * sizeof(unsigned long long)));
            const int npw = nb / 2;
            double ent_avg = 0.0, ent_max = 0.0, dur_avg = 0.0, dur_max = 0.0;
            int cnt = 0;
            for (int i = 1; i < nsp; ++i) {   // launches with a pair part
                double emax = 0.0, dmax = 0.0;
                bool ok = true;
                for (int q = 0; q < npw; ++q) {
                    const unsigned long long e = pr[((size_t)i * 64 + q) * 2], x = pr[((size_t)i * 64 + q) * 2 + 1];
                    if (e < sp[4 * i] || x < e) {
                        ok = false;
                        break;
                    }
                    ent_avg += (double)(e - sp[4 * i]);
                    dur_avg += (double)(x - e);
                    emax = std::max(emax, (double)(e - sp[4 * i]));
                    dmax = std::max(dmax, (double)(x - e));
                    ++cnt;
                }
                if (ok) {
                    ent_max += emax;
                    dur_max += dmax;
                }
            }
            if (cnt)
                fprintf(stderr, "[heev clock] pair workgroups: entry after the launch's first entry avg %.2f us (latest %.2f), entry -> "
                                "exit avg %.2f us (slowest %.2f)\n", ent_avg / cnt * 0.01, ent_max / (nsp - 1) * 0.01,
                        dur_avg / cnt * 0.01, dur_max / (nsp - 1) * 0.01);
        }
        if (nsp > 1)
            fprintf(stderr, "[heev clock] from the first workgroup entry of a launch: last pair workgroup exit %.2f us, last (sampled) "
                            "update workgroup exit %.2f us; last exit -> first entry of the next launch %.2f us; launch period "
                            "%.2f us\n", np_ ? pe / np_ * 0.01 : 0.0, nu ? ue / nu * 0.01 : 0.0, ng ? gap / ng * 0.01 : 0.0,
                    (double)(sp[4 * (nsp - 1)] - sp[0]) / (nsp - 1) * 0.01);
    }
    if (!done) {
        dftk_set_error("dense_heev: Jacobi did not converge in %d sweeps (n=%d)", maxsweeps, n);
        return DFTK_MI_NUM_EIGEN;
    }
    // eigenvalues = diag(W); sort ascending on the host, gather eigenvector columns
    hipLaunchKernelGGL(k_extract_diag<ET>, dim3((np + 255) / 256), dim3(256), 0, b->stream, np, Wb[cur], (int64_t)np, d_diag);
    std::vector<double> diag(np);
    CHK(host_fetch(b, diag.data(), d_diag, np * sizeof(double)));
    std::vector<int> perm(np);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int c) { return diag[a] < diag[c]; });
    for (int i = 0; i < n; ++i) W_h[i] = diag[perm[i]];
    HIPCHK(hipMemcpyAsync(d_perm, perm.data(), n * sizeof(int), hipMemcpyHostToDevice, b->stream));
    if constexpr (REAL)
        hipLaunchKernelGGL(k_gather_cols_real, dim3((n + 255) / 256, n), dim3(256), 0, b->stream, (int64_t)n, Vw,
                           (int64_t)np, d_perm, V, ldv);
    else
        hipLaunchKernelGGL(k_gather_cols, dim3((n + EW_ROWS - 1) / EW_ROWS, n), dim3(256), 0, b->stream, (int64_t)n, Vw,
                           (int64_t)np, d_perm, V, ldv);
    HIPCHK(hipGetLastError());
    CHK(host_wait(b));   // perm (host vector) must outlive the copy
    return 0;
}

int dense_input_norms(dftk_mi_basis* b, int n, const cd* A, int64_t lda, double* off2_out, double* dg2_out, double* im2_out) {
    const int redblocks = 64;
    CHK(dws_ensure(b, &b->dense_ws, &b->dense_ws_bytes, 4096 * sizeof(double)));
    double* d_red = reinterpret_cast<double*>(b->dense_ws);
    hipLaunchKernelGGL(k_offdiag_norm<cd>, dim3(redblocks), dim3(256), 0, b->stream, n, A, lda, d_red);
    std::vector<double> hred(3 * redblocks);
    CHK(host_fetch(b, hred.data(), d_red, hred.size() * sizeof(double)));
    double off2 = 0.0, dg2 = 0.0, im2 = 0.0;
    for (int i = 0; i < redblocks; ++i) {
        off2 += hred[2 * i];
        dg2 += hred[2 * i + 1];
        im2 += hred[2 * redblocks + i];
    }
    *off2_out = off2;
    *dg2_out = dg2;
    *im2_out = im2;
    return 0;
}

int dense_heev(dftk_mi_basis* b, int n, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv) {
    if (n <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_HEEV; o.m = n; o.C = A; o.ldc = lda; o.D = V; o.ldb = ldv; o.host = W_h;
        return batch_record_sync(std::move(o));
    }
    ProfScope scope(b, PROF_HEEV, (double)n);
    return dense_heev_full(b, n, A, lda, W_h, V, ldv);
}

int dense_heev_full(dftk_mi_basis* b, int n, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv) {
    if (n <= 0) return 0;
    // norms of the input (scale of the padding, early exit for a diagonal matrix) and sum Im^2
    double off2 = 0.0, dg2 = 0.0, im2 = 0.0;
    CHK(dense_input_norms(b, n, A, lda, &off2, &dg2, &im2));
    if (!std::isfinite(off2 + dg2)) return DFTK_MI_NUM_NONFINITE;
    // every imaginary part exactly zero (real symmetric input, e.g. the Rayleigh-Ritz matrices of the Gamma-real
    // LOBPCG): real rotations on plain-double work matrices -- a quarter of the matrix-core work and half the bytes
    // per round; the eigenvectors come back with exact zeros in their imaginary parts
    if (im2 == 0.0) return heev_impl<true>(b, n, A, lda, W_h, V, ldv, off2, dg2);
    return heev_impl<false>(b, n, A, lda, W_h, V, ldv, off2, dg2);
}

int apply_D(dftk_mi_kblock* kb, int n_bands, const cd* X, cd* Y) {
    dftk_mi_basis* b = kb->basis;
    const int64_t total = (int64_t)kb->n_p * n_bands;
    hipLaunchKernelGGL(k_apply_D, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, b->stream, kb->n_p, n_bands,
                       kb->D_bw, kb->d_D, X, Y);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- thin launch wrappers -------------------------------------------------------------------
int ew_colnorms(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, double* out_d) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_COLRED; o.mode = 0; o.n = n; o.m = m; o.A = X; o.lda = ldx; o.C = out_d;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 16.0 * (double)n * m : 0.0);
    EW_LAUNCH_COLS(k_col_reduce, n, m, b->stream, 0, n, X, ldx, (const cd*)nullptr, (int64_t)0, (const double*)nullptr, out_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_coldots(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const cd* Y, int64_t ldy,
               double* out_re_d) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_COLRED; o.mode = 1; o.n = n; o.m = m; o.A = X; o.lda = ldx; o.B = Y; o.ldb = ldy; o.C = out_re_d;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 32.0 * (double)n * m : 0.0);
    EW_LAUNCH_COLS(k_col_reduce, n, m, b->stream, 1, n, X, ldx, Y, ldy, (const double*)nullptr, out_re_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_coldots_im(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const cd* Y, int64_t ldy,
                  double* out_im_d) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_COLRED; o.mode = 4; o.n = n; o.m = m; o.A = X; o.lda = ldx; o.B = Y; o.ldb = ldy; o.C = out_im_d;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 32.0 * (double)n * m : 0.0);
    EW_LAUNCH_COLS(k_col_reduce, n, m, b->stream, 4, n, X, ldx, Y, ldy, (const double*)nullptr, out_im_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_weighted_colsums(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const double* w_d,
                        double* out_d) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_COLRED; o.mode = 2; o.n = n; o.m = m; o.A = X; o.lda = ldx; o.W = w_d; o.C = out_d;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 16.0 * (double)n * m : 0.0);
    EW_LAUNCH_COLS(k_col_reduce, n, m, b->stream, 2, n, X, ldx, (const cd*)nullptr, (int64_t)0, w_d, out_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_conj_transpose(dftk_mi_basis* b, int n, const cd* A, int64_t lda, cd* B, int64_t ldb) {
    if (n <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_CTRANS; o.m = n; o.A = A; o.lda = lda; o.C = B; o.ldc = ldb;
        return batch_record(std::move(o));
    }
    hipLaunchKernelGGL(k_conj_transpose, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, b->stream, n, A,
                       lda, B, ldb);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_square(dftk_mi_basis* b, double* d, size_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_unary, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, 0, d, n);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_sqrt(dftk_mi_basis* b, double* d, size_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_unary, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, 1, d, n);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_frob2(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, double* out_d) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_COLRED; o.mode = 3; o.n = n; o.m = m; o.A = X; o.lda = ldx; o.C = out_d;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 16.0 * (double)n * m : 0.0);
    EW_LAUNCH_COLS(k_col_reduce, n, m, b->stream, 3, n, X, ldx, (const cd*)nullptr, (int64_t)0, (const double*)nullptr, out_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_residual(dftk_mi_basis* b, int64_t n, int m, const cd* AX, int64_t lda, const cd* X, int64_t ldx,
                const double* lam_d, cd* R, int64_t ldr, double* norms_d, const double* kin, double* mean_kin_d,
                double* xx_d) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_RESIDUAL; o.n = n; o.m = m; o.A = AX; o.lda = lda; o.B = X; o.ldb = ldx; o.W = lam_d; o.C = R; o.ldc = ldr;
        o.D = norms_d; o.W2 = kin; o.E = mean_kin_d; o.F = xx_d;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 48.0 * (double)n * m : 0.0);
    EW_LAUNCH_COLS(k_residual, n, m, b->stream, n, AX, lda, X, ldx, lam_d, R, ldr, norms_d, kin, mean_kin_d, xx_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_tpa(dftk_mi_basis* b, int64_t n, int m, const cd* src, int64_t lds, cd* dst, int64_t ldd, const double* kin,
           const double* mean_kin_d, double* norms_d, double default_shift) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_TPA; o.n = n; o.m = m; o.A = src; o.lda = lds; o.C = dst; o.ldc = ldd; o.W = kin; o.W2 = mean_kin_d;
        o.D = norms_d; o.s0 = default_shift;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 32.0 * (double)n * m : 0.0);
    EW_LAUNCH_COLS(k_tpa, n, m, b->stream, n, src, lds, dst, ldd, kin, mean_kin_d, norms_d, default_shift);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_scale_cols(dftk_mi_basis* b, int64_t n, int m, cd* X, int64_t ldx, const double* s_d, bool invert) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_SCALE; o.n = n; o.m = m; o.C = X; o.ldc = ldx; o.W = s_d; o.flags = invert ? 1 : 0;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 32.0 * (double)n * m : 0.0);
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((n + EW_ROWS - 1) / EW_ROWS), m), dim3(256), 0, b->stream, n, m, X, ldx,
                       s_d, invert ? 1 : 0);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_copy(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, cd* Y, int64_t ldy) {
    if (m <= 0 || n <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_COPY; o.n = n; o.m = m; o.A = X; o.lda = ldx; o.C = Y; o.ldc = ldy;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 32.0 * (double)n * m : 0.0);
    hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + EW_ROWS - 1) / EW_ROWS), m), dim3(256), 0, b->stream, n, m, X, ldx, Y, ldy);
    HIPCHK(hipGetLastError());
    return 0;
}
__global__ __launch_bounds__(256) void k_add_cols(int64_t n, int m, const cd* __restrict__ X, int64_t ldx, cd* __restrict__ Y,
                                                  int64_t ldy) {
    const int c = blockIdx.y;
    const cd* x = X + (int64_t)c * ldx;
    cd* y = Y + (int64_t)c * ldy;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const cd a = x[i];
        cd v = y[i];
        v.x += a.x;
        v.y += a.y;
        y[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_sub_real(int64_t n, const double* __restrict__ a, const double* __restrict__ c,
                                                  double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] - c[i];
}
int ew_add(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, cd* Y, int64_t ldy) {
    if (m <= 0 || n <= 0) return 0;
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 48.0 * (double)n * m : 0.0);
    hipLaunchKernelGGL(k_add_cols, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048), m), dim3(256), 0, b->stream, n, m, X, ldx,
                       Y, ldy);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_sub_real(dftk_mi_basis* b, int64_t n, const double* a, const double* c, double* out) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_sub_real, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, b->stream, n, a, c, out);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_fill_zero(dftk_mi_basis* b, cd* X, size_t count) {
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_FILL0; o.C = X; o.bytes = count * sizeof(cd);
        return batch_record(std::move(o));
    }
    HIPCHK(hipMemsetAsync(X, 0, count * sizeof(cd), b->stream));
    return 0;
}
int ew_sub_identity_shifted(dftk_mi_basis* b, int rows, int cols, cd* C, int64_t ldc, int row0) {
    if (cols <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_SUBID; o.n = rows; o.m = cols; o.C = C; o.ldc = ldc; o.i0 = row0;
        return batch_record(std::move(o));
    }
    hipLaunchKernelGGL(k_sub_identity_shifted, dim3((cols + 255) / 256), dim3(256), 0, b->stream, rows, cols, C, ldc,
                       row0);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_gather_cols(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const int* perm_d, cd* Y,
                   int64_t ldy) {
    if (m <= 0) return 0;
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_GATHER; o.n = n; o.m = m; o.A = X; o.lda = ldx; o.W = perm_d; o.C = Y; o.ldc = ldy;
        return batch_record(std::move(o));
    }
    ProfScope prof_scope(b, PROF_EW, n >= 4096 ? 32.0 * (double)n * m : 0.0);
    hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)((n + EW_ROWS - 1) / EW_ROWS), m), dim3(256), 0, b->stream, n, X, ldx,
                       perm_d, Y, ldy);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_add_diag(dftk_mi_basis* b, int n, cd* A, int64_t lda, double shift) {
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_ADDDIAG; o.m = n; o.C = A; o.ldc = lda; o.s0 = shift;
        return batch_record(std::move(o));
    }
    hipLaunchKernelGGL(k_add_diag, dim3((n + 255) / 256), dim3(256), 0, b->stream, n, A, lda, shift);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_hermitize_upper(dftk_mi_basis* b, int n, cd* A, int64_t lda) {
    if (batching()) {
        BOp o;
        o.b = b;
        o.type = BOP_HERMIT; o.m = n; o.C = A; o.ldc = lda;
        return batch_record(std::move(o));
    }
    hipLaunchKernelGGL(k_hermitize_upper, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, b->stream, n,
                       A, lda);
    HIPCHK(hipGetLastError());
    return 0;
}
