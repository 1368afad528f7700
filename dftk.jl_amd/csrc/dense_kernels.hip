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
#include <algorithm>
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
// One workgroup per column; deterministic (fixed strides and tree).
// mode 0: out[c] = sqrt(sum |X|^2) ; 1: out[c] = Re sum conj(X) Y ; 2: out[c] = sum w |X|^2 ; 3: sum |X|^2
__global__ __launch_bounds__(256) void k_col_reduce(int mode, int64_t n, const cd* __restrict__ X, int64_t ldx,
                                                    const cd* __restrict__ Y, int64_t ldy,
                                                    const double* __restrict__ w, double* __restrict__ out) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    const cd* x = X + (int64_t)c * ldx;
    const cd* y = Y ? Y + (int64_t)c * ldy : nullptr;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const cd a = x[i];
        if (mode == 1) {
            const cd bb = y[i];
            acc += a.x * bb.x + a.y * bb.y;
        } else if (mode == 2) {
            acc += w[i] * (a.x * a.x + a.y * a.y);
        } else {
            acc += a.x * a.x + a.y * a.y;
        }
    }
    const double r = block_sum256(acc, sh);
    if (threadIdx.x == 0) out[c] = (mode == 0) ? sqrt(r) : r;
}

// R = AX - X * lam ; norms[c] = ||R[:,c]||
__global__ __launch_bounds__(256) void k_residual(int64_t n, const cd* __restrict__ AX, int64_t lda,
                                                  const cd* __restrict__ X, int64_t ldx,
                                                  const double* __restrict__ lam, cd* __restrict__ R, int64_t ldr,
                                                  double* __restrict__ norms) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    const double l = lam[c];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const cd a = AX[(int64_t)c * lda + i];
        const cd x = X[(int64_t)c * ldx + i];
        const cd r = make_double2(a.x - l * x.x, a.y - l * x.y);
        R[(int64_t)c * ldr + i] = r;
        acc += r.x * r.x + r.y * r.y;
    }
    const double s = block_sum256(acc, sh);
    if (threadIdx.x == 0) norms[c] = sqrt(s);
}

// R[:,c] *= mean_kin[c] / (mean_kin[c] + kin)
__global__ void k_tpa(int64_t n, int m, cd* __restrict__ R, int64_t ldr, const double* __restrict__ kin,
                      const double* __restrict__ mean_kin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double k = kin[i];
    for (int c = 0; c < m; ++c) {
        const double mk = mean_kin[c];
        const double f = mk / (mk + k);
        cd r = R[(int64_t)c * ldr + i];
        r.x *= f;
        r.y *= f;
        R[(int64_t)c * ldr + i] = r;
    }
}

__global__ void k_scale_cols(int64_t n, int m, cd* __restrict__ X, int64_t ldx, const double* __restrict__ s,
                             int invert) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < m; ++c) {
        const double f = invert ? 1.0 / s[c] : s[c];
        cd v = X[(int64_t)c * ldx + i];
        v.x *= f;
        v.y *= f;
        X[(int64_t)c * ldx + i] = v;
    }
}

__global__ void k_copy(int64_t n, int m, const cd* __restrict__ X, int64_t ldx, cd* __restrict__ Y, int64_t ldy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (i < n) Y[(int64_t)c * ldy + i] = X[(int64_t)c * ldx + i];
}

__global__ void k_gather_cols(int64_t n, const cd* __restrict__ X, int64_t ldx, const int* __restrict__ perm,
                              cd* __restrict__ Y, int64_t ldy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (i < n) Y[(int64_t)c * ldy + i] = X[(int64_t)perm[c] * ldx + i];
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

__global__ __launch_bounds__(256) void k_has_nonfinite(int64_t n, const cd* __restrict__ X, int64_t ldx,
                                                       double* __restrict__ out) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double bad = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const cd a = X[(int64_t)c * ldx + i];
        if (!(isfinite(a.x) && isfinite(a.y))) bad = 1.0;
    }
    const double r = block_sum256(bad, sh);
    if (threadIdx.x == 0) out[c] = r;
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
// Right-looking unblocked upper Cholesky by ONE workgroup of 1024 threads working in global
// memory (the matrix is L2 resident).  info[0] = 0 on success, else 1-based failing column.
__global__ __launch_bounds__(1024) void k_potrf_upper(int n, cd* __restrict__ A, int64_t lda, int* __restrict__ info) {
    __shared__ double s_d;
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        if (tid == 0) {
            const double d = A[j + (int64_t)j * lda].x;
            if (!(d > 0.0) || !isfinite(d)) {
                s_fail = j + 1;
                s_d = 1.0;
            } else {
                s_d = sqrt(d);
            }
            A[j + (int64_t)j * lda] = make_double2(s_d, 0.0);
        }
        __syncthreads();
        if (s_fail) break;
        const double inv = 1.0 / s_d;
        for (int c = j + 1 + tid; c < n; c += 1024) {
            cd v = A[j + (int64_t)c * lda];
            v.x *= inv;
            v.y *= inv;
            A[j + (int64_t)c * lda] = v;
        }
        __syncthreads();
        // trailing update of the upper triangle: A[r,c] -= conj(R[j,r]) * R[j,c], j < r <= c
        const int t = n - j - 1;
        const int64_t total = (int64_t)t * t;
        for (int64_t e = tid; e < total; e += 1024) {
            const int cc = (int)(e / t), rr = (int)(e - (int64_t)cc * t);
            if (rr > cc) continue;
            const int r = j + 1 + rr, c = j + 1 + cc;
            const cd a = A[j + (int64_t)r * lda];
            const cd bb = A[j + (int64_t)c * lda];
            cd v = A[r + (int64_t)c * lda];
            v.x -= a.x * bb.x + a.y * bb.y;
            v.y -= a.x * bb.y - a.y * bb.x;
            A[r + (int64_t)c * lda] = v;
        }
        __syncthreads();
    }
    if (tid == 0) info[0] = s_fail;
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

// out[0] = max |diag|, out[1] = sum of |offdiag|^2 over the upper triangle, out[2] = non-finite flag
__global__ __launch_bounds__(256) void k_normest_upper(int n, const cd* __restrict__ M, int64_t ldm,
                                                       double* __restrict__ out) {
    __shared__ double sh[4];
    __shared__ double smax[256];
    double off = 0.0, mx = 0.0, bad = 0.0;
    const int64_t total = (int64_t)n * n;
    for (int64_t e = threadIdx.x; e < total; e += 256) {
        const int c = (int)(e / n), r = (int)(e - (int64_t)c * n);
        if (r > c) continue;
        const cd v = M[r + (int64_t)c * ldm];
        if (!(isfinite(v.x) && isfinite(v.y))) bad = 1.0;
        const double a2 = v.x * v.x + v.y * v.y;
        if (r == c)
            mx = fmax(mx, sqrt(a2));
        else
            off += a2;
    }
    smax[threadIdx.x] = mx;
    __syncthreads();
    const double o = block_sum256(off, sh);
    const double bsum = block_sum256(bad, sh);
    if (threadIdx.x == 0) {
        double m2 = 0.0;
        for (int i = 0; i < 256; ++i) m2 = fmax(m2, smax[i]);
        out[0] = m2;
        out[1] = o;
        out[2] = bsum;
    }
}

// ---------------------------------------------------------------------------- Hermitian eigensolver
// Blocked two-sided Jacobi with round-robin (tournament) ordering.  Block size JB; per round the
// n/(2 JB) disjoint block pairs are (1) diagonalised approximately in LDS by one cyclic Jacobi
// sweep (k_jacobi_pair), (2) the rotations are applied to the columns of A and V and (3) to the
// rows of A.
#define JB 16
#define J2B (2 * JB)

__device__ __forceinline__ void tournament_pair(int nb, int round, int k, int& p, int& q) {
    // nb even, round in [0, nb-1), k in [0, nb/2): pair k of the round
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

__global__ __launch_bounds__(256) void k_jacobi_pair(int n, int nb, int round, cd* __restrict__ A, int64_t lda,
                                                     cd* __restrict__ Ubuf, int inner_sweeps) {
    __shared__ cd S[J2B][J2B + 1];
    __shared__ cd U[J2B][J2B + 1];
    __shared__ cd rot_s[JB];
    __shared__ double rot_c[JB];
    __shared__ int rot_p[JB], rot_q[JB];
    int bp, bq;
    tournament_pair(nb, round, blockIdx.x, bp, bq);
    const int tid = threadIdx.x;
    // load the 2x2 block sub-matrix (global index of local i)
    for (int e = tid; e < J2B * J2B; e += 256) {
        const int c = e / J2B, r = e - c * J2B;
        const int gr = (r < JB ? bp * JB + r : bq * JB + (r - JB));
        const int gc = (c < JB ? bp * JB + c : bq * JB + (c - JB));
        S[r][c] = A[gr + (int64_t)gc * lda];
        U[r][c] = make_double2(r == c ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    for (int sw = 0; sw < inner_sweeps; ++sw) {
        for (int rd = 0; rd < J2B - 1; ++rd) {
            if (tid < JB) {
                int p, q;
                tournament_pair(J2B, rd, tid, p, q);
                const cd beta = S[p][q];
                const double ab = sqrt(beta.x * beta.x + beta.y * beta.y);
                const double al = S[p][p].x, ga = S[q][q].x;
                double c = 1.0;
                cd s = make_double2(0.0, 0.0);
                if (ab > 1e-300 && ab > 1e-18 * sqrt(fabs(al * ga) + 1e-300)) {
                    const double tau = (ga - al) / (2.0 * ab);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    const double f = t * c / ab;
                    s = make_double2(f * beta.x, f * beta.y);
                }
                rot_c[tid] = c;
                rot_s[tid] = s;
                rot_p[tid] = p;
                rot_q[tid] = q;
            }
            __syncthreads();
            // column rotations of S and U: (x_p, x_q) <- (c x_p - conj(s) x_q, s x_p + c x_q)
            for (int e = tid; e < 2 * J2B * JB; e += 256) {
                const int which = e / (J2B * JB);
                const int rem = e - which * (J2B * JB);
                const int k = rem / J2B, r = rem - k * J2B;
                const int p = rot_p[k], q = rot_q[k];
                const double c = rot_c[k];
                const cd s = rot_s[k];
                cd(*Mx)[J2B + 1] = which ? U : S;
                const cd xp = Mx[r][p], xq = Mx[r][q];
                // conj(s) * xq = (s.x xq.x + s.y xq.y, s.x xq.y - s.y xq.x)
                Mx[r][p] = make_double2(c * xp.x - (s.x * xq.x + s.y * xq.y), c * xp.y - (s.x * xq.y - s.y * xq.x));
                Mx[r][q] = make_double2(s.x * xp.x - s.y * xp.y + c * xq.x, s.x * xp.y + s.y * xp.x + c * xq.y);
            }
            __syncthreads();
            // row rotations of S: (row_p, row_q) <- (c row_p - s row_q, conj(s) row_p + c row_q)
            for (int e = tid; e < J2B * JB; e += 256) {
                const int k = e / J2B, cidx = e - k * J2B;
                const int p = rot_p[k], q = rot_q[k];
                const double c = rot_c[k];
                const cd s = rot_s[k];
                const cd xp = S[p][cidx], xq = S[q][cidx];
                S[p][cidx] = make_double2(c * xp.x - (s.x * xq.x - s.y * xq.y), c * xp.y - (s.x * xq.y + s.y * xq.x));
                S[q][cidx] = make_double2(s.x * xp.x + s.y * xp.y + c * xq.x, s.x * xp.y - s.y * xp.x + c * xq.y);
            }
            __syncthreads();
        }
    }
    cd* Uo = Ubuf + (int64_t)blockIdx.x * J2B * J2B;
    for (int e = tid; e < J2B * J2B; e += 256) {
        const int c = e / J2B, r = e - c * J2B;
        Uo[e] = U[r][c];   // column-major 2b x 2b
    }
}

// columns: M[:, cols(pair)] <- M[:, cols(pair)] * U  for M = A (rows [0,n)) and V (rows [0,n))
// grid (npairs, ceil(2n / 256)); thread = one row of the stacked [A; V]
__global__ __launch_bounds__(256) void k_jacobi_cols(int n, int nb, int round, cd* __restrict__ A, int64_t lda,
                                                     cd* __restrict__ V, int64_t ldv, const cd* __restrict__ Ubuf) {
    __shared__ cd U[J2B * J2B];
    int bp, bq;
    tournament_pair(nb, round, blockIdx.x, bp, bq);
    const cd* Ui = Ubuf + (int64_t)blockIdx.x * J2B * J2B;
    for (int e = threadIdx.x; e < J2B * J2B; e += 256) U[e] = Ui[e];
    __syncthreads();
    const int row = blockIdx.y * 256 + threadIdx.x;
    if (row >= 2 * n) return;
    cd* M = row < n ? A : V;
    const int64_t ld = row < n ? lda : ldv;
    const int r = row < n ? row : row - n;
    cd x[J2B];
#pragma unroll
    for (int k = 0; k < J2B; ++k) {
        const int gc = (k < JB ? bp * JB + k : bq * JB + (k - JB));
        x[k] = M[r + (int64_t)gc * ld];
    }
#pragma unroll 4
    for (int c = 0; c < J2B; ++c) {
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int k = 0; k < J2B; ++k) {
            const cd u = U[k + c * J2B];
            sr += x[k].x * u.x - x[k].y * u.y;
            si += x[k].x * u.y + x[k].y * u.x;
        }
        const int gc = (c < JB ? bp * JB + c : bq * JB + (c - JB));
        M[r + (int64_t)gc * ld] = make_double2(sr, si);
    }
}

// rows: A[rows(pair), :] <- U^H * A[rows(pair), :]; thread = one column of A
__global__ __launch_bounds__(256) void k_jacobi_rows(int n, int nb, int round, cd* __restrict__ A, int64_t lda,
                                                     const cd* __restrict__ Ubuf) {
    __shared__ cd U[J2B * J2B];
    int bp, bq;
    tournament_pair(nb, round, blockIdx.x, bp, bq);
    const cd* Ui = Ubuf + (int64_t)blockIdx.x * J2B * J2B;
    for (int e = threadIdx.x; e < J2B * J2B; e += 256) U[e] = Ui[e];
    __syncthreads();
    const int col = blockIdx.y * 256 + threadIdx.x;
    if (col >= n) return;
    cd x[J2B];
#pragma unroll
    for (int k = 0; k < J2B; ++k) {
        const int gr = (k < JB ? bp * JB + k : bq * JB + (k - JB));
        x[k] = A[gr + (int64_t)col * lda];
    }
#pragma unroll 4
    for (int r = 0; r < J2B; ++r) {
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int k = 0; k < J2B; ++k) {
            const cd u = U[k + r * J2B];   // conj(U[k, r])
            sr += u.x * x[k].x + u.y * x[k].y;
            si += u.x * x[k].y - u.y * x[k].x;
        }
        const int gr = (r < JB ? bp * JB + r : bq * JB + (r - JB));
        A[gr + (int64_t)col * lda] = make_double2(sr, si);
    }
}

// out[0] = sum |offdiag|^2, out[1] = sum |diag|^2   (whole matrix)
__global__ __launch_bounds__(256) void k_offdiag_norm(int n, const cd* __restrict__ A, int64_t lda,
                                                      double* __restrict__ out) {
    __shared__ double sh[4];
    double off = 0.0, dg = 0.0;
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e / n), r = (int)(e - (int64_t)c * n);
        const cd v = A[r + (int64_t)c * lda];
        const double a2 = v.x * v.x + v.y * v.y;
        if (r == c)
            dg += a2;
        else
            off += a2;
    }
    const double o = block_sum256(off, sh);
    const double d = block_sum256(dg, sh);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = o;
        out[2 * blockIdx.x + 1] = d;
    }
}

__global__ void k_set_identity(int n, cd* __restrict__ V, int64_t ldv) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    const int j = (int)(idx / n), i = (int)(idx - (int64_t)j * n);
    V[i + (int64_t)j * ldv] = make_double2(i == j ? 1.0 : 0.0, 0.0);
}

// copy A (n x n) into the padded work matrix W (np x np), padding diagonal with `big` values
__global__ void k_pad_matrix(int n, int np, const cd* __restrict__ A, int64_t lda, cd* __restrict__ W, double big) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)np * np) return;
    const int j = (int)(idx / np), i = (int)(idx - (int64_t)j * np);
    cd v = make_double2(0.0, 0.0);
    if (i < n && j < n)
        v = A[i + (int64_t)j * lda];
    else if (i == j)
        v = make_double2(big * (1.0 + 1e-3 * (i - n)), 0.0);
    W[idx] = v;
}

__global__ void k_extract_diag(int n, const cd* __restrict__ W, int64_t ldw, double* __restrict__ d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = W[i + (int64_t)i * ldw].x;
}

// ======================================================================================== host side
static int dws_ensure(dftk_mi_basis* b, void** buf, size_t* cur, size_t bytes) {
    if (bytes <= *cur) return 0;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (*buf) HIPCHK(hipFree(*buf));
    *buf = nullptr;
    *cur = 0;
    HIPCHK(hipMalloc(buf, bytes));
    *cur = bytes;
    return 0;
}

// scratch owned by this translation unit (one set per process; calls are serialised per basis stream)
static void* g_dense_ws = nullptr;
static size_t g_dense_ws_bytes = 0;

static int fetch_scalars(dftk_mi_basis* b, int count) {
    HIPCHK(hipMemcpyAsync(b->h_scalars, b->d_scalars, count * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

int dense_potrf_trtri(dftk_mi_basis* b, int n, cd* A, int64_t lda, cd* invR, int64_t ldi, double* normest_R,
                      double* normest_invR) {
    const int ps = prof_begin(b, PROF_CHOL, (double)n);
    int* d_info = reinterpret_cast<int*>(b->d_scalars + 200);
    hipLaunchKernelGGL(k_potrf_upper, dim3(1), dim3(1024), 0, b->stream, n, A, lda, d_info);
    hipLaunchKernelGGL(k_trtri_upper, dim3(n), dim3(64), (size_t)n * sizeof(cd), b->stream, n, A, lda, invR, ldi);
    hipLaunchKernelGGL(k_normest_upper, dim3(1), dim3(256), 0, b->stream, n, A, lda, b->d_scalars);
    hipLaunchKernelGGL(k_normest_upper, dim3(1), dim3(256), 0, b->stream, n, invR, ldi, b->d_scalars + 3);
    prof_end(b, ps);
    HIPCHK(hipGetLastError());
    CHK(fetch_scalars(b, 208));
    int info;
    std::memcpy(&info, (const void*)(b->h_scalars + 200), sizeof(int));
    const double* h = b->h_scalars;
    if (info != 0 || h[2] != 0.0 || h[5] != 0.0) return DFTK_MI_NUM_CHOLESKY;
    if (normest_R) *normest_R = h[0] + sqrt(h[1]);
    if (normest_invR) *normest_invR = h[3] + sqrt(h[4]);
    return 0;
}

int dense_heev(dftk_mi_basis* b, int n, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv) {
    if (n <= 0) return 0;
    const int pslot = prof_begin(b, PROF_HEEV, (double)n);
    struct ProfGuard {
        dftk_mi_basis* b;
        int s;
        ~ProfGuard() { prof_end(b, s); }
    } guard{b, pslot};
    int nb = (n + JB - 1) / JB;
    if (nb % 2) nb += 1;
    if (nb < 2) nb = 2;
    const int np = nb * JB;
    const int npairs = nb / 2;
    // workspace: W (np x np), Vw (np x np), Ubuf (npairs x 2b x 2b), diag (np doubles), perm (np ints)
    const size_t szW = (size_t)np * np * sizeof(cd);
    const size_t szU = (size_t)npairs * J2B * J2B * sizeof(cd);
    const size_t total = 2 * szW + szU + (size_t)np * (sizeof(double) + sizeof(int)) + 4096 * sizeof(double);
    CHK(dws_ensure(b, &g_dense_ws, &g_dense_ws_bytes, total));
    char* base = reinterpret_cast<char*>(g_dense_ws);
    cd* W = reinterpret_cast<cd*>(base);
    cd* Vw = reinterpret_cast<cd*>(base + szW);
    cd* Ubuf = reinterpret_cast<cd*>(base + 2 * szW);
    double* d_diag = reinterpret_cast<double*>(base + 2 * szW + szU);
    int* d_perm = reinterpret_cast<int*>(d_diag + np);
    double* d_red = reinterpret_cast<double*>(d_perm + np + (np & 1));

    // scale for the padding: Gershgorin-like bound from the Frobenius norm
    const int redblocks = 64;
    hipLaunchKernelGGL(k_offdiag_norm, dim3(redblocks), dim3(256), 0, b->stream, n, A, lda, d_red);
    std::vector<double> hred(2 * redblocks);
    HIPCHK(hipMemcpyAsync(hred.data(), d_red, hred.size() * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    double off2 = 0.0, dg2 = 0.0;
    for (int i = 0; i < redblocks; ++i) {
        off2 += hred[2 * i];
        dg2 += hred[2 * i + 1];
    }
    const double fro = sqrt(off2 + dg2);
    if (!std::isfinite(fro)) return DFTK_MI_NUM_NONFINITE;
    const double big = 2.0 * fro + 1.0;
    hipLaunchKernelGGL(k_pad_matrix, dim3((unsigned)(((size_t)np * np + 255) / 256)), dim3(256), 0, b->stream, n, np,
                       A, lda, W, big);
    hipLaunchKernelGGL(k_set_identity, dim3((unsigned)(((size_t)np * np + 255) / 256)), dim3(256), 0, b->stream, np,
                       Vw, (int64_t)np);
    // off-diagonal Frobenius norm relative to ||A||_F; the round-off floor of the blocked sweeps
    // grows like eps*sqrt(n), so accept 1e-14 outright or a stagnated sweep below 1e-12
    const double tol = 1e-14;
    double prev_off = -1.0;
    int sweep = 0;
    const int maxsweeps = 40;
    bool done = (off2 <= tol * tol * (dg2 + off2)) && off2 == 0.0;
    for (; sweep < maxsweeps && !done; ++sweep) {
        for (int round = 0; round < nb - 1; ++round) {
            hipLaunchKernelGGL(k_jacobi_pair, dim3(npairs), dim3(256), 0, b->stream, np, nb, round, W, (int64_t)np,
                               Ubuf, 1);
            hipLaunchKernelGGL(k_jacobi_cols, dim3(npairs, (2 * np + 255) / 256), dim3(256), 0, b->stream, np, nb,
                               round, W, (int64_t)np, Vw, (int64_t)np, Ubuf);
            hipLaunchKernelGGL(k_jacobi_rows, dim3(npairs, (np + 255) / 256), dim3(256), 0, b->stream, np, nb, round,
                               W, (int64_t)np, Ubuf);
        }
        hipLaunchKernelGGL(k_offdiag_norm, dim3(redblocks), dim3(256), 0, b->stream, np, W, (int64_t)np, d_red);
        HIPCHK(hipMemcpyAsync(hred.data(), d_red, hred.size() * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        double o2 = 0.0;
        for (int i = 0; i < redblocks; ++i) o2 += hred[2 * i];
        if (!std::isfinite(o2)) return DFTK_MI_NUM_NONFINITE;
        const double off = sqrt(o2);
        if (off <= tol * fro) done = true;
        if (!done && prev_off >= 0.0 && off > 0.5 * prev_off && off <= 1e-12 * fro) done = true;
        prev_off = off;
    }
    HIPCHK(hipGetLastError());
    if (!done) {
        dftk_set_error("dense_heev: Jacobi did not converge in %d sweeps (n=%d)", maxsweeps, n);
        return DFTK_MI_NUM_EIGEN;
    }
    // eigenvalues = diag(W); sort ascending on the host, gather eigenvector columns
    hipLaunchKernelGGL(k_extract_diag, dim3((np + 255) / 256), dim3(256), 0, b->stream, np, W, (int64_t)np, d_diag);
    std::vector<double> diag(np);
    HIPCHK(hipMemcpyAsync(diag.data(), d_diag, np * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    std::vector<int> perm(np);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int c) { return diag[a] < diag[c]; });
    for (int i = 0; i < n; ++i) W_h[i] = diag[perm[i]];
    HIPCHK(hipMemcpyAsync(d_perm, perm.data(), n * sizeof(int), hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(k_gather_cols, dim3((n + 255) / 256, n), dim3(256), 0, b->stream, (int64_t)n, Vw, (int64_t)np,
                       d_perm, V, ldv);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));   // perm (host vector) must outlive the copy
    return 0;
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
    hipLaunchKernelGGL(k_col_reduce, dim3(m), dim3(256), 0, b->stream, 0, n, X, ldx, (const cd*)nullptr, (int64_t)0,
                       (const double*)nullptr, out_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_coldots(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const cd* Y, int64_t ldy,
               double* out_re_d) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_col_reduce, dim3(m), dim3(256), 0, b->stream, 1, n, X, ldx, Y, ldy, (const double*)nullptr,
                       out_re_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_weighted_colnorm2(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const double* w,
                         double* out_d) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_col_reduce, dim3(m), dim3(256), 0, b->stream, 2, n, X, ldx, (const cd*)nullptr, (int64_t)0,
                       w, out_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_frob2(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, double* out_d) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_col_reduce, dim3(m), dim3(256), 0, b->stream, 3, n, X, ldx, (const cd*)nullptr, (int64_t)0,
                       (const double*)nullptr, out_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_residual(dftk_mi_basis* b, int64_t n, int m, const cd* AX, int64_t lda, const cd* X, int64_t ldx,
                const double* lam_d, cd* R, int64_t ldr, double* norms_d) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_residual, dim3(m), dim3(256), 0, b->stream, n, AX, lda, X, ldx, lam_d, R, ldr, norms_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_tpa(dftk_mi_basis* b, int64_t n, int m, cd* R, int64_t ldr, const double* kin, const double* mean_kin_d) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_tpa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, n, m, R, ldr, kin,
                       mean_kin_d);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_scale_cols(dftk_mi_basis* b, int64_t n, int m, cd* X, int64_t ldx, const double* s_d, bool invert) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, n, m, X, ldx, s_d,
                       invert ? 1 : 0);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_copy(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, cd* Y, int64_t ldy) {
    if (m <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256), m), dim3(256), 0, b->stream, n, m, X, ldx, Y, ldy);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_fill_zero(dftk_mi_basis* b, cd* X, size_t count) {
    HIPCHK(hipMemsetAsync(X, 0, count * sizeof(cd), b->stream));
    return 0;
}
int ew_sub_identity_shifted(dftk_mi_basis* b, int rows, int cols, cd* C, int64_t ldc, int row0) {
    if (cols <= 0) return 0;
    hipLaunchKernelGGL(k_sub_identity_shifted, dim3((cols + 255) / 256), dim3(256), 0, b->stream, rows, cols, C, ldc,
                       row0);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_gather_cols(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, const int* perm_d, cd* Y,
                   int64_t ldy) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)((n + 255) / 256), m), dim3(256), 0, b->stream, n, X, ldx,
                       perm_d, Y, ldy);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_add_diag(dftk_mi_basis* b, int n, cd* A, int64_t lda, double shift) {
    hipLaunchKernelGGL(k_add_diag, dim3((n + 255) / 256), dim3(256), 0, b->stream, n, A, lda, shift);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_hermitize_upper(dftk_mi_basis* b, int n, cd* A, int64_t lda) {
    hipLaunchKernelGGL(k_hermitize_upper, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, b->stream, n,
                       A, lda);
    HIPCHK(hipGetLastError());
    return 0;
}
int ew_has_nonfinite(dftk_mi_basis* b, int64_t n, int m, const cd* X, int64_t ldx, double* out_d) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_has_nonfinite, dim3(m), dim3(256), 0, b->stream, n, X, ldx, out_d);
    HIPCHK(hipGetLastError());
    return 0;
}
