// eig_kernels.hip -- the LOWEST nev eigenpairs of a dense Hermitian matrix, GEMM-rich (gfx950).
//
// What rayleigh_ritz needs (src/eigen/lobpcg_hyper_impl.jl:141-153): `vectors[:, 1:N]`, `values[1:N]` of the
// 2N x 2N / 3N x 3N matrix Y' A Y -- the lowest THIRD (or half) of the spectrum.  The blocked Jacobi of
// dense_kernels.hip computes all of it with latency-bound rounds (29.7 ms at 1509^2).  Here the spectrum is split once:
//
//   1. sigma just above the nev-th smallest diagonal entry d_(nev).  The diagonal entries of the leading block of an
//      LOBPCG Rayleigh-Ritz matrix ARE its eigenvalues (X'AX = diag(lambda) of the previous iteration), so by Cauchy
//      interlacing at least nev eigenvalues of the whole matrix lie below sigma; for a general matrix the count is
//      verified after the split (k >= nev) and the full solver takes over if it fails.
//   2. U = sign(A - sigma I) by scaled Newton-Schulz iterations  X <- X (1.5 a I - 0.5 a^3 X^2), pure f64-MFMA GEMMs
//      (two per iteration: X'X with the UPPER flag, then X * coefficient matrix).  Scaling a(l) = sqrt(3 / (1 + l + l^2))
//      [the optimal cubic on [l, 1]]; l is held at L_HAT = 1e-2 for the first m1 iterations -- eigenvalues already
//      inside [L_HAT, 1] bounce inside [2.6e-2, 1], the small ones grow by 2.57 per iteration -- then follows its own
//      recurrence to 1 (8 iterations).  Holding l at 1e-2 instead of the true (tiny) gap keeps every eigenvalue above
//      2.6e-2 once it got there: round-off (eps / smallest magnitude passed through) cannot mix the two sides.
//   3. P = (I - U) / 2 is the spectral projector of the k = trace(P) eigenvalues below sigma (k >= nev, any k is fine).
//      An orthonormal basis B (n x k) of its range: the k columns of P with the largest diagonal entries (leverage
//      scores; for a converged LOBPCG block these are the X columns themselves), two passes of Cholesky-QR.
//   4. The k x k projected matrix B'(A - sigma)B / norm goes to the blocked Jacobi (k is a third of n: a ninth of the
//      work per sweep, a third of the rounds), V = B W[:, 1:nev] is one more GEMM.
//
// Real symmetric input (Gamma-real LOBPCG: every imaginary part exactly zero) runs on REAL matrix products: n x c
// matrices are held as "tall" arrays -- (n + 1) / 2 complex rows whose (re, im) are two consecutive REAL rows, the
// operand format of zgemm's DFTK_MI_GEMM_REAL flag ('C': A^T B over all real rows -> complex-with-zero-imaginary
// coefficient matrix; 'N': tall x Re(coefficient) -> tall).  Complex Hermitian input uses the plain 3M zgemm and tall =
// ordinary column-major.
#include "common.h"
#include "batch.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <numeric>
#include <algorithm>
#include <chrono>

static const double EIG_L_HAT = 1e-2;

namespace {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double bsum256(double v, double* sh) {   // result valid in thread 0
    v = wsum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// column j: diag[j] = Re A_jj, rowabs[j] = sum_i |A_ij|  (Hermitian: column sums = row sums)
__global__ __launch_bounds__(256) void k_eig_colstats(int n, const cd* __restrict__ A, int64_t lda,
                                                      double* __restrict__ diag, double* __restrict__ rowabs) {
    __shared__ double sh[4];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const cd v = A[i + (int64_t)j * lda];
        s += sqrt(v.x * v.x + v.y * v.y);
    }
    const double t = bsum256(s, sh);
    if (threadIdx.x == 0) {
        rowabs[j] = t;
        diag[j] = A[j + (int64_t)j * lda].x;
    }
}

// X0 = (A - sigma I) * inv_nrm in the tall format (REAL: packed doubles, pad row zero)
template <bool REAL>
__global__ __launch_bounds__(256) void k_eig_init(int n, const cd* __restrict__ A, int64_t lda, double sigma, double inv_nrm,
                                                  cd* __restrict__ X, int64_t ldt) {
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (REAL) {
        const int nrows = 2 * (int)ldt;           // real rows incl. padding
        if (i >= nrows) return;
        double v = 0.0;
        if (i < n) v = (A[i + (int64_t)j * lda].x - (i == j ? sigma : 0.0)) * inv_nrm;
        reinterpret_cast<double*>(X + (int64_t)j * ldt)[i] = v;
    } else {
        if (i >= n) return;
        cd v = A[i + (int64_t)j * lda];
        if (i == j) v.x -= sigma;
        X[i + (int64_t)j * ldt] = make_double2(v.x * inv_nrm, v.y * inv_nrm);
    }
}

// Y (n x n coefficient matrix, only the upper triangle valid) -> c1 I + c2 Y, Hermitian completion; per-tile partial
// sums of  sum_i (1 - Re y_ii)  and  ||I - Y||_F^2  (over the full matrix) into part[2 * tile + {0, 1}]
#define EPT 32
__global__ __launch_bounds__(256) void k_eig_poly(int n, cd* __restrict__ Y, int64_t ldy, double c1, double c2, int ntile,
                                                  double* __restrict__ part) {
    __shared__ cd tile[EPT][EPT + 1];
    __shared__ double sh[4];
    // tile index -> (bi <= bj)
    int t = blockIdx.x, bj = 0;
    while (t > bj) {
        t -= bj + 1;
        ++bj;
    }
    const int bi = t;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    double tr = 0.0, fro = 0.0;
#pragma unroll
    for (int r = 0; r < EPT; r += 8) {
        const int i = bi * EPT + tx, j = bj * EPT + ty + r;
        cd v = make_double2(0.0, 0.0);
        if (i < n && j < n && i <= j) {
            const cd y = Y[i + (int64_t)j * ldy];
            const double dre = (i == j ? 1.0 : 0.0) - y.x;
            const double e2 = dre * dre + y.y * y.y;
            if (i == j) {
                tr += dre;
                fro += dre * dre;          // (the imaginary part of a diagonal entry is round-off: dropped)
            } else {
                fro += 2.0 * e2;
            }
            v = make_double2((i == j ? c1 : 0.0) + c2 * y.x, i == j ? 0.0 : c2 * y.y);
            Y[i + (int64_t)j * ldy] = v;
        }
        tile[ty + r][tx] = v;
    }
    __syncthreads();
    // mirrored tile (bj, bi): entry (j, i) = conj(entry (i, j)), strictly lower part only
#pragma unroll
    for (int r = 0; r < EPT; r += 8) {
        const int j = bj * EPT + tx, i = bi * EPT + ty + r;      // writes row j (fast), column i
        if (i < n && j < n && i < j) {
            const cd v = tile[tx][ty + r];
            Y[j + (int64_t)i * ldy] = make_double2(v.x, -v.y);
        }
    }
    const double a = bsum256(tr, sh);
    const double f = bsum256(fro, sh);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = a;
        part[2 * blockIdx.x + 1] = f;
    }
    (void)ntile;
}

template <bool REAL>
__global__ void k_eig_diag_tall(int n, const cd* __restrict__ U, int64_t ldt, double* __restrict__ d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    if (REAL)
        d[c] = reinterpret_cast<const double*>(U + (int64_t)c * ldt)[c];
    else
        d[c] = U[c + (int64_t)c * ldt].x;
}

// B[:, c] = column sel[c] of the projector (I - U) / 2, tall format
template <bool REAL>
__global__ __launch_bounds__(256) void k_eig_proj_cols(int n, const cd* __restrict__ U, int64_t ldt, const int* __restrict__ sel,
                                                       cd* __restrict__ Bt, int64_t ldb) {
    const int c = blockIdx.y;
    const int s = sel[c];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (REAL) {
        if (i >= 2 * (int)ldb) return;
        double v = 0.0;
        if (i < n) v = -0.5 * reinterpret_cast<const double*>(U + (int64_t)s * ldt)[i] + (i == s ? 0.5 : 0.0);
        reinterpret_cast<double*>(Bt + (int64_t)c * ldb)[i] = v;
    } else {
        if (i >= n) return;
        const cd u = U[i + (int64_t)s * ldt];
        Bt[i + (int64_t)c * ldb] = make_double2(-0.5 * u.x + (i == s ? 0.5 : 0.0), -0.5 * u.y);
    }
}

// REAL tall (n real rows x c) -> complex-with-zero-imaginary column-major (n x c)
__global__ __launch_bounds__(256) void k_eig_tall_to_cd(int n, const cd* __restrict__ T, int64_t ldt, cd* __restrict__ C,
                                                        int64_t ldc) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    C[i + (int64_t)c * ldc] = make_double2(reinterpret_cast<const double*>(T + (int64_t)c * ldt)[i], 0.0);
}

int ws_ensure(dftk_mi_basis* b, size_t bytes) {
    if (bytes <= b->eig_ws_bytes) return 0;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->eig_ws) HIPCHK(hipFree(b->eig_ws));
    b->eig_ws = nullptr;
    b->eig_ws_bytes = 0;
    HIPCHK(dftk_scratch_malloc(&b->eig_ws, bytes));
    b->eig_ws_bytes = bytes;
    return 0;
}

double a_of(double l) { return std::sqrt(3.0 / (1.0 + l + l * l)); }
double l_next(double l, double a) { return 0.5 * a * l * (3.0 - a * a * l * l); }

}   // namespace

// Host-only statement of the sigma rule (also exported for the CPU test-suite): the nev-th smallest diagonal entry plus
// `margin` = half the mean level spacing of the nev smallest diagonal entries; gap_guess = a tenth of that spacing
int eig_choose_sigma_host(int n, const double* diag, int nev, double* sigma, double* gap_guess) {
    if (n < 2 || nev < 1 || nev >= n || !diag || !sigma || !gap_guess) return DFTK_MI_EINVAL;
    std::vector<double> ds(diag, diag + n);
    std::nth_element(ds.begin(), ds.begin() + (nev - 1), ds.end());
    const double dn = ds[nev - 1];
    const double d1 = *std::min_element(ds.begin(), ds.begin() + nev);
    const double dnext = *std::min_element(ds.begin() + nev, ds.end());
    double spacing = (dn - d1) / std::max(nev - 1, 1);
    if (!(spacing > 0.0)) spacing = 1e-8 * std::max(1.0, std::fabs(dn));
    double margin = 0.5 * spacing;
    // a wide gap right above d_(nev) in the diagonal (a converged block): sit well inside it, but never further than
    // four spacings -- the diagonal entries of the trailing blocks are Rayleigh quotients, not eigenvalues
    const double gap = dnext - dn;
    if (gap > 2.0 * margin) margin = std::min(0.5 * gap, 4.0 * spacing);
    *sigma = dn + margin;
    *gap_guess = 0.2 * margin;
    return 0;
}

// Number of scaled iterations at fixed l = L_HAT before the recurrence phase (host; exported for the CPU tests)
int eig_hold_iterations_host(double gap_over_norm) {
    const double grow = 1.5 * a_of(EIG_L_HAT);
    const double l0 = std::max(gap_over_norm, 1e-14);
    if (l0 >= EIG_L_HAT) return 0;
    const int m1 = (int)std::ceil(std::log(EIG_L_HAT / l0) / std::log(grow));
    return std::min(m1, 34);
}

template <bool REAL>
static int heev_lowest_impl(dftk_mi_basis* b, int n, int nev, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv,
                            bool* fell_back) {
    static const bool trace = getenv("DFTK_MI_HEEV_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&](const std::chrono::steady_clock::time_point& t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    *fell_back = true;
    const int rf = REAL ? DFTK_MI_GEMM_REAL : 0;
    const int64_t ldt = REAL ? (n + 1) / 2 : n;             // leading dimension (complex units) of tall arrays
    const int64_t tk = REAL ? ldt : n;                        // inner dimension of a 'C' product over all rows
    const int kcap = std::min(n - 1, std::max(nev + 64, (3 * nev) / 2));
    const int ntile1 = (n + EPT - 1) / EPT, ntile = ntile1 * (ntile1 + 1) / 2;
    // ---- workspace
    const size_t szTall = (size_t)ldt * n, szY = (size_t)n * n, szB = (size_t)ldt * kcap, szK = (size_t)kcap * kcap;
    const size_t n_cd = 3 * szTall + szY + 3 * szB + (REAL ? (size_t)n * kcap : 0) + 4 * szK;
    const size_t n_dbl = 3 * (size_t)n + 2 * (size_t)ntile + 64;
    CHK(ws_ensure(b, n_cd * sizeof(cd) + n_dbl * sizeof(double) + (size_t)n * sizeof(int) + 256));
    cd* w = reinterpret_cast<cd*>(b->eig_ws);
    auto take = [&](size_t c) {
        cd* r = w;
        w += c;
        return r;
    };
    cd* X0 = take(szTall);
    cd* Xa = take(szTall);
    cd* Xb = take(szTall);
    cd* Y = take(szY);
    cd* Bt[2] = {take(szB), take(szB)};
    cd* Tt = take(szB);
    cd* Bc = REAL ? take((size_t)n * kcap) : nullptr;
    cd* O = take(szK);
    cd* invR = take(szK);
    cd* Gk = take(szK);
    cd* Vk = take(szK);
    double* d_diag = reinterpret_cast<double*>(w);
    double* d_rowabs = d_diag + n;
    double* d_udiag = d_rowabs + n;
    double* d_part = d_udiag + n;
    int* d_sel = reinterpret_cast<int*>(d_part + 2 * ntile + 64);
    const cd ONE = make_double2(1.0, 0.0), ZERO = make_double2(0.0, 0.0);

    // ---- sigma, norm bound
    hipLaunchKernelGGL(k_eig_colstats, dim3(n), dim3(256), 0, b->stream, n, A, lda, d_diag, d_rowabs);
    std::vector<double> hd(2 * (size_t)n);
    CHK(host_fetch(b, hd.data(), d_diag, hd.size() * sizeof(double)));
    double sigma, gap_guess;
    CHK(eig_choose_sigma_host(n, hd.data(), nev, &sigma, &gap_guess));
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!std::isfinite(hd[i]) || !std::isfinite(hd[n + i])) return DFTK_MI_NUM_NONFINITE;
        nrm = std::max(nrm, std::fabs(hd[i] - sigma) + hd[n + i] - std::fabs(hd[i]));   // Gershgorin: ||A - sigma I||_2 <= this
    }
    nrm *= 1.0 + 1e-8;            // (round-off of the sums: the bound must hold strictly, x > 1 would change sign)
    if (!(nrm > 0.0)) return 0;   // (fell_back stays true: the caller runs the full solver)
    const unsigned rowblocks = (unsigned)((2 * ldt + 255) / 256);
    hipLaunchKernelGGL(k_eig_init<REAL>, dim3(rowblocks, n), dim3(256), 0, b->stream, n, A, lda, sigma, 1.0 / nrm, X0, ldt);
    HIPCHK(hipGetLastError());

    // ---- U = sign(X0)
    cd* Xc = X0;               // current iterate (X0 itself is kept: the projected matrix is formed from it)
    cd* Xn = Xa;
    int its = 0;
    std::vector<double> hpart(2 * (size_t)ntile);
    auto gram_poly = [&](double a) -> int {   // Y = c1 I + c2 Xc' Xc, partial sums of the error in d_part
        CHK(zgemm(b, 'C', n, n, tk, ONE, Xc, ldt, Xc, ldt, ZERO, Y, n, rf | DFTK_MI_GEMM_UPPER));
        hipLaunchKernelGGL(k_eig_poly, dim3(ntile), dim3(256), 0, b->stream, n, Y, (int64_t)n, 1.5 * a, -0.5 * a * a * a, ntile,
                           d_part);
        HIPCHK(hipGetLastError());
        return 0;
    };
    auto advance = [&]() -> int {             // Xn = Xc * Y
        CHK(zgemm(b, 'N', ldt, n, n, ONE, Xc, ldt, Y, n, ZERO, Xn, ldt, rf));
        Xc = Xn;
        Xn = (Xc == Xa) ? Xb : Xa;
        ++its;
        return 0;
    };
    auto error_norm = [&](double* e) -> int { // ||I - Xc'Xc||_F of the iterate the last gram_poly saw
        CHK(host_fetch(b, hpart.data(), d_part, hpart.size() * sizeof(double)));
        double f = 0.0;
        for (int t = 0; t < ntile; ++t) f += hpart[2 * t + 1];
        if (!std::isfinite(f)) return DFTK_MI_NUM_NONFINITE;
        *e = std::sqrt(f);
        return 0;
    };
    int m1 = eig_hold_iterations_host(gap_guess / nrm);
    bool converged = false;
    int rescues = 0;
    for (; rescues <= 3 && !converged; ++rescues) {
        const double ah = a_of(EIG_L_HAT);
        for (int i = 0; i < m1; ++i) {
            CHK(gram_poly(ah));
            CHK(advance());
        }
        for (double l = EIG_L_HAT; l < 1.0 - 1e-9;) {
            const double a = a_of(l);
            CHK(gram_poly(a));
            CHK(advance());
            l = l_next(l, a);
        }
        // checked phase, a = 1 (plain Newton-Schulz: x -> x (3 - x^2) / 2 converges from any x in (0, sqrt 3), quadratically
        // near 1).  An eigenvalue that was closer to sigma than assumed is still small here: above ||I - X^2||_F = 0.9
        // (|x| < 0.3, or several such) another held phase is cheaper than waiting for the factor 1.5 per iteration.
        for (int chk = 0; chk < 10; ++chk) {
            double e;
            CHK(gram_poly(1.0));
            CHK(error_norm(&e));
            if (trace) fprintf(stderr, "[heev lowest%s] n=%d nev=%d its=%d ||I - X^2||_F=%.3e\n", REAL ? " real" : "", n, nev, its, e);
            if (e < 1e-10) {
                // one more a = 1 step unless the iterate already sits at round-off (Y is formed: one product): eigenvalues next
                // to sigma may still be ~e / 2 away from +-1, and that much of the neighbouring eigenvectors would leak into
                // the projector -- a residual floor of e * |lambda' - lambda| for tight LOBPCG tolerances
                if (e >= 1e-13) CHK(advance());
                converged = true;
                break;
            }
            if (e > 0.9) break;
            CHK(advance());
            if (e < 3e-6) {                    // quadratic from here: the next iterate is at round-off
                converged = true;
                break;
            }
        }
        m1 = 6;
    }
    if (!converged) {
        if (trace) fprintf(stderr, "[heev lowest] n=%d: no convergence of the sign iteration after %d iterations -> full solver\n", n, its);
        return 0;
    }
    const double ms_sign = ms_since(t_start);   // (the last check was a synchronising fetch)
    // ---- k and the columns of the projector with the largest leverage
    if (REAL)
        hipLaunchKernelGGL(k_eig_diag_tall<true>, dim3((n + 255) / 256), dim3(256), 0, b->stream, n, Xc, ldt, d_udiag);
    else
        hipLaunchKernelGGL(k_eig_diag_tall<false>, dim3((n + 255) / 256), dim3(256), 0, b->stream, n, Xc, ldt, d_udiag);
    std::vector<double> p(n);
    CHK(host_fetch(b, p.data(), d_udiag, n * sizeof(double)));
    double trP = 0.0;
    for (int i = 0; i < n; ++i) {
        p[i] = 0.5 * (1.0 - p[i]);
        trP += p[i];
    }
    const int k = (int)std::lround(trP);
    if (!(std::fabs(trP - k) < 1e-6) || k < nev || k > kcap) {
        if (trace) fprintf(stderr, "[heev lowest] n=%d nev=%d: trace(P)=%.9f (k cap %d) -> full solver\n", n, nev, trP, kcap);
        return 0;
    }
    std::vector<int> sel(n);
    std::iota(sel.begin(), sel.end(), 0);
    std::stable_sort(sel.begin(), sel.end(), [&](int x, int y) { return p[x] > p[y]; });
    sel.resize(k);
    std::sort(sel.begin(), sel.end());
    HIPCHK(hipMemcpyAsync(d_sel, sel.data(), k * sizeof(int), hipMemcpyHostToDevice, b->stream));
    CHK(host_wait(b));         // (sel is a host vector: the copy is complete before any early return can destroy it)
    hipLaunchKernelGGL(k_eig_proj_cols<REAL>, dim3(rowblocks, k), dim3(256), 0, b->stream, n, Xc, ldt, d_sel, Bt[0], ldt);
    HIPCHK(hipGetLastError());
    // ---- Cholesky-QR passes
    int cur = 0;
    for (int pass = 0; pass < 3; ++pass) {
        CHK(zgemm(b, 'C', k, k, tk, ONE, Bt[cur], ldt, Bt[cur], ldt, ZERO, O, k, rf | DFTK_MI_GEMM_UPPER));
        CHK(ew_hermitize_upper(b, k, O, k));
        double nR = 0.0, nI = 0.0;
        const int st = dense_potrf_trtri(b, k, O, k, invR, k, &nR, &nI, REAL);
        if (st > 0) {
            if (trace) fprintf(stderr, "[heev lowest] n=%d k=%d: Cholesky-QR pass %d failed -> full solver\n", n, k, pass);
            return 0;
        }
        if (st != 0) return st;
        CHK(zgemm(b, 'N', ldt, k, k, ONE, Bt[cur], ldt, invR, k, ZERO, Bt[cur ^ 1], ldt, rf | DFTK_MI_GEMM_B_UPPER));
        cur ^= 1;
        const double cond = nR * nI;
        if (trace) fprintf(stderr, "[heev lowest] n=%d k=%d sigma=%.6f its=%d pass %d cond(R)~%.2e\n", n, k, sigma, its, pass, cond);
        if (pass >= 1 && 2.220446049250313e-16 * cond * cond < 1e-15) break;
        if (pass == 0 && cond < 16.0) break;      // (eps cond^2 < 6e-14: orthonormal to round-off after one pass)
    }
    cd* Bq = Bt[cur];
    // ---- projected matrix  Gk = Bq' X0 Bq
    const cd* Bcoef = Bq;
    if (REAL) {
        hipLaunchKernelGGL(k_eig_tall_to_cd, dim3((n + 255) / 256, k), dim3(256), 0, b->stream, n, Bq, ldt, Bc, (int64_t)n);
        Bcoef = Bc;
    }
    CHK(zgemm(b, 'N', ldt, k, n, ONE, X0, ldt, Bcoef, n, ZERO, Tt, ldt, rf));
    CHK(zgemm(b, 'C', k, k, tk, ONE, Bq, ldt, Tt, ldt, ZERO, Gk, k, rf | DFTK_MI_GEMM_UPPER));
    CHK(ew_hermitize_upper(b, k, Gk, k));
    std::vector<double> th(k);
    double ms_basis = 0.0;
    if (trace) {
        CHK(host_wait(b));
        ms_basis = ms_since(t_start) - ms_sign;
    }
    {
        const int st = dense_heev_full(b, k, Gk, k, th.data(), Vk, k);
        if (st != 0) return st;
    }
    if (trace)
        fprintf(stderr, "[heev lowest timing] n=%d nev=%d k=%d its=%d: sign %.2f ms, basis + projection %.2f ms, Jacobi(k) %.2f ms\n", n,
                nev, k, its, ms_sign, ms_basis, ms_since(t_start) - ms_sign - ms_basis);
    // ---- V[:, :nev] = Bq Vk[:, :nev]
    if (REAL) {
        CHK(zgemm(b, 'N', ldt, nev, k, ONE, Bq, ldt, Vk, k, ZERO, Tt, ldt, rf));
        hipLaunchKernelGGL(k_eig_tall_to_cd, dim3((n + 255) / 256, nev), dim3(256), 0, b->stream, n, Tt, ldt, V, ldv);
    } else {
        CHK(zgemm(b, 'N', n, nev, k, ONE, Bq, ldt, Vk, k, ZERO, V, ldv, rf));
    }
    HIPCHK(hipGetLastError());
    for (int i = 0; i < nev; ++i) W_h[i] = sigma + nrm * th[i];
    *fell_back = false;
    return 0;
}

// Lowest nev eigenpairs (W_h[0 .. nev), V[:, 0 .. nev)) of the Hermitian n x n matrix A (full storage, left intact
// unless the full solver takes over).  Falls back to dense_heev (which then fills all n pairs) for small problems, inside
// batched calls, when nev is more than 60 % of n, or when the split fails its checks.
int dense_heev_lowest(dftk_mi_basis* b, int n, int nev, cd* A, int64_t lda, double* W_h, cd* V, int64_t ldv) {
    if (n <= 0) return 0;
    static const int min_n = getenv("DFTK_MI_HEEV_PARTIAL_MIN") ? atoi(getenv("DFTK_MI_HEEV_PARTIAL_MIN")) : 384;
    static const bool off = getenv("DFTK_MI_HEEV_PARTIAL") && atoi(getenv("DFTK_MI_HEEV_PARTIAL")) == 0;
    if (off || batching() || !b->use_mfma || n < min_n || nev < 1 || 10 * nev > 6 * n)
        return dense_heev(b, n, A, lda, W_h, V, ldv);
    bool fell_back = true;
    int st;
    {
        ProfScope scope(b, PROF_HEEV, (double)n);   // the whole call, as dense_heev books itself
        ProfMute mute(b);                           // ... its products and factorisations are not booked a second time
        // real symmetric input?  (sum of squared imaginary parts exactly zero: the Rayleigh-Ritz matrices of the
        // Gamma-real iteration, lobpcg.cpp)
        double off2 = 0.0, dg2 = 0.0, im2 = 0.0;
        CHK(dense_input_norms(b, n, A, lda, &off2, &dg2, &im2));
        if (!std::isfinite(off2 + dg2 + im2)) return DFTK_MI_NUM_NONFINITE;
        st = im2 == 0.0 ? heev_lowest_impl<true>(b, n, nev, A, lda, W_h, V, ldv, &fell_back)
                        : heev_lowest_impl<false>(b, n, nev, A, lda, W_h, V, ldv, &fell_back);
        // the split declined: the full solver INSIDE this scope (one PROF_HEEV booking per call on either path)
        if (st == 0 && fell_back) return dense_heev_full(b, n, A, lda, W_h, V, ldv);
    }
    if (st != 0) return st;
    // Output contract by path: nev pairs with A intact here, n pairs with A destroyed after a fallback.  Callers may only
    // read the first nev; DFTK_MI_POISON=1 makes a reader of the rest visible (NaN eigenvalues and vector columns).
    static const bool poison = getenv("DFTK_MI_POISON") != nullptr;
    if (poison && nev < n) {
        for (int i = nev; i < n; ++i) W_h[i] = std::nan("");
        HIPCHK(hipMemset2DAsync(V + (int64_t)nev * ldv, (size_t)ldv * sizeof(cd), 0xFF, (size_t)n * sizeof(cd), (size_t)(n - nev),
                                b->stream));
    }
    return 0;
}
