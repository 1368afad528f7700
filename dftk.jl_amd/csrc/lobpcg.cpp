// lobpcg.cpp -- host driver of the LOBPCG ("hyper") eigensolver, all block algebra on the device.
//
// Mirrors LOBPCG(A, X, I, precon, tol, maxiter; miniter, ortho_tol, n_conv_check) of
// src/eigen/lobpcg_hyper_impl.jl:354-582 (B = I) including rayleigh_ritz (:141-171),
// safe_cholesky (:190-210), ortho!(X) (:216-261), drop_small! (:264-268), ortho!(X,Y,BY)
// (:271-323), final_retval (:325-338), compute_lambda (:341-344) and the TPA preconditioner of
// src/eigen/preconditioners.jl:50-77.  Control flow (locking, active views, adaptive
// orthogonalisation loops) runs on the host from a handful of reduced scalars per iteration;
// every n_G-sized operation is a kernel on the basis' stream.
#include "common.h"
#include "batch.h"
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstring>
#include <functional>
#include <numeric>
#include <random>
#include <utility>
#include <vector>

namespace {

const double EPS = 2.220446049250313e-16;
// ortho!(X, Y) on blocks up to this many elements lets its drop_small! fetch ride on the next Cholesky status (ortho_XY)
const int64_t DEFER_FETCH_MAX_ELEMS = 1 << 16;
const cd ONE = {1.0, 0.0}, ZERO = {0.0, 0.0}, MONE = {-1.0, 0.0};

struct Mat {            // column-major view
    cd* p;
    int64_t ld;
    int64_t rows;
    int cols;
    Mat cols_from(int c0, int nc = -1) const {
        return Mat{p + (int64_t)c0 * ld, ld, rows, nc < 0 ? cols - c0 : nc};
    }
    Mat rows_from(int64_t r0, int64_t nr) const { return Mat{p + r0, ld, nr, cols}; }
};

struct Ctx {
    dftk_mi_kblock* kb;
    dftk_mi_basis* b;
    // small scratch (device)
    cd *O, *Rw, *invR, *BYX, *tmpS;
    cd* Vh;                   // m x m: V^H of the SVD fallback (survives the Cholesky-QR polish)
    double *d_a, *d_b;        // M-sized double scratch; d_b == d_a + dstride (adjacent: one fetch serves both)
    int dstride = 0;
    std::vector<double> h;    // host scratch
    std::mt19937_64 rng;      // re-randomised columns of n_G-sized blocks (per-rank stream: each rank draws its slab)
    std::mt19937_64 rng_rep;  // ... of REPLICATED small matrices of a sharded run: must be identical on all ranks
    bool replicated = false;  // inside a NoComm scope
    bool small = false;       // inside a NoComm scope (sharded or not): the operands are small replicated matrices
    int n_svd = 0;            // SVD fallbacks taken
    // Gamma-real mode (gamma_kernels.hip): the n_G-sized blocks are half-sphere images of real-symmetric vectors.
    // rf = DFTK_MI_GEMM_REAL is OR-ed into every product that involves them; small matrices are real (stored as
    // complex with zero imaginary parts) and keep the general complex kernels (rf = 0 inside NoComm).
    bool real_mode = false;
    bool holds_g0 = false;    // this rank's slab starts with the G = 0 row (whose imaginary part must stay zero)
    int rf = 0;
    int mm(char transA, int64_t m, int64_t n, int64_t k, cd alpha, const cd* A, int64_t lda, const cd* B, int64_t ldb,
           cd beta, cd* C, int64_t ldc, int upper = 0) {
        return zgemm(b, transA, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, upper | rf);
    }
    // Row-slab (plane-wave) sharding: every product with the long dimension n_G as its inner dimension and
    // every column reduction is a LOCAL partial sum followed by an all-reduce over the block's communicator
    // (no-ops for an unsharded block).  reduce_norms: the buffer holds sqrt(local sums).
    dftk_mi_comm* comm = nullptr;
    // (every call outside a NoComm scope is also booked in PROF_AR_MODEL: on ONE rank that is the list of all-reduces
    //  a plane-wave-sharded run of the same iteration performs -- bench.py's Amdahl model is built from it)
    int reduce_c(cd* d, size_t n) {
        if (!small) prof_count(b, PROF_AR_MODEL, 16.0 * (double)n);
        return comm ? comm_allreduce(comm, c_b(), reinterpret_cast<double*>(d), 2 * n) : 0;
    }
    int reduce_d(double* d, size_t n) {
        if (!small) prof_count(b, PROF_AR_MODEL, 8.0 * (double)n);
        return comm ? comm_allreduce(comm, c_b(), d, n) : 0;
    }
    int reduce_norms(double* d, size_t n) {
        if (!small) prof_count(b, PROF_AR_MODEL, 8.0 * (double)n);
        return comm ? comm_allreduce_norms(comm, c_b(), d, n) : 0;
    }
    dftk_mi_basis* c_b() { return b; }
};

// host <-> device traffic of the driver.  Inside a fiber of a batched multi-k call (batch.h) these are recorded like
// every other device operation: the copies of all k-blocks of a round travel together, and "wait for the result" is
// the point where the fiber yields to its siblings.
int h2d(dftk_mi_basis* b, void* dst_d, const void* src_h, size_t bytes) { return dev_h2d(b, dst_d, src_h, bytes); }
int stream_sync(dftk_mi_basis* b) { return dev_stream_sync(b); }
int d2h_sync(dftk_mi_basis* b, void* dst_h, const void* src_d, size_t bytes) { return dev_d2h_sync(b, dst_h, src_d, bytes); }

int d2h(Ctx& c, const double* d, int n) {
    if ((int)c.h.size() < n) c.h.resize(n);
    return d2h_sync(c.b, c.h.data(), d, n * sizeof(double));
}

// Frobenius norm^2 of a matrix (sum over columns)
int frob2(Ctx& c, const Mat& A, double* out) {
    CHK(ew_frob2(c.b, A.rows, A.cols, A.p, A.ld, c.d_a));
    CHK(d2h(c, c.d_a, A.cols));
    double s = 0.0;
    for (int i = 0; i < A.cols; ++i) s += c.h[i];
    *out = s;
    return 0;
}

// safe_cholesky(O): R'R = O with shift-and-retry.  On success Rw holds R (upper), invR its inverse.
int safe_cholesky(Ctx& c, int m, int* nchol_out, double* nR, double* nI) {
    int nchol = 0;
    double alpha = 100.0;
    for (;;) {
        if (nchol >= 5) {
            *nchol_out = 10000;
            return 0;
        }
        nchol += 1;
        CHK(ew_copy(c.b, m, m, c.O, m, c.Rw, m));
        // (Gamma-real mode: the Gram matrices are real symmetric, stored as complex with exactly zero imaginary parts)
        int st = dense_potrf_trtri(c.b, m, c.Rw, m, c.invR, m, nR, nI, c.real_mode);
        if (st == 0) break;
        if (st != DFTK_MI_NUM_CHOLESKY) return st;
        double f2;
        CHK(frob2(c, Mat{c.O, m, m, m}, &f2));
        if (!std::isfinite(f2)) return DFTK_MI_NUM_NONFINITE;
        CHK(ew_add_diag(c.b, m, c.O, m, alpha * EPS * std::sqrt(f2)));
        alpha *= 10.0;
    }
    *nchol_out = nchol;
    return 0;
}

int randomize_column(Ctx& c, Mat X, int col);

// SVD fallback of ortho! (lobpcg_hyper_impl.jl:226-231, :307-314): X <- U V' with X = U S V' (the unitary polar
// factor).  Built from the m x m Gram matrix: X'X = V S^2 V' (blocked Jacobi heev), W = X V has orthogonal
// columns of norm s_i; columns with s_i <= 1e-7 s_max carry no direction (LAPACK completes U arbitrarily there)
// and are re-randomised; the scaled W is polished by Cholesky-QR passes (it is close to orthonormal, so these
// are benign), then X = W V'.  `X` holds the data, `scratch` (n x m, leading dimension scratch_ld) is destroyed.
int svd_polar(Ctx& c, Mat X, cd* scratch, int64_t scratch_ld);

// ortho!(X): Cholesky-QR until the a-posteriori estimate eps*cond(R)^2 < tol.
// tmp must hold rows x cols elements (leading dimension tmp_ld, default rows).
// `hook` (optional): called once, right after the first Cholesky factorisation has come back (the first host
// synchronisation of the call) and before X is touched; a non-zero result is handed to the caller in *hook_out and the call
// returns at once with X unchanged (ortho_XY lets a fetch of its own ride on that synchronisation).
int ortho_X(Ctx& c, Mat X, cd* tmp, double tol, int* nchol_total_out, double* growth_out, bool allow_svd = true,
            int64_t tmp_ld = 0, bool force_svd = false, const std::function<int()>* hook = nullptr, int* hook_out = nullptr) {
    double growth = 1.0;
    int nchol_total = 0;
    const int m = X.cols;
    if (m == 0) {
        *growth_out = 1.0;
        *nchol_total_out = 0;
        return 0;
    }
    if (tmp_ld <= 0) tmp_ld = X.rows;
    // the passes alternate between X and tmp (X * invR cannot be formed in place); one copy at the end
    // if the result happens to sit in tmp
    Mat src = X, dst{tmp, tmp_ld, X.rows, m};
    for (int pass = 0;; ++pass) {
        if (pass >= 30) {
            dftk_set_error("ortho!(X) did not reach the orthogonality tolerance in 30 Cholesky-QR passes");
            return DFTK_MI_NUM_CHOLESKY;
        }
        CHK(c.mm('C', m, m, src.rows, ONE, src.p, src.ld, src.p, src.ld, ZERO, c.O, m, /*upper=*/1));
        CHK(c.reduce_c(c.O, (size_t)m * m));
        CHK(ew_hermitize_upper(c.b, m, c.O, m));
        int nchol = 10000;
        double nR = 0, nI = 0;
        int st_chol = 0;
        if (!force_svd) st_chol = safe_cholesky(c, m, &nchol, &nR, &nI);
        // a speculative call (hook): the caller's verdict on X comes first -- a non-finite column must be reported as
        // such, not as whatever the factorisation of its Gram matrix ran into
        if (st_chol != 0 && pass == 0 && hook) CHK(stream_sync(c.b));   // (the hook reads a fetch that rides on a synchronisation)
        if (pass == 0 && hook) {
            const int hr = (*hook)();
            if (hr != 0) {
                *hook_out = hr;
                *growth_out = 1.0;
                *nchol_total_out = 0;
                return 0;
            }
        }
        if (st_chol != 0) return st_chol;
        nchol_total += nchol;
        if (nchol > 10) {
            if (!allow_svd) {
                dftk_set_error("ortho!(X): Cholesky keeps failing inside the SVD fallback");
                return DFTK_MI_NUM_CHOLESKY;
            }
            // "Ortho(X) is failing badly, falling back to SVD": X = U V', nchol = 100, growth_factor = 1
            CHK(svd_polar(c, src, dst.p, dst.ld));
            if (src.p != X.p) CHK(ew_copy(c.b, X.rows, m, src.p, src.ld, X.p, X.ld));
            *growth_out = 1.0;
            *nchol_total_out = 100;
            c.n_svd += 1;
            return 0;
        }
        // X <- X * invR
        CHK(c.mm('N', src.rows, m, m, ONE, src.p, src.ld, c.invR, m, ZERO, dst.p, dst.ld,
                  /*B upper triangular=*/2));
        std::swap(src, dst);
        growth *= nI;
        const double condR = nR * nI;
        const double est = EPS * condR * condR;
        if (nchol == 1 && est < tol) break;
    }
    if (src.p != X.p) CHK(ew_copy(c.b, X.rows, m, src.p, src.ld, X.p, X.ld));
    *growth_out = growth;
    *nchol_total_out = nchol_total;
    return 0;
}

int svd_polar(Ctx& c, Mat X, cd* scratch, int64_t scratch_ld) {
    const int m = X.cols;
    // c.O holds the hermitised Gram matrix X'X (ortho_X computed it; the shifts of safe_cholesky may have
    // touched it, so recompute)
    CHK(c.mm('C', m, m, X.rows, ONE, X.p, X.ld, X.p, X.ld, ZERO, c.O, m, /*upper=*/1));
    CHK(c.reduce_c(c.O, (size_t)m * m));
    CHK(ew_hermitize_upper(c.b, m, c.O, m));
    std::vector<double> w(m);
    CHK(dense_heev(c.b, m, c.O, m, w.data(), c.Rw, m));          // V in Rw (O is destroyed)
    Mat W{scratch, scratch_ld, X.rows, m};
    CHK(c.mm('N', X.rows, m, m, ONE, X.p, X.ld, c.Rw, m, ZERO, W.p, W.ld));
    CHK(ew_colnorms(c.b, W.rows, m, W.p, W.ld, c.d_a));
    CHK(c.reduce_norms(c.d_a, m));
    CHK(d2h(c, c.d_a, m));
    double smax = 0.0;
    for (int i = 0; i < m; ++i) {
        if (!std::isfinite(c.h[i])) return DFTK_MI_NUM_NONFINITE;
        smax = std::max(smax, c.h[i]);
    }
    std::vector<double> sig(c.h.begin(), c.h.begin() + m);
    bool any_null = false;
    for (int i = 0; i < m; ++i)
        if (!(sig[i] > 1e-7 * smax) || smax == 0.0) {
            CHK(randomize_column(c, W, i));
            any_null = true;
        }
    if (any_null) {
        CHK(ew_colnorms(c.b, W.rows, m, W.p, W.ld, c.d_a));
        CHK(c.reduce_norms(c.d_a, m));
    }
    CHK(ew_scale_cols(c.b, W.rows, m, W.p, W.ld, c.d_a, true));
    CHK(ew_conj_transpose(c.b, m, c.Rw, m, c.Vh, m));                         // V' (the polish below reuses O, Rw, invR)
    int nch;
    double gr;
    CHK(ortho_X(c, W, X.p, 2 * EPS, &nch, &gr, /*allow_svd=*/false, X.ld));   // X's storage is free now
    CHK(c.mm('N', X.rows, m, m, ONE, W.p, W.ld, c.Vh, m, ZERO, X.p, X.ld));
    return 0;
}

// X[:, col] = randn (complex) -- host RNG, rare path of drop_small!
int randomize_column(Ctx& c, Mat X, int col) {
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> v(2 * X.rows);
    std::mt19937_64& gen = c.replicated ? c.rng_rep : c.rng;
    for (auto& x : v) x = nd(gen);
    if (c.real_mode) {
        // small matrices are real; a half-sphere vector is free except for the imaginary part of its G = 0 row
        if (c.small)
            for (size_t i = 1; i < v.size(); i += 2) v[i] = 0.0;
        else if (c.holds_g0)
            v[1] = 0.0;
    }
    CHK(h2d(c.b, X.p + (int64_t)col * X.ld, v.data(), v.size() * sizeof(double)));
    return stream_sync(c.b);
}

// true when the blocks are adjacent column ranges of one array (same leading dimension)
bool contiguous(const std::vector<Mat>& Ys) {
    for (size_t i = 0; i + 1 < Ys.size(); ++i)
        if (Ys[i].ld != Ys[i + 1].ld || Ys[i].rows != Ys[i + 1].rows ||
            Ys[i].p + (int64_t)Ys[i].cols * Ys[i].ld != Ys[i + 1].p)
            return false;
    return !Ys.empty();
}
int total_cols(const std::vector<Mat>& Ys) {
    int n = 0;
    for (auto& Y : Ys) n += Y.cols;
    return n;
}

// ortho!(X, Y, BY) with Y = hcat(Ys...), B = I
// norms_d (optional): device array of the column norms of X, when the producer of X already has them
int ortho_XY(Ctx& c, Mat X, const std::vector<Mat>& Ys, cd* tmp, double tol, const double* norms_d = nullptr) {
    if (X.cols == 0) return 0;
    if (!norms_d) {
        CHK(ew_colnorms(c.b, X.rows, X.cols, X.p, X.ld, c.d_a));
        CHK(c.reduce_norms(c.d_a, X.cols));
        norms_d = c.d_a;
    }
    CHK(ew_scale_cols(c.b, X.rows, X.cols, X.p, X.ld, norms_d, true));
    int ny = 0;
    for (auto& Y : Ys) ny += Y.cols;
    // adjacent blocks (all of X followed by the new P in lobpcg_run's layout) act as ONE matrix
    std::vector<Mat> merged;
    if (Ys.size() > 1 && contiguous(Ys)) merged = {Mat{Ys[0].p, Ys[0].ld, Ys[0].rows, ny}};
    const std::vector<Mat>& Yl = merged.empty() ? Ys : merged;
    int niter = 1;
    std::vector<double> hb;     // landing zone of the (possibly deferred) fetch below: outlives every round of the loop
    for (;;) {
        // BYX = Y' X ; X -= Y BYX
        int off = 0;
        for (auto& Y : Yl) {
            if (Y.cols == 0) continue;
            CHK(c.mm('C', Y.cols, X.cols, X.rows, ONE, Y.p, Y.ld, X.p, X.ld, ZERO, c.BYX + off, ny));
            off += Y.cols;
        }
        CHK(c.reduce_c(c.BYX, (size_t)ny * X.cols));
        off = 0;
        for (auto& Y : Yl) {
            if (Y.cols == 0) continue;
            CHK(c.mm('N', X.rows, X.cols, Y.cols, MONE, Y.p, Y.ld, c.BYX + off, ny, ONE, X.p, X.ld));
            off += Y.cols;
        }
        // drop_small!  (the column sums of |BYX|^2 for the convergence test below ride on the same fetch: every host
        // synchronisation of this loop is a round of latency, in the batched multi-k driver a whole scheduling round)
        CHK(ew_colnorms(c.b, X.rows, X.cols, X.p, X.ld, c.d_a));
        CHK(c.reduce_norms(c.d_a, X.cols));
        const bool one_fetch = c.dstride > 0 && c.d_b == c.d_a + c.dstride && X.cols <= c.dstride;
        // Small blocks (the k-point workloads): the fetch does not get a synchronisation of its own -- it rides on the first
        // Cholesky status of the ortho!(X) that follows, which runs ahead speculatively (X is not touched before that status
        // is back).  In the common case (nothing to drop, not converged yet) that is one scheduling round less; if a column
        // has to be re-randomised or the loop is over, the speculative Gram matrix + factorisation are thrown away (a few
        // kiloflops here; for the n_G x 503 blocks of the large cells they would be milliseconds, so those keep the fetch).
        const bool defer = one_fetch && (int64_t)X.rows * X.cols <= DEFER_FETCH_MAX_ELEMS;
        // hb: norms [0, cols) and (one_fetch) column sums of |BYX|^2 at [dstride, dstride + cols)
        if (one_fetch) {
            CHK(ew_frob2(c.b, ny, X.cols, c.BYX, ny, c.d_b));
            hb.resize(c.dstride + X.cols);
            if (defer)
                CHK(dev_d2h_async(c.b, hb.data(), c.d_a, hb.size() * sizeof(double)));
            else
                CHK(d2h_sync(c.b, hb.data(), c.d_a, hb.size() * sizeof(double)));
        } else {
            hb.resize(X.cols);
            CHK(d2h_sync(c.b, hb.data(), c.d_a, hb.size() * sizeof(double)));
        }
        double byx2 = 0.0;
        std::vector<int> dropped;
        bool nonfinite = false;
        auto evaluate = [&]() {
            byx2 = 0.0;
            dropped.clear();
            if (one_fetch)
                for (int j = 0; j < X.cols; ++j) byx2 += hb[c.dstride + j];
            for (int j = 0; j < X.cols; ++j) {
                if (!std::isfinite(hb[j])) nonfinite = true;
                if (hb[j] <= tol) dropped.push_back(j);
            }
        };
        bool speculated = false;      // ortho!(X) of this round has already run (behind the deferred fetch)
        int ninner = 0;
        double growth = 1.0;
        if (defer) {
            const std::function<int()> hook = [&]() -> int {
                evaluate();
                if (nonfinite) return 3;
                if (!dropped.empty()) return 2;
                if (std::sqrt(byx2) < tol && niter > 1) return 1;
                return 0;
            };
            int hr = 0;
            CHK(ortho_X(c, X, tmp, tol, &ninner, &growth, true, 0, false, &hook, &hr));
            if (hr == 3) return DFTK_MI_NUM_NONFINITE;
            if (hr == 1) break;
            speculated = hr == 0;     // (hr == 2: columns to drop -- handled below, then ortho!(X) runs as usual)
        } else {
            evaluate();
            if (nonfinite) return DFTK_MI_NUM_NONFINITE;
        }
        for (int j : dropped) {
            CHK(randomize_column(c, X, j));
            Mat xj = X.cols_from(j, 1);
            int o2 = 0;
            // X[:, j] -= Y (Y' X[:, j])   (uses the tail of BYX as scratch: ny x 1 beyond the block)
            cd* scr = c.BYX + (int64_t)ny * X.cols;
            for (auto& Y : Yl) {
                if (Y.cols == 0) continue;
                CHK(c.mm('C', Y.cols, 1, X.rows, ONE, Y.p, Y.ld, xj.p, xj.ld, ZERO, scr + o2, ny));
                o2 += Y.cols;
            }
            CHK(c.reduce_c(scr, (size_t)ny));
            o2 = 0;
            for (auto& Y : Yl) {
                if (Y.cols == 0) continue;
                CHK(c.mm('N', X.rows, 1, Y.cols, MONE, Y.p, Y.ld, scr + o2, ny, ONE, xj.p, xj.ld));
                o2 += Y.cols;
            }
        }
        if (!one_fetch) CHK(frob2(c, Mat{c.BYX, ny, ny, X.cols}, &byx2));
        if (!speculated) {
            if (std::sqrt(byx2) < tol && niter > 1) break;
            CHK(ortho_X(c, X, tmp, tol, &ninner, &growth));
        }
        if (growth * EPS < tol) break;
        if (niter > 10) {
            // "Ortho(X, Y) is failing badly, falling back to SVD" (:307-314): X = U V' and return
            CHK(svd_polar(c, X, tmp, X.rows));
            c.n_svd += 1;
            return 0;
        }
        niter += 1;
    }
    return 0;
}

// ortho!(X, Y) of ortho_XY above for SMALL replicated matrices (the Ritz coefficient blocks cP against cX of a k-block with
// a handful of bands), on the host: on the device the loop is four host synchronisations per LOBPCG iteration -- in the
// lock-step multi-k driver four scheduling rounds of ~0.2 ms each for a few kiloflops.  Same algorithm, same tolerances,
// same estimates (normest = max |diag| + ||offdiag||_F as k_normest_upper); only the COMMON path: anything unusual (a
// column to re-randomise, a failing Cholesky factorisation, more than 10 rounds) returns 0 with X untouched and the caller
// takes the device path with its fallbacks.  X: n x m, Y: n x ny (column-major, leading dimension n), orthonormal columns.
typedef std::complex<double> zd;
static int host_ortho_small_impl(std::vector<zd>& Xio, int n, int m, const zd* Y, int ny, double tol, int* why, std::mt19937_64* gen,
                                 bool real);
// gen (optional): drop_small! (lobpcg_hyper_impl.jl:262-271, :291-294) is then taken on the host as well -- a column whose
// norm is <= tol after the projection is redrawn from `gen` (2 n normal deviates, as randomize_column draws them; imaginary
// parts zeroed for the real matrices of the Gamma-real iteration) and projected against Y.  Not a rare branch: the reference's
// cP = (cX - e)[:, Xn_indices] subtracts the identity from its first lenXn - newly_locked columns only, so with every newly
// locked vector the last columns of cP are plain columns of cX and vanish in the projection against cX (measured on the Al
// workload: one LOBPCG call in eight; without this the whole lock-step batch follows that k-point through the ~70 tiny
// launches and 5-6 scheduling rounds of the device path).
int host_ortho_small(std::vector<zd>& Xio, int n, int m, const zd* Y, int ny, double tol, std::mt19937_64* gen = nullptr,
                     bool real = false) {
    int why = 0;
    const int ok = host_ortho_small_impl(Xio, n, m, Y, ny, tol, &why, gen, real);
    static const bool trace = getenv("DFTK_MI_KBATCH_TRACE") != nullptr;
    if (!ok && trace) fprintf(stderr, "[host cP ortho] %d x %d against %d columns: device path (reason %d)\n", n, m, ny, why);
    return ok;
}
static int host_ortho_small_impl(std::vector<zd>& Xio, int n, int m, const zd* Y, int ny, double tol, int* why, std::mt19937_64* gen,
                                 bool real) {
    std::vector<zd> X = Xio, T((size_t)n * m), BYX((size_t)ny * m), O((size_t)m * m), R((size_t)m * m), Ri((size_t)m * m);
    auto colnorm = [&](int j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += std::norm(X[i + (size_t)j * n]);
        return std::sqrt(s);
    };
    for (int j = 0; j < m; ++j) {
        const double f = 1.0 / colnorm(j);
        for (int i = 0; i < n; ++i) X[i + (size_t)j * n] *= f;
    }
    auto normest = [&](const std::vector<zd>& U) {
        double dmax = 0.0, off = 0.0;
        for (int j = 0; j < m; ++j)
            for (int i = 0; i <= j; ++i) {
                if (i == j)
                    dmax = std::max(dmax, std::abs(U[i + (size_t)j * m]));
                else
                    off += std::norm(U[i + (size_t)j * m]);
            }
        return dmax + std::sqrt(off);
    };
    for (int niter = 1;; ++niter) {
        double byx2 = 0.0;
        for (int j = 0; j < m; ++j)
            for (int a = 0; a < ny; ++a) {
                zd s = 0.0;
                for (int i = 0; i < n; ++i) s += std::conj(Y[i + (size_t)a * n]) * X[i + (size_t)j * n];
                BYX[a + (size_t)j * ny] = s;
                byx2 += std::norm(s);
            }
        for (int j = 0; j < m; ++j)
            for (int a = 0; a < ny; ++a) {
                const zd s = BYX[a + (size_t)j * ny];
                for (int i = 0; i < n; ++i) X[i + (size_t)j * n] -= Y[i + (size_t)a * n] * s;
            }
        for (int j = 0; j < m; ++j) {
            const double nj = colnorm(j);
            if (!std::isfinite(nj) || (nj <= tol && !gen)) {   // non-finite (or drop_small! without a generator): device path
                *why = 1;
                return 0;
            }
            if (nj <= tol) {
                // drop_small!: X[:, j] = randn; X[:, j] -= Y (Y' X[:, j])
                std::normal_distribution<double> nd(0.0, 1.0);
                for (int i = 0; i < n; ++i) {
                    const double re = nd(*gen), im = nd(*gen);
                    X[i + (size_t)j * n] = zd(re, real ? 0.0 : im);
                }
                for (int a = 0; a < ny; ++a) {
                    zd s = 0.0;
                    for (int i = 0; i < n; ++i) s += std::conj(Y[i + (size_t)a * n]) * X[i + (size_t)j * n];
                    BYX[a + (size_t)j * ny] = s;     // (scratch: ||BYX|| of this round was taken above, as the device path does)
                }
                for (int a = 0; a < ny; ++a) {
                    const zd s = BYX[a + (size_t)j * ny];
                    for (int i = 0; i < n; ++i) X[i + (size_t)j * n] -= Y[i + (size_t)a * n] * s;
                }
            }
        }
        if (std::sqrt(byx2) < tol && niter > 1) break;
        // ortho!(X): Cholesky-QR passes until eps cond(R)^2 < tol
        double growth = 1.0;
        for (int pass = 0;; ++pass) {
            if (pass >= 30) {
                *why = 2;
                return 0;
            }
            for (int j = 0; j < m; ++j)
                for (int i = 0; i <= j; ++i) {
                    zd s = 0.0;
                    for (int r = 0; r < n; ++r) s += std::conj(X[r + (size_t)i * n]) * X[r + (size_t)j * n];
                    O[i + (size_t)j * m] = (i == j) ? zd(s.real(), 0.0) : s;
                }
            // upper Cholesky O = R^H R (column by column)
            std::fill(R.begin(), R.end(), zd(0.0));
            for (int j = 0; j < m; ++j) {
                for (int i = 0; i < j; ++i) {
                    zd s = O[i + (size_t)j * m];
                    for (int k = 0; k < i; ++k) s -= std::conj(R[k + (size_t)i * m]) * R[k + (size_t)j * m];
                    R[i + (size_t)j * m] = s / R[i + (size_t)i * m].real();
                }
                double d = O[j + (size_t)j * m].real();
                for (int k = 0; k < j; ++k) d -= std::norm(R[k + (size_t)j * m]);
                if (!(d > 0.0) || !std::isfinite(d)) {            // safe_cholesky's shift-and-retry: device path
                    *why = 3;
                    return 0;
                }
                R[j + (size_t)j * m] = std::sqrt(d);
            }
            // inverse of the upper triangular R (back substitution, column by column)
            std::fill(Ri.begin(), Ri.end(), zd(0.0));
            for (int j = 0; j < m; ++j) {
                Ri[j + (size_t)j * m] = 1.0 / R[j + (size_t)j * m].real();
                for (int i = j - 1; i >= 0; --i) {
                    zd s = 0.0;
                    for (int k = i + 1; k <= j; ++k) s += R[i + (size_t)k * m] * Ri[k + (size_t)j * m];
                    Ri[i + (size_t)j * m] = -s / R[i + (size_t)i * m].real();
                }
            }
            const double nR = normest(R), nI = normest(Ri);
            if (!std::isfinite(nR) || !std::isfinite(nI)) {
                *why = 4;
                return 0;
            }
            for (int j = 0; j < m; ++j)
                for (int i = 0; i < n; ++i) {
                    zd s = 0.0;
                    for (int k = 0; k <= j; ++k) s += X[i + (size_t)k * n] * Ri[k + (size_t)j * m];
                    T[i + (size_t)j * n] = s;
                }
            X.swap(T);
            growth *= nI;
            const double condR = nR * nI;
            if (EPS * condR * condR < tol) break;
        }
        if (growth * EPS < tol) break;
        if (niter > 10) {
            *why = 5;
            return 0;
        }
    }
    Xio.swap(X);
    return 1;
}
// largest coefficient block (elements) orthogonalised on the host: 32 KiB per k-block of a batched call
const size_t HOST_CP_MAX_ELEMS = 2048;

// C = sum_b Yb * coef[rows of b]   (LazyHcat * Matrix, lobpcg_hyper_impl.jl:124-132).  The active
// blocks are kept adjacent in memory (see the workspace layout in lobpcg_run), so this is ONE GEMM
// with k = sum of the block widths; the per-block loop only serves non-adjacent callers.
int hcat_mul(Ctx& c, const std::vector<Mat>& Ys, const cd* coef, int64_t ldcoef, int ncols, Mat C) {
    if (contiguous(Ys))
        return c.mm('N', Ys[0].rows, ncols, total_cols(Ys), ONE, Ys[0].p, Ys[0].ld, coef, ldcoef, ZERO, C.p,
                     C.ld);
    int64_t off = 0;
    bool first = true;
    for (auto& Y : Ys) {
        if (Y.cols == 0) continue;
        CHK(c.mm('N', Y.rows, ncols, Y.cols, ONE, Y.p, Y.ld, coef + off, ldcoef, first ? ZERO : ONE, C.p, C.ld));
        first = false;
        off += Y.cols;
    }
    return 0;
}

// small replicated matrices (Ritz coefficients) are orthogonalised with the same routines: no reductions there
struct NoComm {
    Ctx& c;
    dftk_mi_comm* saved;
    int saved_rf;
    explicit NoComm(Ctx& ctx) : c(ctx), saved(ctx.comm), saved_rf(ctx.rf) {
        c.comm = nullptr;
        c.replicated = saved != nullptr;   // (unsharded runs keep drawing from the one generator)
        c.small = true;
        c.rf = 0;
    }
    ~NoComm() {
        c.comm = saved;
        c.replicated = false;
        c.small = false;
        c.rf = saved_rf;
    }
};

}  // namespace

int lobpcg_ortho(dftk_mi_basis* b, int64_t n, int m, cd* X, int64_t ldx, int force_svd, int* n_chol, int* used_svd) {
    if (m <= 0) return 0;
    const size_t elems = (size_t)n * m + 4 * (size_t)m * m;
    void* buf = nullptr;
    HIPCHK(hipMalloc(&buf, elems * sizeof(cd) + 2 * (size_t)(m + 8) * sizeof(double)));
    Ctx c;
    c.kb = nullptr;
    c.b = b;
    cd* w = reinterpret_cast<cd*>(buf);
    cd* tmp = w;
    c.O = w + (size_t)n * m;
    c.Rw = c.O + (size_t)m * m;
    c.invR = c.Rw + (size_t)m * m;
    c.Vh = c.invR + (size_t)m * m;
    c.BYX = c.tmpS = nullptr;
    c.d_a = reinterpret_cast<double*>(c.Vh + (size_t)m * m);
    c.d_b = c.d_a + (m + 8);
    c.dstride = m + 8;
    c.rng.seed(0x9E3779B97F4A7C15ull);
    int nch = 0;
    double gr = 1.0;
    int st = ortho_X(c, Mat{X, ldx, n, m}, tmp, 2 * EPS, &nch, &gr, true, n, force_svd != 0);
    if (st == 0 && hipStreamSynchronize(b->stream) != hipSuccess) st = DFTK_MI_EHIP;
    hipFree(buf);
    if (n_chol) *n_chol = nch;
    if (used_svd) *used_svd = c.n_svd;
    return st;
}

std::atomic<int64_t> g_ax_reuse_count{0};     // calls that started from the kept A X (dftk_mi_ax_reuse_count)
static int lobpcg_run_general(dftk_mi_kblock* kb, int M, cd* Xp, int64_t ldX, double tol, int miniter, int maxiter,
                              int n_conv_check, int use_tpa, uint64_t seed, double* lambda_h, double* resid_h, int* n_iter_out,
                              int* converged_out, int64_t* n_matvec_out) {
    dftk_mi_basis* b = kb->basis;
    if (!(kb->n_G > 3 * (int64_t)M)) {
        dftk_set_error("The eigenproblem is too small (n_G=%lld, M=%d): N > 3M required", (long long)kb->n_G, M);
        return DFTK_MI_NUM_TOO_SMALL;
    }
    // a sharded block works on this rank's row slab of every n_G-sized array (see Ctx::reduce_*)
    dftk_mi_comm* comm = (kb->sh_comm && comm_size(kb->sh_comm) > 1) ? kb->sh_comm : nullptr;
    const int64_t row0 = comm ? (*kb->sh_rows)[comm_rank(comm)] : 0;
    // Gamma-real block: iterate on the half-sphere images of real-symmetric vectors (the caller's X is projected
    // onto that subspace on entry and expanded back to the full sphere on exit)
    const bool real_mode = kb->gr && kb->gr->on;
    const int64_t N = real_mode ? gamma_local_rows(kb) : comm ? (*kb->sh_rows)[comm_rank(comm) + 1] - row0 : kb->n_G;
    const double* kin = !use_tpa ? nullptr : real_mode ? kb->gr->d_kin_half + gamma_row0(kb) : kb->d_kin + row0;
    auto apply_H = [&](int nb, const cd* in, int64_t ldin, cd* out, int64_t ldout) -> int {
        if (real_mode) return gamma_apply_H(kb, 7, nb, in, ldin, out, ldout);
        return dftk_mi_apply_H(kb, nb, reinterpret_cast<const dftk_mi_cplx*>(in), ldin,
                               reinterpret_cast<dftk_mi_cplx*>(out), ldout);
    };
    if (N < 1) {
        dftk_set_error("sharded k-block: empty row slab on rank %d", comm_rank(comm));
        return DFTK_MI_EINVAL;
    }
    if (n_conv_check <= 0 || n_conv_check > M) n_conv_check = M;
    const double ortho_tol = 2 * EPS;

    // ---- workspace -------------------------------------------------------------------------
    const size_t blk = (size_t)N * M;                   // elements of one n_G x M block
    const size_t nbig = 14;                             // 2 x Y(3: X R P), 2 x AY(3), newR, tmp
    const size_t m3 = 3 * (size_t)M;
    const size_t small_elems = m3 * m3 * 2              // G, V
                               + m3 * M * 2             // cP, tmpS
                               + (size_t)M * M * 4      // O, Rw, invR, Vh
                               + (2 * (size_t)M + m3) * (M + 1);   // BYX (+1 scratch column)
    const size_t dbl = 9 * (size_t)(M + 8);
    const size_t need = (nbig * blk + small_elems) * sizeof(cd) + dbl * sizeof(double) + m3 * sizeof(int) + 1024;
    // A X kept from the last exit of this driver on this block (dftk_mi_kblock_reuse_AX): consumed or dropped by this call
    const bool reuse_asked = kb->ax_reuse_next;
    kb->ax_reuse_next = false;
    cd* ax_prev = kb->ax_keep;
    const int64_t ax_prev_ld = kb->ax_ld;
    const bool ax_shape_ok = ax_prev != nullptr && kb->ax_M == M && kb->ax_rows == N;
    kb->ax_keep = nullptr;
    if (need > kb->lob_bytes) {
        CHK(host_wait(b));
        if (kb->lob_buf) HIPCHK(hipFree(kb->lob_buf));
        kb->lob_buf = nullptr;
        kb->lob_bytes = 0;
        ax_prev = nullptr;               // (it lived in the buffer that has just gone)
        HIPCHK(dftk_scratch_malloc((void**)&kb->lob_buf, need));
        kb->lob_bytes = need;
    }
    bool reuse_ax = reuse_asked && ax_shape_ok && ax_prev != nullptr && kb->d_Vs != nullptr && kb->d_Vs_ax != nullptr;
    cd* w = kb->lob_buf;
    auto take = [&](size_t n) {
        cd* r = w;
        w += n;
        return r;
    };
    // Y = [X | R | P] and AY = [AX | AR | AP] live in n_G x 3M arrays.  X keeps columns [0, M);
    // the ACTIVE residual block sits at columns [M, M + nact) and the active search-direction block
    // right behind it at [M + nact, M + 2 nact), so that hcat(X_active, R, P) -- columns
    // [lo, M + 2 nact) -- is one contiguous matrix: Rayleigh-Ritz is one Gram GEMM and the block
    // updates one GEMM with k = 3 nact instead of per-block products.
    // There are TWO such pairs of arrays: an iteration reads Y/AY of the current pair and writes the new
    // X, AX, P, AP straight into the other one (no copy-back of n_G x M blocks); locked columns are kept
    // identical in both, then the roles swap.
    Mat Yb[2] = {Mat{take(3 * blk), N, N, 3 * M}, Mat{take(3 * blk), N, N, 3 * M}};
    Mat AYb[2] = {Mat{take(3 * blk), N, N, 3 * M}, Mat{take(3 * blk), N, N, 3 * M}};
    int cur = 0;
    // memory order [X | P | R]: hcat(X_active, P, R) is contiguous for the Rayleigh-Ritz products and
    // hcat(X, P) (all of X and the new P) for ortho!(R, [X P]); the block ORDER inside the hcat only
    // permutes the rows of the Ritz coefficient matrix.  Until P exists (iterations 0, 1) R sits right after X.
    auto Pblk = [&](const Mat& buf, int nact) { return buf.cols_from(M, nact); };
    auto Rblk = [&](const Mat& buf, int nact, bool has_p) { return buf.cols_from(M + (has_p ? nact : 0), nact); };
    Mat newR{take(blk), N, N, M};
    cd* tmp = take(blk);
    cd* G = take(m3 * m3);
    cd* V = take(m3 * m3);
    cd* cP = take(m3 * M);
    Ctx c;
    c.kb = kb;
    c.b = b;
    c.comm = comm;
    c.real_mode = real_mode;
    c.holds_g0 = real_mode && gamma_row0(kb) == 0;
    c.rf = real_mode ? DFTK_MI_GEMM_REAL : 0;
    c.tmpS = take(m3 * M);
    c.O = take((size_t)M * M);
    c.Rw = take((size_t)M * M);
    c.invR = take((size_t)M * M);
    c.Vh = take((size_t)M * M);
    c.BYX = take((2 * (size_t)M + m3) * (M + 1));
    double* dd = reinterpret_cast<double*>(w);
    c.d_a = dd;
    c.d_b = dd + (M + 8);
    c.dstride = M + 8;
    double* d_lam = dd + 2 * (M + 8);
    double* d_norms = dd + 3 * (M + 8);
    double* d_mk = dd + 4 * (M + 8);
    double* d_xx = dd + 5 * (M + 8);
    double* d_rn = dd + 6 * (M + 8);    // (slot 7 holds the final permutation)
    // every rank of a sharded block draws its own slab of a re-randomised column
    c.rng.seed((seed ? seed : 0x9E3779B97F4A7C15ull) + 0x632BE59BD9B4E019ull * (uint64_t)comm_rank(comm));
    c.rng_rep.seed((seed ? seed : 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull);
    Mat X = Yb[0].cols_from(0, M), AX = AYb[0].cols_from(0, M);   // views of the CURRENT pair (rebound on swap)
    kb->last_AX = real_mode ? nullptr : AX.p;

    Mat Xuser{Xp, ldX, N, M};
    if (real_mode)
        CHK(gamma_lobpcg_load(kb, M, Xp, ldX, X.p, X.ld, /*align=*/true));
    else
        CHK(ew_copy(b, N, M, Xuser.p, Xuser.ld, X.p, X.ld));
    std::vector<double> resid_history((size_t)M * (maxiter + 1), 0.0);
    auto RH = [&](int i, int it) -> double& { return resid_history[(size_t)i + (size_t)M * it]; };
    std::vector<double> full_lam(M, 0.0);

    // ---- X = ortho!(copy(X)); AX = A X --------------------------------------------------------
    {
        int nch;
        double gr;
        CHK(ortho_X(c, X, tmp, ortho_tol, &nch, &gr));
        // the kept A X follows X through ONE plain Cholesky-QR pass (X <- X inv(R), the common case: X comes back orthonormal
        // to round-off from the previous step); anything else (several passes, shifts, the SVD fallback) takes the full H X
        if (reuse_ax && !(nch == 1)) reuse_ax = false;
    }
    int64_t n_matvec = M;
    if (reuse_ax) {
        g_ax_reuse_count.fetch_add(1);
        // A_new X = (A_old X) inv(R) + (V_new - V_old) X: kinetic and nonlocal parts are those of the last call.  Saves the two
        // projector products of an H X (P' psi and P (D P' psi): 11.5 of 131 ms per late SCF step of the 1000-electron cell)
        // for one triangular product, one local-only application and two element-wise passes.
        dftk_mi_basis* bb = b;
        const size_t cube = (size_t)bb->nz * bb->ny * bb->nxp;
        if (!kb->d_dVs) HIPCHK(hipMalloc((void**)&kb->d_dVs, cube * sizeof(double)));
        CHK(ew_sub_real(bb, (int64_t)cube, kb->d_Vs, kb->d_Vs_ax, kb->d_dVs));
        // (A_old X) inv(R) -> AX (straight into place unless the kept block IS AX's storage: through newR then)
        if (ax_prev != AX.p) {
            CHK(c.mm('N', N, M, M, ONE, ax_prev, ax_prev_ld, c.invR, M, ZERO, AX.p, AX.ld, /*B upper triangular=*/2));
        } else {
            CHK(c.mm('N', N, M, M, ONE, ax_prev, ax_prev_ld, c.invR, M, ZERO, newR.p, newR.ld, /*B upper triangular=*/2));
            CHK(ew_copy(b, N, M, newR.p, newR.ld, AX.p, AX.ld));
        }
        double* const Vs_bound = kb->d_Vs;
        kb->d_Vs = kb->d_dVs;                                  // (the kernels take the pointer at launch)
        const int st_dv = real_mode ? gamma_apply_H(kb, 1, M, X.p, X.ld, newR.p, newR.ld)
                                    : dftk_mi_apply_H_parts(kb, 1, M, reinterpret_cast<const dftk_mi_cplx*>(X.p), X.ld,
                                                            reinterpret_cast<dftk_mi_cplx*>(newR.p), newR.ld);
        kb->d_Vs = Vs_bound;
        CHK(st_dv);
        CHK(ew_add(b, N, M, newR.p, newR.ld, AX.p, AX.ld));
        static const bool ax_check = getenv("DFTK_MI_AX_REUSE_CHECK") != nullptr;
        if (ax_check) {     // diagnostic: the full application beside it (costs what the path saves, and a synchronisation)
            CHK(apply_H(M, X.p, X.ld, newR.p, newR.ld));
            std::vector<cd> h1((size_t)N * M), h2((size_t)N * M);
            CHK(stream_sync(b));
            HIPCHK(hipMemcpy(h1.data(), AX.p, h1.size() * sizeof(cd), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(h2.data(), newR.p, h2.size() * sizeof(cd), hipMemcpyDeviceToHost));
            double worst = 0.0, nrm = 0.0;
            int wc = -1;
            for (int j = 0; j < M; ++j) {
                double d = 0.0, r = 0.0;
                for (int64_t i = 0; i < N; ++i) {
                    const cd a = h1[(size_t)i + (size_t)j * N], f = h2[(size_t)i + (size_t)j * N];
                    d += (a.x - f.x) * (a.x - f.x) + (a.y - f.y) * (a.y - f.y);
                    r += f.x * f.x + f.y * f.y;
                }
                nrm = std::max(nrm, std::sqrt(r));
                if (std::sqrt(d) > worst) {
                    worst = std::sqrt(d);
                    wc = j;
                }
            }
            fprintf(stderr, "[ax reuse check] M=%d N=%lld: max column error %.3e (column %d), max ||H x|| %.3e\n", M, (long long)N, worst,
                    wc, nrm);
        }
    } else {
        CHK(apply_H(M, X.p, X.ld, AX.p, AX.ld));
    }
    // (R is written at the end of iteration 0 and P at the end of iteration 1, before their first use)
    // lambda = Re(X'AX)/(X'X) column-wise.  The reference's "any(!isfinite, AX)" check (:380) rides on the same
    // pass: a non-finite entry of AX makes its column's dot non-finite (0 * inf and x * nan are nan).
    CHK(ew_coldots(b, N, M, X.p, X.ld, AX.p, AX.ld, c.d_a));
    CHK(ew_coldots(b, N, M, X.p, X.ld, X.p, X.ld, c.d_b));
    CHK(c.reduce_d(c.d_a, 2 * (size_t)(M + 8)));
    CHK(d2h(c, c.d_a, 2 * (M + 8)));
    for (int i = 0; i < M; ++i) {
        if (!std::isfinite(c.h[i])) {
            dftk_set_error("non-finite values in H*X");
            return DFTK_MI_NUM_NONFINITE;
        }
        full_lam[i] = c.h[i] / c.h[(M + 8) + i];
    }

    int nlocked = 0, niter = 0, lo = 0;
    int status_final = 0;
    bool finished = false;
    int final_iter = maxiter;
    int ncx = 0;   // columns of cX of the current iteration

    while (true) {
        const int nact = M - lo;
        Mat &Yc = Yb[cur], &AYc = AYb[cur], &Yn = Yb[cur ^ 1], &AYn = AYb[cur ^ 1];
        X = Yc.cols_from(0, M);
        AX = AYc.cols_from(0, M);
        Mat Xa = X.cols_from(lo), AXa = AX.cols_from(lo);
        Mat Ra = Rblk(Yc, nact, niter > 1), ARa = Rblk(AYc, nact, niter > 1), Pa = Pblk(Yc, nact), APa = Pblk(AYc, nact);
        // iteration 0 has no update: the "new" X is X itself; afterwards it is written into the other pair
        Mat nX = niter > 0 ? Yn.cols_from(lo, nact) : Xa, nAX = niter > 0 ? AYn.cols_from(lo, nact) : AXa;
        Mat nR = newR.cols_from(0, nact);
        std::vector<Mat> Ys, AYs;
        int nY = 0;
        cd* cX = V;
        bool host_cp = false;
        std::vector<zd> h_cX;
        if (niter > 0) {
            CHK(apply_H(nact, Ra.p, Ra.ld, ARa.p, ARa.ld));
            n_matvec += nact;
            if (niter > 1) {
                Ys = {Xa, Pa, Ra};
                AYs = {AXa, APa, ARa};
            } else {
                Ys = {Xa, Ra};
                AYs = {AXa, ARa};
            }
            nY = (int)Ys.size() * nact;
            // rayleigh_ritz: G = Y' AY (upper triangle), eigen, take the lowest nact
            if (contiguous(Ys) && contiguous(AYs)) {
                CHK(c.mm('C', nY, nY, N, ONE, Ys[0].p, Ys[0].ld, AYs[0].p, AYs[0].ld, ZERO, G, nY, /*upper=*/1));
                CHK(c.reduce_c(G, (size_t)nY * nY));
            } else {
                for (size_t ib = 0; ib < Ys.size(); ++ib)
                    for (size_t ia = 0; ia <= ib; ++ia)
                        CHK(c.mm('C', nact, nact, N, ONE, Ys[ia].p, Ys[ia].ld, AYs[ib].p, AYs[ib].ld, ZERO,
                                  G + (int64_t)ia * nact + (int64_t)ib * nact * nY, nY));
                CHK(c.reduce_c(G, (size_t)nY * nY));
            }
            CHK(ew_hermitize_upper(b, nY, G, nY));
            std::vector<double> wv(nY);
            {
                // only `vectors[:, 1:N]` and the N lowest values are used (lobpcg_hyper_impl.jl:146-150): the partial solver
                int st = dense_heev_lowest(b, nY, nact, G, nY, wv.data(), V, nY);
                if (st != 0) return st;
            }
            ncx = nact;
            // No re-orthonormalisation of the Ritz coefficient block: the reference does that only for the
            // CPU-LAPACK `syevr` of Julia < 1.12 (lobpcg_hyper_impl.jl:152-171); its GPU arrays take the generic
            // method (:145-151) and Julia >= 1.12 `syevd`, both without it.  The Jacobi eigenvectors are an
            // accumulated product of unitary rotations (||V'V - I|| ~ 1e-13, tests/test_gpu_kernels.py::test_heev).
            for (int i = 0; i < nact; ++i) full_lam[lo + i] = wv[i];
            // small coefficient blocks: cP is orthogonalised on the host further down (host_ortho_small); the copy of cX
            // rides on the residual fetch below
            host_cp = (size_t)nY * nact <= HOST_CP_MAX_ELEMS;
            if (host_cp) {
                h_cX.resize((size_t)nY * nact);
                CHK(dev_d2h_async(b, h_cX.data(), cX, h_cX.size() * sizeof(cd)));
            }
            CHK(hcat_mul(c, Ys, cX, nY, nact, nX));
            CHK(hcat_mul(c, AYs, cX, nY, nact, nAX));
        }

        // residuals
        CHK(h2d(b, d_lam, full_lam.data() + lo, nact * sizeof(double)));
        // residuals; the same pass over the new X yields precondprep!'s mean kinetic energies and <x,x>
        CHK(ew_residual(b, N, nact, nAX.p, nAX.ld, nX.p, nX.ld, d_lam, nR.p, nR.ld, d_norms, kin, d_mk, d_xx));
        // norms hold sqrt(local sums); mean_kin and <x,x> are plain sums (adjacent slots)   (no-ops for an unsharded block)
        CHK(c.reduce_norms(d_norms, nact));
        CHK(c.reduce_d(d_mk, (size_t)(M + 8) + nact));
        // norms, mean kinetic energies and <x,x> sit (M + 8) apart: ONE fetch; <x,x> is checked further down
        CHK(d2h(c, d_norms, 2 * (M + 8) + nact));
        std::vector<double> h_xx(c.h.begin() + 2 * (M + 8), c.h.begin() + 2 * (M + 8) + nact);
        for (int i = 0; i < nact; ++i) {
            if (!std::isfinite(c.h[i])) {
                dftk_set_error("non-finite residual norm in LOBPCG iteration %d", niter);
                return DFTK_MI_NUM_NONFINITE;
            }
            RH(nlocked + i, niter) = c.h[i];
        }
        // (preconditioning -- precondprep!(new_X); ldiv!(precon, new_R) -- is applied further down, to the columns
        //  that stay active only and straight into their place in the next iteration's Y)
        // locking
        const int prev_nlocked = nlocked;
        if (niter >= miniter) {
            for (int i = nlocked; i < M; ++i) {
                if (RH(i, niter) < tol)
                    nlocked += 1;
                else
                    break;
            }
        }
        const int tgt = niter > 0 ? (cur ^ 1) : cur;   // the pair that now holds the up-to-date X, AX
        if (nlocked >= n_conv_check) {
            cur = tgt;     // locked columns [0, lo) are identical in both pairs, [lo, M) were just written
            final_iter = niter;
            finished = true;
            break;
        }
        const int newly_locked = nlocked - prev_nlocked;
        const int lenXn = nact - newly_locked;   // == M - nlocked

        Mat nP = Pblk(Yb[tgt], lenXn), nAP = Pblk(AYb[tgt], lenXn);   // next iteration's P, AP: written in place
        if (niter > 0) {
            // cP = (cX - e)[:, newly_locked:], then orthogonalise against all of cX
            Mat cPm{cP, nY, nY, lenXn};
            bool cp_done = false;
            if (host_cp) {
                // (h_cX is valid: the residual fetch above was a synchronising call)
                std::vector<zd> h_cP(h_cX.begin() + (size_t)newly_locked * nY, h_cX.begin() + (size_t)(newly_locked + lenXn) * nY);
                for (int a = 0; a < lenXn - newly_locked; ++a)
                    if (2 * newly_locked + a < nY) h_cP[(size_t)(2 * newly_locked + a) + (size_t)a * nY] -= 1.0;
                if (host_ortho_small(h_cP, nY, lenXn, h_cX.data(), ncx, ortho_tol, &c.rng_rep, c.real_mode)) {
                    CHK(h2d(b, cP, h_cP.data(), h_cP.size() * sizeof(cd)));
                    cp_done = true;
                }
            }
            if (!cp_done) {
                CHK(ew_copy(b, nY, lenXn, cX + (int64_t)newly_locked * nY, nY, cP, nY));
                CHK(ew_sub_identity_shifted(b, nY, lenXn - newly_locked, cP, nY, 2 * newly_locked));
                std::vector<Mat> cXs = {Mat{cX, nY, nY, ncx}};
                NoComm replicated(c);
                CHK(ortho_XY(c, cPm, cXs, c.tmpS, ortho_tol));
            }
            CHK(hcat_mul(c, Ys, cP, nY, lenXn, nP));
            CHK(hcat_mul(c, AYs, cP, nY, lenXn, nAP));
        }
        // sanity: |<x,x> - 1| < sqrt(eps)
        for (int i = 0; i < nact; ++i)
            if (!(std::fabs(h_xx[i] - 1.0) < std::sqrt(EPS))) {
                dftk_set_error("LOBPCG is badly failing to keep the vectors normalized (column %d: %g; iteration %d, "
                               "%d locked, %d active, %d ranks)", lo + i, h_xx[i], niter, nlocked, nact, comm_size(comm));
                return DFTK_MI_NUM_NORMALIZATION;
            }
        // newly locked columns never change again: keep them identical in both pairs
        if (newly_locked > 0) {
            CHK(ew_copy(b, N, newly_locked, Yb[tgt].p + (int64_t)lo * N, N, Yb[tgt ^ 1].p + (int64_t)lo * N, N));
            CHK(ew_copy(b, N, newly_locked, AYb[tgt].p + (int64_t)lo * N, N, AYb[tgt ^ 1].p + (int64_t)lo * N, N));
        }
        // restrict to active
        lo = nlocked;
        cur = tgt;
        X = Yb[cur].cols_from(0, M);
        AX = AYb[cur].cols_from(0, M);
        Mat Rn = Rblk(Yb[cur], lenXn, niter > 0);   // next iteration's residual block (behind P once P exists)
        CHK(ew_tpa(b, N, lenXn, nR.p + (int64_t)newly_locked * nR.ld, nR.ld, Rn.p, Rn.ld, kin, d_mk + newly_locked,
                   d_rn));
        CHK(c.reduce_norms(d_rn, lenXn));
        std::vector<Mat> Zs = {X};
        if (niter > 0) Zs.push_back(nP);
        CHK(ortho_XY(c, Rn, Zs, tmp, ortho_tol, d_rn));
        static const bool dbg_check = getenv("DFTK_MI_LOBPCG_CHECK") != nullptr;
        if (dbg_check) {   // debugging aid: || [X P R]' [X P R] - I ||_max after the orthogonalisations of this iteration
            const int nc = M + (niter > 0 ? 2 : 1) * lenXn;
            CHK(c.mm('C', nc, nc, N, ONE, Yb[cur].p, N, Yb[cur].p, N, ZERO, G, nc));
            CHK(c.reduce_c(G, (size_t)nc * nc));
            std::vector<double> hg(2 * (size_t)nc * nc);
            CHK(d2h_sync(b, hg.data(), G, hg.size() * sizeof(double)));
            double worst[3][3] = {{0}};
            auto blk_of = [&](int j) { return j < M ? 0 : (niter > 0 && j < M + lenXn ? 1 : 2); };
            for (int j = 0; j < nc; ++j)
                for (int i = 0; i < nc; ++i) {
                    const double re = hg[2 * ((size_t)i + (size_t)j * nc)] - (i == j ? 1.0 : 0.0);
                    const double im = hg[2 * ((size_t)i + (size_t)j * nc) + 1];
                    double& w = worst[blk_of(i)][blk_of(j)];
                    w = std::max(w, std::sqrt(re * re + im * im));
                }
            fprintf(stderr, "[lobpcg-check rank %d] it %d locked %d act %d  XX %.1e XP %.1e XR %.1e PP %.1e PR %.1e RR %.1e\n",
                    comm_rank(comm), niter, nlocked, lenXn, worst[0][0], worst[0][1], worst[0][2], worst[1][1],
                    worst[1][2], worst[2][2]);
        }

        if (niter >= maxiter) break;
        niter += 1;
    }
    X = Yb[cur].cols_from(0, M);
    AX = AYb[cur].cols_from(0, M);
    kb->last_AX = real_mode ? nullptr : AX.p;   // (half-format blocks are not handed out)
    if (!finished) final_iter = maxiter;
    (void)status_final;

    // final_retval: sort by lambda if needed
    std::vector<int> perm(M);
    std::iota(perm.begin(), perm.end(), 0);
    bool sorted = std::is_sorted(full_lam.begin(), full_lam.end());
    if (!sorted) {
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int d) { return full_lam[a] < full_lam[d]; });
        int* d_perm = reinterpret_cast<int*>(dd + 7 * (M + 8));
        CHK(h2d(b, d_perm, perm.data(), M * sizeof(int)));
        CHK(ew_gather_cols(b, N, M, X.p, X.ld, d_perm, tmp, N));
        CHK(ew_copy(b, N, M, tmp, N, X.p, X.ld));
        CHK(ew_gather_cols(b, N, M, AX.p, AX.ld, d_perm, tmp, N));
        CHK(ew_copy(b, N, M, tmp, N, AX.p, AX.ld));
        CHK(stream_sync(b));
    }
    // hand the eigenvectors back to the caller's array
    if (real_mode)
        CHK(gamma_lobpcg_store(kb, M, X.p, X.ld, Xp, ldX));
    else
        CHK(ew_copy(b, N, M, X.p, X.ld, Xuser.p, Xuser.ld));
    double maxres = 0.0;
    for (int i = 0; i < M; ++i) {
        lambda_h[i] = full_lam[perm[i]];
        resid_h[i] = RH(perm[i], final_iter);
    }
    for (int i = 0; i < n_conv_check; ++i) maxres = std::max(maxres, resid_h[i]);
    // residual history of this call, rows permuted like the returned eigenpairs (final_retval :325-338)
    if (!kb->lob_hist) kb->lob_hist = new std::vector<double>();
    kb->lob_hist->assign((size_t)M * (final_iter + 1), 0.0);
    for (int it = 0; it <= final_iter; ++it)
        for (int i = 0; i < M; ++i) (*kb->lob_hist)[(size_t)i + (size_t)M * it] = RH(perm[i], it);
    kb->lob_hist_M = M;
    kb->lob_hist_iters = final_iter;
    kb->lob_n_svd = c.n_svd;
    *converged_out = (maxres < tol) ? 1 : 0;
    *n_iter_out = final_iter;
    *n_matvec_out = n_matvec;
    // keep A X (sorted like the returned X) and the potential it belongs to for a dftk_mi_kblock_reuse_AX start of the next call
    if (kb->d_Vs != nullptr && !batching()) {
        const size_t cube = (size_t)b->nz * b->ny * b->nxp;
        if (!kb->d_Vs_ax) HIPCHK(hipMalloc((void**)&kb->d_Vs_ax, cube * sizeof(double)));
        HIPCHK(hipMemcpyAsync(kb->d_Vs_ax, kb->d_Vs, cube * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        kb->ax_keep = AX.p;
        kb->ax_ld = AX.ld;
        kb->ax_M = M;
        kb->ax_rows = N;
    }
    return stream_sync(b);
}


// ---------------------------------------------------------------------------------------------------------------------
// Small blocks (the k-point workloads: n_G ~ 1e3, M <= 8) inside a batched multi-k call: the same algorithm with ONE host
// synchronisation per LOBPCG iteration.  The general driver above yields to the host 4-6 times per iteration (Cholesky
// statuses of the ortho! loops, the Ritz values, drop_small!'s norms, the residual norms) -- each a whole scheduling
// round of the lock-step batch (~0.2 ms whatever is in it; 27-31 rounds per SCF step of the
// 72-k-point Al workload).  Here
//   * ortho!(X) and ortho!(X, Y) are ONE kernel each (k_b_ortho: adaptive loops, safe_cholesky's shifts and all
//     estimates on the device; BOP_ORTHO),
//   * the Ritz values stay on the device (the residual pass reads them there) and reach the host with the residual
//     norms, as do the statuses of the orthogonalisation and of the eigensolver, the Rayleigh quotients of the start
//     block and the Ritz coefficients the host needs for cP,
//   * what the device cannot finish by itself (a column for drop_small! to re-randomise, the SVD fallbacks, a failed
//     eigensolver) is only DETECTED: the call then starts again on the general driver from the caller's start block,
//     which is untouched until the end (counted: DFTK_MI_KBATCH_TRACE).
// Control flow, tolerances, locking, the order of the blocks and every n_G-sized kernel are those of lobpcg_run_general.
std::atomic<int64_t> g_small_calls{0}, g_small_restarts{0};

bool lobpcg_small_eligible(const dftk_mi_kblock* kb, int M) {
    static const bool off = getenv("DFTK_MI_LOBPCG_SMALL") && atoi(getenv("DFTK_MI_LOBPCG_SMALL")) == 0;
    // (DFTK_MI_KBATCH_SEQUENTIAL -- read per call, like the recorder reads it -- runs every recorded operation through its
    //  original entry point, and the fused kernels have none: the general driver then)
    if (off || getenv("DFTK_MI_KBATCH_SEQUENTIAL") != nullptr || kb->sh_comm || (kb->gr && kb->gr->on)) return false;
    return M >= 1 && M <= 8 && kb->n_G * (int64_t)M <= DEFER_FETCH_MAX_ELEMS && kb->n_G > 3 * (int64_t)M;
}

static int rec_ortho(dftk_mi_basis* b, Mat X, const cd* Y, int64_t ldy, int ny, const double* norms_d, double tol, double* res4) {
    BOp o;
    o.b = b;
    o.type = BOP_ORTHO;
    o.n = X.rows;
    o.m = X.cols;
    o.k = ny;
    o.C = X.p;
    o.ldc = X.ld;
    o.A = Y;
    o.lda = ldy;
    o.W = norms_d;
    o.s0 = tol;
    o.host = res4;
    return batch_record(std::move(o));
}

static int lobpcg_run_small(dftk_mi_kblock* kb, int M, cd* Xp, int64_t ldX, double tol, int miniter, int maxiter,
                            int n_conv_check, int use_tpa, uint64_t seed, double* lambda_h, double* resid_h, int* n_iter_out,
                            int* converged_out, int64_t* n_matvec_out, bool* restart) {
    dftk_mi_basis* b = kb->basis;
    *restart = false;
    const int64_t N = kb->n_G;
    const double* kin = use_tpa ? kb->d_kin : nullptr;
    if (n_conv_check <= 0 || n_conv_check > M) n_conv_check = M;
    const double ortho_tol = 2 * EPS;
    // ---- workspace (the general driver's layout; the double scratch holds the Ritz values as well) ----
    const size_t blk = (size_t)N * M;
    const size_t nbig = 14;
    const size_t m3 = 3 * (size_t)M;
    const size_t small_elems = m3 * m3 * 2 + m3 * M * 2 + (size_t)M * M * 4 + (2 * (size_t)M + m3) * (M + 1);
    const int DS = (int)m3 + 8;                              // stride of the double slots
    const size_t dbl = 9 * (size_t)DS;
    const size_t need = (nbig * blk + small_elems) * sizeof(cd) + dbl * sizeof(double) + m3 * sizeof(int) + 1024;
    if (need > kb->lob_bytes) {
        CHK(host_wait(b));
        if (kb->lob_buf) HIPCHK(hipFree(kb->lob_buf));
        kb->lob_buf = nullptr;
        kb->lob_bytes = 0;
        HIPCHK(dftk_scratch_malloc((void**)&kb->lob_buf, need));
        kb->lob_bytes = need;
    }
    cd* w = kb->lob_buf;
    auto take = [&](size_t n) {
        cd* r = w;
        w += n;
        return r;
    };
    Mat Yb[2] = {Mat{take(3 * blk), N, N, 3 * M}, Mat{take(3 * blk), N, N, 3 * M}};
    Mat AYb[2] = {Mat{take(3 * blk), N, N, 3 * M}, Mat{take(3 * blk), N, N, 3 * M}};
    int cur = 0;
    auto Pblk = [&](const Mat& buf, int nact) { return buf.cols_from(M, nact); };
    auto Rblk = [&](const Mat& buf, int nact, bool has_p) { return buf.cols_from(M + (has_p ? nact : 0), nact); };
    Mat newR{take(blk), N, N, M};
    cd* tmp = take(blk);
    cd* G = take(m3 * m3);
    cd* V = take(m3 * m3);
    cd* cP = take(m3 * M);
    Ctx c;
    c.kb = kb;
    c.b = b;
    c.tmpS = take(m3 * M);
    c.O = take((size_t)M * M);
    c.Rw = take((size_t)M * M);
    c.invR = take((size_t)M * M);
    c.Vh = take((size_t)M * M);
    c.BYX = take((2 * (size_t)M + m3) * (M + 1));
    double* dd = reinterpret_cast<double*>(w);
    // [ d_a | d_b | d_norms | d_mk | d_xx ] are fetched together (5 slots); d_ev, d_rn, perm follow
    c.d_a = dd;
    c.d_b = dd + DS;
    c.dstride = DS;
    double* d_norms = dd + 2 * DS;
    double* d_mk = dd + 3 * DS;
    double* d_xx = dd + 4 * DS;
    double* d_ev = dd + 5 * DS;
    double* d_rn = dd + 6 * DS;
    c.rng.seed((seed ? seed : 0x9E3779B97F4A7C15ull));
    c.rng_rep.seed((seed ? seed : 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull);
    Mat X = Yb[0].cols_from(0, M), AX = AYb[0].cols_from(0, M);
    kb->last_AX = AX.p;
    Mat Xuser{Xp, ldX, N, M};
    CHK(ew_copy(b, N, M, Xuser.p, Xuser.ld, X.p, X.ld));
    std::vector<double> resid_history((size_t)M * (maxiter + 1), 0.0);
    auto RH = [&](int i, int it) -> double& { return resid_history[(size_t)i + (size_t)M * it]; };
    std::vector<double> full_lam(M, 0.0);
    std::vector<double> hf(5 * (size_t)DS);              // landing zone of the per-iteration fetch
    double o_res[4] = {0.0, 0.0, 0.0, 1.0};              // result of the orthogonalisation in flight
    bool o_pending = false;
    auto give_up = [&]() -> int {                         // a rare branch: the general driver takes the whole call
        *restart = true;
        return 0;
    };

    // ---- X = ortho!(copy(X)); AX = A X; Rayleigh quotients; residuals of iteration 0: ONE round ----
    CHK(rec_ortho(b, X, nullptr, 0, 0, nullptr, ortho_tol, o_res));
    o_pending = true;
    int64_t n_matvec = M;
    CHK(dftk_mi_apply_H(kb, M, reinterpret_cast<const dftk_mi_cplx*>(X.p), X.ld, reinterpret_cast<dftk_mi_cplx*>(AX.p), AX.ld));
    CHK(ew_coldots(b, N, M, X.p, X.ld, AX.p, AX.ld, c.d_a));
    batch_join_next();
    CHK(ew_coldots(b, N, M, X.p, X.ld, X.p, X.ld, c.d_b));

    int nlocked = 0, niter = 0, lo = 0;
    bool finished = false;
    int final_iter = maxiter;
    int ncx = 0;
    // state of the iteration whose device part is in flight (set by `head`, consumed after the fetch)
    int nY = 0;
    std::vector<Mat> Ys, AYs;
    std::vector<zd> h_cX;
    std::vector<double> wv(m3);
    int heev_st = 0;
    bool heev_pending = false;
    cd* cX = V;

    // device part of iteration `niter` up to the residuals (everything the general driver does before its locking decision)
    auto head = [&]() -> int {
        const int nact = M - lo;
        Mat &Yc = Yb[cur], &AYc = AYb[cur], &Yn = Yb[cur ^ 1], &AYn = AYb[cur ^ 1];
        X = Yc.cols_from(0, M);
        AX = AYc.cols_from(0, M);
        Mat Xa = X.cols_from(lo), AXa = AX.cols_from(lo);
        Mat Ra = Rblk(Yc, nact, niter > 1), ARa = Rblk(AYc, nact, niter > 1), Pa = Pblk(Yc, nact), APa = Pblk(AYc, nact);
        Mat nX = niter > 0 ? Yn.cols_from(lo, nact) : Xa, nAX = niter > 0 ? AYn.cols_from(lo, nact) : AXa;
        Mat nR = newR.cols_from(0, nact);
        Ys.clear();
        AYs.clear();
        nY = 0;
        if (niter > 0) {
            CHK(dftk_mi_apply_H(kb, nact, reinterpret_cast<const dftk_mi_cplx*>(Ra.p), Ra.ld, reinterpret_cast<dftk_mi_cplx*>(ARa.p),
                                ARa.ld));
            n_matvec += nact;
            if (niter > 1) {
                Ys = {Xa, Pa, Ra};
                AYs = {AXa, APa, ARa};
            } else {
                Ys = {Xa, Ra};
                AYs = {AXa, ARa};
            }
            nY = (int)Ys.size() * nact;
            // (the blocks are adjacent by construction of the layout: one Gram product)
            CHK(c.mm('C', nY, nY, N, ONE, Ys[0].p, Ys[0].ld, AYs[0].p, AYs[0].ld, ZERO, G, nY, /*upper=*/1));
            CHK(ew_hermitize_upper(b, nY, G, nY));
            {
                BOp o;
                o.b = b;
                o.type = BOP_HEEV;
                o.m = nY;
                o.C = G;
                o.ldc = nY;
                o.D = V;
                o.ldb = nY;
                o.host = wv.data();
                o.E = d_ev;
                o.status_out = &heev_st;
                heev_st = 0;
                heev_pending = true;
                CHK(batch_record(std::move(o)));
            }
            ncx = nact;
            h_cX.resize((size_t)nY * nact);
            CHK(dev_d2h_async(b, h_cX.data(), cX, h_cX.size() * sizeof(cd)));
            CHK(hcat_mul(c, Ys, cX, nY, nact, nX));
            batch_join_next();                               // X = Y cX and AX = AY cX: one launch
            CHK(hcat_mul(c, AYs, cX, nY, nact, nAX));
        }
        // residuals with the Ritz values as the device holds them (iteration 0: the Rayleigh quotients d_a / d_b)
        BOp r;
        r.b = b;
        r.type = BOP_RESIDUAL;
        r.n = N;
        r.m = nact;
        r.A = nAX.p;
        r.lda = nAX.ld;
        r.B = nX.p;
        r.ldb = nX.ld;
        r.W = niter > 0 ? d_ev : c.d_a;
        r.W3 = niter > 0 ? nullptr : c.d_b;
        r.C = nR.p;
        r.ldc = nR.ld;
        r.D = d_norms;
        r.W2 = kin;
        r.E = d_mk;
        r.F = d_xx;
        CHK(batch_record(std::move(r)));
        // THE synchronisation of the iteration
        CHK(d2h_sync(b, hf.data(), dd, hf.size() * sizeof(double)));
        return 0;
    };

    CHK(head());
    for (int i = 0; i < M; ++i) {
        if (!std::isfinite(hf[i])) {
            dftk_set_error("non-finite values in H*X");
            return DFTK_MI_NUM_NONFINITE;
        }
        full_lam[i] = hf[i] / hf[DS + i];
    }

    while (true) {
        const int nact = M - lo;
        // ---- what came back with the fetch ----
        if (o_pending) {
            o_pending = false;
            if (o_res[0] == 2.0) return DFTK_MI_NUM_NONFINITE;
            if (o_res[0] != 0.0) return give_up();
        }
        if (heev_pending) {
            heev_pending = false;
            if (heev_st == DFTK_MI_NUM_NONFINITE) return heev_st;
            if (heev_st != 0) return give_up();
            for (int i = 0; i < nact; ++i) full_lam[lo + i] = wv[i];
        }
        Mat nR = newR.cols_from(0, nact);
        const double* h_norms = hf.data() + 2 * DS;
        const double* h_xx = hf.data() + 4 * DS;
        for (int i = 0; i < nact; ++i) {
            if (!std::isfinite(h_norms[i])) {
                dftk_set_error("non-finite residual norm in LOBPCG iteration %d", niter);
                return DFTK_MI_NUM_NONFINITE;
            }
            RH(nlocked + i, niter) = h_norms[i];
        }
        // locking
        const int prev_nlocked = nlocked;
        if (niter >= miniter) {
            for (int i = nlocked; i < M; ++i) {
                if (RH(i, niter) < tol)
                    nlocked += 1;
                else
                    break;
            }
        }
        const int tgt = niter > 0 ? (cur ^ 1) : cur;
        if (nlocked >= n_conv_check) {
            cur = tgt;
            final_iter = niter;
            finished = true;
            break;
        }
        const int newly_locked = nlocked - prev_nlocked;
        const int lenXn = nact - newly_locked;
        Mat nP = Pblk(Yb[tgt], lenXn), nAP = Pblk(AYb[tgt], lenXn);
        if (niter > 0) {
            Mat cPm{cP, nY, nY, lenXn};
            bool cp_done = false;
            {
                std::vector<zd> h_cP(h_cX.begin() + (size_t)newly_locked * nY, h_cX.begin() + (size_t)(newly_locked + lenXn) * nY);
                for (int a = 0; a < lenXn - newly_locked; ++a)
                    if (2 * newly_locked + a < nY) h_cP[(size_t)(2 * newly_locked + a) + (size_t)a * nY] -= 1.0;
                if (host_ortho_small(h_cP, nY, lenXn, h_cX.data(), ncx, ortho_tol, &c.rng_rep, c.real_mode)) {
                    CHK(h2d(b, cP, h_cP.data(), h_cP.size() * sizeof(cd)));
                    cp_done = true;
                }
            }
            if (!cp_done) {
                CHK(ew_copy(b, nY, lenXn, cX + (int64_t)newly_locked * nY, nY, cP, nY));
                CHK(ew_sub_identity_shifted(b, nY, lenXn - newly_locked, cP, nY, 2 * newly_locked));
                std::vector<Mat> cXs = {Mat{cX, nY, nY, ncx}};
                NoComm replicated(c);
                CHK(ortho_XY(c, cPm, cXs, c.tmpS, ortho_tol));
            }
            CHK(hcat_mul(c, Ys, cP, nY, lenXn, nP));
            batch_join_next();                               // P = Y cP and AP = AY cP: one launch
            CHK(hcat_mul(c, AYs, cP, nY, lenXn, nAP));
        }
        for (int i = 0; i < nact; ++i)
            if (!(std::fabs(h_xx[i] - 1.0) < std::sqrt(EPS))) {
                dftk_set_error("LOBPCG is badly failing to keep the vectors normalized (column %d: %g; iteration %d, "
                               "%d locked, %d active, small-block driver)", lo + i, h_xx[i], niter, nlocked, nact);
                return DFTK_MI_NUM_NORMALIZATION;
            }
        if (newly_locked > 0) {
            CHK(ew_copy(b, N, newly_locked, Yb[tgt].p + (int64_t)lo * N, N, Yb[tgt ^ 1].p + (int64_t)lo * N, N));
            CHK(ew_copy(b, N, newly_locked, AYb[tgt].p + (int64_t)lo * N, N, AYb[tgt ^ 1].p + (int64_t)lo * N, N));
        }
        lo = nlocked;
        cur = tgt;
        X = Yb[cur].cols_from(0, M);
        AX = AYb[cur].cols_from(0, M);
        Mat Rn = Rblk(Yb[cur], lenXn, niter > 0);
        CHK(ew_tpa(b, N, lenXn, nR.p + (int64_t)newly_locked * nR.ld, nR.ld, Rn.p, Rn.ld, kin, d_mk + newly_locked, d_rn));
        // ortho!(R, [X P]): X (all M columns) and the new P are adjacent in the layout -- one kernel, status with the next fetch
        CHK(rec_ortho(b, Rn, X.p, X.ld, M + (niter > 0 ? lenXn : 0), d_rn, ortho_tol, o_res));
        o_pending = true;
        static const bool dbg_check = getenv("DFTK_MI_LOBPCG_CHECK") != nullptr;
        if (dbg_check) {   // (a diagnostic: costs a synchronisation of its own)
            const int nc = M + (niter > 0 ? 2 : 1) * lenXn;
            CHK(c.mm('C', nc, nc, N, ONE, Yb[cur].p, N, Yb[cur].p, N, ZERO, G, nc));
            std::vector<double> hg(2 * (size_t)nc * nc);
            CHK(d2h_sync(b, hg.data(), G, hg.size() * sizeof(double)));
            double worst = 0.0;
            for (int j = 0; j < nc; ++j)
                for (int i = 0; i < nc; ++i) {
                    const double re = hg[2 * ((size_t)i + (size_t)j * nc)] - (i == j ? 1.0 : 0.0);
                    const double im = hg[2 * ((size_t)i + (size_t)j * nc) + 1];
                    worst = std::max(worst, std::sqrt(re * re + im * im));
                }
            fprintf(stderr, "[lobpcg-check small] it %d locked %d act %d  ||[X P R]'[X P R] - I||_max %.1e (ortho status %g)\n", niter,
                    nlocked, lenXn, worst, o_res[0]);
        }
        if (niter >= maxiter) break;
        niter += 1;
        CHK(head());
    }
    if (o_pending && !finished) {
        // (maxiter reached with an orthogonalisation in flight whose result is never used)
        o_pending = false;
    }
    X = Yb[cur].cols_from(0, M);
    AX = AYb[cur].cols_from(0, M);
    kb->last_AX = AX.p;
    if (!finished) final_iter = maxiter;
    std::vector<int> perm(M);
    std::iota(perm.begin(), perm.end(), 0);
    const bool sorted = std::is_sorted(full_lam.begin(), full_lam.end());
    if (!sorted) {
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int d) { return full_lam[a] < full_lam[d]; });
        int* d_perm = reinterpret_cast<int*>(dd + 7 * DS);
        CHK(h2d(b, d_perm, perm.data(), M * sizeof(int)));
        CHK(ew_gather_cols(b, N, M, X.p, X.ld, d_perm, tmp, N));
        CHK(ew_copy(b, N, M, tmp, N, X.p, X.ld));
        CHK(ew_gather_cols(b, N, M, AX.p, AX.ld, d_perm, tmp, N));
        CHK(ew_copy(b, N, M, tmp, N, AX.p, AX.ld));
    }
    CHK(ew_copy(b, N, M, X.p, X.ld, Xuser.p, Xuser.ld));
    double maxres = 0.0;
    for (int i = 0; i < M; ++i) {
        lambda_h[i] = full_lam[perm[i]];
        resid_h[i] = RH(perm[i], final_iter);
    }
    for (int i = 0; i < n_conv_check; ++i) maxres = std::max(maxres, resid_h[i]);
    if (!kb->lob_hist) kb->lob_hist = new std::vector<double>();
    kb->lob_hist->assign((size_t)M * (final_iter + 1), 0.0);
    for (int it = 0; it <= final_iter; ++it)
        for (int i = 0; i < M; ++i) (*kb->lob_hist)[(size_t)i + (size_t)M * it] = RH(perm[i], it);
    kb->lob_hist_M = M;
    kb->lob_hist_iters = final_iter;
    kb->lob_n_svd = c.n_svd;
    *converged_out = (maxres < tol) ? 1 : 0;
    *n_iter_out = final_iter;
    *n_matvec_out = n_matvec;
    // (no synchronisation of its own: the copies above are queued on the fiber and run in the batch's closing round; the
    //  batched call returns only after the stream has drained)
    return 0;
}

// the fused kernel on a stand-alone block (tests; what lobpcg_run_small records per iteration)
extern "C" int dftk_mi_ortho_small(dftk_mi_basis* b, int64_t n, int m, dftk_mi_cplx* X, int64_t ldx, int ny, const dftk_mi_cplx* Y,
                                   int64_t ldy, const double* norms_d, double tol, double* res4_h) {
    if (!b || !X || n < 1 || m < 1 || m > 8 || ny < 0 || ny > 16 || (ny > 0 && !Y) || ldx < n || (ny > 0 && ldy < n) || !res4_h)
        return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    double clocks[8] = {0};
    std::vector<std::function<int()>> bodies;
    bodies.push_back([&]() -> int {
        BOp o;
        o.b = b;
        o.type = BOP_ORTHO;
        o.n = n;
        o.m = m;
        o.k = ny;
        o.C = X;
        o.ldc = ldx;
        o.A = Y;
        o.lda = ldy;
        o.W = norms_d;
        o.s0 = tol;
        o.host = res4_h;
        o.host2 = clocks;
        CHK(batch_record(std::move(o)));
        return dev_stream_sync(b);
    });
    std::vector<int> rets;
    const int st = batch_run(b, bodies, rets);
    if (getenv("DFTK_MI_ORTHO_CLOCKS"))       // shader clocks of the register-resident kernel's phases (diagnostic)
        fprintf(stderr, "[ortho clocks] n=%lld m=%d ny=%d rounds=%g: total %.0f = load %.0f + BYX %.0f + update/norms %.0f + Gram %.0f + "
                "Cholesky %.0f + X inv(R) %.0f + store %.0f\n", (long long)n, m, ny, res4_h[1], clocks[0], clocks[1], clocks[2], clocks[3],
                clocks[4], clocks[5], clocks[6], clocks[7]);
    return st != 0 ? st : rets[0];
}

extern "C" int dftk_mi_ax_reuse_count(int64_t* calls) {
    if (calls) *calls = g_ax_reuse_count.load();
    return 0;
}

extern "C" int dftk_mi_lobpcg_small_stats(int64_t* calls, int64_t* restarts) {
    if (calls) *calls = g_small_calls.load();
    if (restarts) *restarts = g_small_restarts.load();
    return 0;
}

int lobpcg_run(dftk_mi_kblock* kb, int M, cd* Xp, int64_t ldX, double tol, int miniter, int maxiter,
               int n_conv_check, int use_tpa, uint64_t seed, double* lambda_h, double* resid_h, int* n_iter_out,
               int* converged_out, int64_t* n_matvec_out) {
    if (lobpcg_small_eligible(kb, M)) {
        if (batching()) {
            bool restart = false;
            g_small_calls.fetch_add(1);
            const int st = lobpcg_run_small(kb, M, Xp, ldX, tol, miniter, maxiter, n_conv_check, use_tpa, seed, lambda_h, resid_h,
                                            n_iter_out, converged_out, n_matvec_out, &restart);
            if (!restart) return st;
            g_small_restarts.fetch_add(1);
            // the general driver below starts over from the caller's block, which the small-block driver has not written
        } else {
            // a single small block: a batch of one fiber (the fused kernels exist in their batched form only)
            int status = 0;
            cd* Xs[1] = {Xp};
            const int64_t lds[1] = {ldX};
            const uint64_t seeds[1] = {seed};
            dftk_mi_kblock* kbs[1] = {kb};
            const int st = lobpcg_run_multi(1, kbs, M, Xs, lds, tol, miniter, maxiter, n_conv_check, use_tpa, seeds, lambda_h, resid_h,
                                            n_iter_out, converged_out, n_matvec_out, &status);
            return st != 0 ? st : status;
        }
    }
    return lobpcg_run_general(kb, M, Xp, ldX, tol, miniter, maxiter, n_conv_check, use_tpa, seed, lambda_h, resid_h, n_iter_out,
                              converged_out, n_matvec_out);
}

// diagonalize_all_kblocks' loop over k-points (src/eigen/diag.jl:24-48) as ONE call: every k-block runs lobpcg_run as a
// fiber whose device operations are recorded and merged with its siblings' (batch.h)
int lobpcg_run_multi(int n_kb, dftk_mi_kblock* const* kbs, int M, cd* const* X, const int64_t* ldX, double tol, int miniter,
                     int maxiter, int n_conv_check, int use_tpa, const uint64_t* seeds, double* lambda_h, double* resid_h,
                     int* n_iter, int* converged, int64_t* n_matvec, int* status) {
    if (n_kb <= 0) return 0;
    dftk_mi_basis* b = kbs[0]->basis;
    for (int i = 0; i < n_kb; ++i) {
        if (kbs[i]->basis != b) {
            dftk_set_error("lobpcg_multi: all k-blocks must belong to ONE basis handle (one stream, one scratch pool)");
            return DFTK_MI_EINVAL;
        }
        if (kbs[i]->sh_comm || (kbs[i]->gr && kbs[i]->gr->on)) {
            dftk_set_error("lobpcg_multi: plane-wave sharded / Gamma-real k-blocks take dftk_mi_lobpcg");
            return DFTK_MI_EINVAL;
        }
        // the LOBPCG workspace of a k-block is (re)allocated with real synchronisations: do it before the fibers start
        for (int j = 0; j < i; ++j)
            if (kbs[j] == kbs[i]) {
                dftk_set_error("lobpcg_multi: k-block %d appears twice", i);
                return DFTK_MI_EINVAL;
            }
    }
    std::vector<std::function<int()>> bodies;
    for (int i = 0; i < n_kb; ++i)
        bodies.push_back([=]() {
            return lobpcg_run(kbs[i], M, X[i], ldX[i], tol, miniter, maxiter, n_conv_check, use_tpa,
                              seeds ? seeds[i] : 0, lambda_h + (size_t)i * M, resid_h + (size_t)i * M, n_iter + i,
                              converged + i, n_matvec + i);
        });
    std::vector<int> rets;
    const int st = batch_run(b, bodies, rets);
    for (int i = 0; i < n_kb; ++i) status[i] = rets[i];
    return st;
}
