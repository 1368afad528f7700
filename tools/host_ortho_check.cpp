// Host-only check of host_ortho_small (dftk.jl_amd/csrc/lobpcg.cpp): the ortho!(X, Y) loop that the LOBPCG driver runs
// on the HOST for the Ritz coefficient blocks of small k-blocks (lobpcg_hyper_impl.jl:271-323 with :216-261 inside).
// The function lives in the driver's anonymous namespace, so this translation unit INCLUDES the driver; everything
// else it refers to comes from the in-tree library.  No device call is made: runs on any host.
//   exit 0 + "host_ortho_check OK" when every case holds.
#include "../dftk.jl_amd/csrc/lobpcg.cpp"
#include <cstdio>

namespace {
typedef std::complex<double> Z;
double max_abs_gram_minus(const std::vector<Z>& A, int n, int ma, const std::vector<Z>& B, int mb, bool identity) {
    double worst = 0.0;
    for (int a = 0; a < ma; ++a)
        for (int b = 0; b < mb; ++b) {
            Z s = 0.0;
            for (int i = 0; i < n; ++i) s += std::conj(A[i + (size_t)a * n]) * B[i + (size_t)b * n];
            if (identity && a == b) s -= 1.0;
            worst = std::max(worst, std::abs(s));
        }
    return worst;
}
// orthonormal columns by modified Gram-Schmidt (twice)
void orthonormalise(std::vector<Z>& Q, int n, int m) {
    for (int rep = 0; rep < 2; ++rep)
        for (int j = 0; j < m; ++j) {
            for (int k = 0; k < j; ++k) {
                Z s = 0.0;
                for (int i = 0; i < n; ++i) s += std::conj(Q[i + (size_t)k * n]) * Q[i + (size_t)j * n];
                for (int i = 0; i < n; ++i) Q[i + (size_t)j * n] -= s * Q[i + (size_t)k * n];
            }
            double nn = 0.0;
            for (int i = 0; i < n; ++i) nn += std::norm(Q[i + (size_t)j * n]);
            nn = std::sqrt(nn);
            for (int i = 0; i < n; ++i) Q[i + (size_t)j * n] /= nn;
        }
}
}  // namespace

int main() {
    std::mt19937_64 gen(12345);
    std::normal_distribution<double> nd(0.0, 1.0);
    const double tol = 2 * EPS;
    int failures = 0;
    struct Case { int n, ny, m; bool real, coeff; };
    const Case cases[] = {{18, 6, 6, false, true}, {24, 8, 5, false, true}, {54, 18, 18, false, true}, {21, 7, 7, true, true},
                          {40, 10, 12, false, false}, {9, 3, 3, true, false}, {96, 32, 21, false, true}};
    for (const Case& cs : cases) {
        const int n = cs.n, ny = cs.ny, m = cs.m;
        // Y: the first ny columns of a random unitary (the Ritz coefficients cX are columns of the eigenvector matrix)
        std::vector<Z> Q((size_t)n * n);
        for (auto& q : Q) q = cs.real ? Z(nd(gen), 0.0) : Z(nd(gen), nd(gen));
        orthonormalise(Q, n, n);
        std::vector<Z> Y(Q.begin(), Q.begin() + (size_t)n * ny), X((size_t)n * m);
        if (cs.coeff) {   // cP = (cX - e): columns of Y with a unit entry subtracted, as the driver builds them
            for (int j = 0; j < m; ++j) {
                for (int i = 0; i < n; ++i) X[i + (size_t)j * n] = Y[i + (size_t)(j % ny) * n];
                X[(j % n) + (size_t)j * n] -= 1.0;
                if (j >= ny) X[((j * 7 + 3) % n) + (size_t)j * n] += 0.5;   // keep the columns independent
            }
        } else {
            for (auto& x : X) x = cs.real ? Z(nd(gen), 0.0) : Z(nd(gen), nd(gen));
        }
        const std::vector<Z> X0 = X;
        const int done = host_ortho_small(X, n, m, Y.data(), ny, tol);
        const double e_orth = max_abs_gram_minus(X, n, m, X, m, true), e_y = max_abs_gram_minus(Y, n, ny, X, m, false);
        // span: X0 - Y Y' X0 must lie in span(X):  (I - X X')(I - Y Y') X0 = 0
        double e_span = 0.0;
        {
            std::vector<Z> R = X0;
            for (int j = 0; j < m; ++j) {
                for (int a = 0; a < ny; ++a) {
                    Z s = 0.0;
                    for (int i = 0; i < n; ++i) s += std::conj(Y[i + (size_t)a * n]) * X0[i + (size_t)j * n];
                    for (int i = 0; i < n; ++i) R[i + (size_t)j * n] -= Y[i + (size_t)a * n] * s;
                }
            }
            std::vector<Z> R2 = R;
            for (int j = 0; j < m; ++j)
                for (int a = 0; a < m; ++a) {
                    Z s = 0.0;
                    for (int i = 0; i < n; ++i) s += std::conj(X[i + (size_t)a * n]) * R[i + (size_t)j * n];
                    for (int i = 0; i < n; ++i) R2[i + (size_t)j * n] -= X[i + (size_t)a * n] * s;
                }
            for (auto& r : R2) e_span = std::max(e_span, std::abs(r));
        }
        double im = 0.0;
        if (cs.real)
            for (auto& x : X) im = std::max(im, std::abs(x.imag()));
        const bool ok = done == 1 && e_orth < 1e-14 && e_y < 1e-14 && e_span < 1e-12 && im == 0.0;
        printf("n=%d ny=%d m=%d %s %s: done=%d |X'X-I|=%.1e |Y'X|=%.1e span=%.1e%s\n", n, ny, m, cs.real ? "real" : "complex",
               cs.coeff ? "cX-e" : "random", done, e_orth, e_y, e_span, ok ? "" : "   <-- FAILED");
        failures += ok ? 0 : 1;
    }
    // a column inside span(Y) has nothing left after the projection: drop_small! territory -> the host path must decline
    // (return 0) and leave X as it was, the device path with its re-randomisation takes over
    {
        const int n = 12, ny = 4, m = 3;
        std::vector<Z> Q((size_t)n * n);
        for (auto& q : Q) q = Z(nd(gen), nd(gen));
        orthonormalise(Q, n, n);
        std::vector<Z> Y(Q.begin(), Q.begin() + (size_t)n * ny), X((size_t)n * m);
        for (auto& x : X) x = Z(nd(gen), nd(gen));
        for (int i = 0; i < n; ++i) X[i + (size_t)1 * n] = 2.0 * Y[i] - 0.5 * Y[i + (size_t)2 * n];
        const std::vector<Z> X0 = X;
        const int done = host_ortho_small(X, n, m, Y.data(), ny, tol);
        const bool ok = done == 0 && X == X0;
        printf("column in span(Y): done=%d, X %s%s\n", done, X == X0 ? "untouched" : "MODIFIED", ok ? "" : "   <-- FAILED");
        failures += ok ? 0 : 1;
    }
    // the same with a generator: drop_small! on the host -- the column is redrawn, projected, and the block comes back
    // orthonormal and orthogonal to Y (the case of every LOBPCG iteration that locks a vector: the last columns of the
    // reference's cP are plain columns of cX); real blocks stay real
    for (int real = 0; real < 2; ++real) {
        const int n = 18, ny = 6, m = 4;
        std::vector<Z> Q((size_t)n * n);
        for (auto& q : Q) q = real ? Z(nd(gen), 0.0) : Z(nd(gen), nd(gen));
        orthonormalise(Q, n, n);
        std::vector<Z> Y(Q.begin(), Q.begin() + (size_t)n * ny), X((size_t)n * m);
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < n; ++i) X[i + (size_t)j * n] = Y[i + (size_t)(j + 2) * n];
        X[4 + 0 * (size_t)n] -= 1.0;      // cP = (cX - e)[:, Xn_indices] with newly_locked = 2: identity under the first two
        X[5 + 1 * (size_t)n] -= 1.0;      // columns only, the last two are columns of cX
        std::mt19937_64 g2(99);
        const int done = host_ortho_small(X, n, m, Y.data(), ny, tol, &g2, real != 0);
        const double e_orth = max_abs_gram_minus(X, n, m, X, m, true), e_y = max_abs_gram_minus(Y, n, ny, X, m, false);
        double im = 0.0;
        if (real)
            for (auto& x : X) im = std::max(im, std::abs(x.imag()));
        const bool ok = done == 1 && e_orth < 1e-14 && e_y < 1e-14 && im == 0.0;
        printf("columns of cX in cP (%s), host drop_small!: done=%d |X'X-I|=%.1e |Y'X|=%.1e%s\n", real ? "real" : "complex", done, e_orth,
               e_y, ok ? "" : "   <-- FAILED");
        failures += ok ? 0 : 1;
    }
    // linearly dependent columns: the Cholesky factorisation breaks down (or the estimate never settles) -> decline
    {
        const int n = 10, ny = 2, m = 3;
        std::vector<Z> Q((size_t)n * n);
        for (auto& q : Q) q = Z(nd(gen), nd(gen));
        orthonormalise(Q, n, n);
        std::vector<Z> Y(Q.begin(), Q.begin() + (size_t)n * ny), X((size_t)n * m);
        for (auto& x : X) x = Z(nd(gen), nd(gen));
        for (int i = 0; i < n; ++i) X[i + (size_t)2 * n] = X[i] + X[i + (size_t)n];
        const std::vector<Z> X0 = X;
        const int done = host_ortho_small(X, n, m, Y.data(), ny, tol);
        const bool ok = done == 0 && X == X0;
        printf("dependent columns: done=%d, X %s%s\n", done, X == X0 ? "untouched" : "MODIFIED", ok ? "" : "   <-- FAILED");
        failures += ok ? 0 : 1;
    }
    if (failures) {
        printf("host_ortho_check: %d case(s) FAILED\n", failures);
        return 1;
    }
    printf("host_ortho_check OK\n");
    return 0;
}
