// cube_kernels.hip -- density-sized (cube) operations of the SCF glue on the device:
//   symmetrize_rho                src/symmetry.jl:282-357   accumulate_over_symmetries! + lowpass_for_symmetry! as ONE
//                                                           Fourier-space kernel between the library's two cube FFTs
//   KerkerMixing / DielectricMixing   src/scf/mixing.jl:54-105, :152-172   Fourier multipliers evaluated on the fly
//   DielectricModel (chi0)        src/scf/chi0models.jl:54-80
//   apply_kernel(::TermHartree)   src/terms/hartree.jl:68-81  (any precomputed real multiplier cube)
//   grad rho / divergence of GGA  src/terms/xc.jl:356-409, :576-584  (i G_a multipliers, cartesian)
// All of them are HBM-bound streaming passes over the cube; the FFTs are the library's own pruned pipeline run on a
// k-block whose "sphere" is the whole cube.  G vectors are never stored: a thread derives its integer G from its
// linear index (FFT order [0 .. (n-1)/2, -ceil((n-1)/2) .. -1], src/fft.jl:27-30) and, where needed, |B G|^2 from the
// nine entries of the reciprocal lattice passed by value.
#include "common.h"
#include <cmath>
#include <vector>

namespace dftk_cube {   // (named: kernels of an anonymous namespace lose their names in rocprofv3 traces)
const int CUBE_BLOCKS = 2048;
const int MAX_SYMM = 192;     // 48 point operations x up to 4 centring translations

struct Lat9 {                 // recip_lattice, row-major: G_cart[a] = sum_j B[3 a + j] G_red[j]
    double B[9];
};

__device__ __forceinline__ int signed_freq(int i, int n) { return i <= (n - 1) / 2 ? i : i - n; }
__device__ __forceinline__ bool in_range(int g, int n) { return g >= -((n - 1) - (n - 1) / 2) && g <= (n - 1) / 2; }
__device__ __forceinline__ int wrap(int g, int n) { return g < 0 ? g + n : g; }

__global__ __launch_bounds__(256) void k_r2c(int64_t n, const double* __restrict__ x, const double* __restrict__ y,
                                             cd* __restrict__ out) {
    // out = x (* y): real cube(s) -> complex cube
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = make_double2(y ? x[i] * y[i] : x[i], 0.0);
}

// out = scale * Re(c) (+ shift)
__global__ __launch_bounds__(256) void k_c2r(int64_t n, const cd* __restrict__ c, double scale, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = scale * c[i].x;
}

// accumulate_over_symmetries! (symmetry.jl:282-319) and lowpass_for_symmetry! (:323-343) fused, one thread per G:
//   out(G) = keep(G) * scale * sum_s [ S_s^-1 G on the grid ] e^{-2 pi i G.tau_s} in(S_s^-1 G),
//   keep(G) = prod_s [ S_s G on the grid ]          (only when lowpass != 0)
__global__ __launch_bounds__(256) void k_symmetrize(int nx, int ny, int nz, int n_sym, const int* __restrict__ invS,
                                                    const int* __restrict__ S, const double* __restrict__ tau,
                                                    const cd* __restrict__ in, cd* __restrict__ out, int lowpass,
                                                    double scale) {
    __shared__ int s_invS[MAX_SYMM * 9];
    __shared__ int s_S[MAX_SYMM * 9];
    __shared__ double s_tau[MAX_SYMM * 3];
    for (int t = threadIdx.x; t < n_sym * 9; t += 256) {
        s_invS[t] = invS[t];
        s_S[t] = S[t];
    }
    for (int t = threadIdx.x; t < n_sym * 3; t += 256) s_tau[t] = tau[t];
    __syncthreads();
    const int64_t N = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((int64_t)nx * ny));
        const int g0 = signed_freq(ix, nx), g1 = signed_freq(iy, ny), g2 = signed_freq(iz, nz);
        double ar = 0.0, ai = 0.0;
        bool keep = true;
        for (int s = 0; s < n_sym; ++s) {
            const int* M = s_invS + 9 * s;
            const int h0 = M[0] * g0 + M[1] * g1 + M[2] * g2;
            const int h1 = M[3] * g0 + M[4] * g1 + M[5] * g2;
            const int h2 = M[6] * g0 + M[7] * g1 + M[8] * g2;
            if (in_range(h0, nx) && in_range(h1, ny) && in_range(h2, nz)) {
                const cd v = in[wrap(h0, nx) + (int64_t)nx * (wrap(h1, ny) + (int64_t)ny * wrap(h2, nz))];
                const double t0 = s_tau[3 * s], t1 = s_tau[3 * s + 1], t2 = s_tau[3 * s + 2];
                if (t0 == 0.0 && t1 == 0.0 && t2 == 0.0) {
                    ar += v.x;
                    ai += v.y;
                } else {
                    double sn, cs;
                    sincospi(-2.0 * (g0 * t0 + g1 * t1 + g2 * t2), &sn, &cs);     // cis2pi(-G.tau)
                    ar += cs * v.x - sn * v.y;
                    ai += cs * v.y + sn * v.x;
                }
            }
            if (lowpass) {
                const int* F = s_S + 9 * s;
                keep = keep && in_range(F[0] * g0 + F[1] * g1 + F[2] * g2, nx) &&
                       in_range(F[3] * g0 + F[4] * g1 + F[5] * g2, ny) && in_range(F[6] * g0 + F[7] * g1 + F[8] * g2, nz);
            }
        }
        out[i] = keep ? make_double2(scale * ar, scale * ai) : make_double2(0.0, 0.0);
    }
}

enum { MULT_KERKER = 0, MULT_DIELECTRIC = 1, MULT_CHI0_DIELECTRIC = 2, MULT_ARRAY = 3, MULT_GRADIENT = 4 };

// c <- m(G) c  (MULT_GRADIENT: out <- i G_alpha c, or out += with accumulate) on the full cube in natural order
template <int KIND>
__global__ __launch_bounds__(256) void k_multiplier(int nx, int ny, int nz, Lat9 L, double p0, double p1, int alpha,
                                                    const double* __restrict__ marr, const cd* __restrict__ in,
                                                    cd* __restrict__ out, int accumulate) {
    const int64_t N = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((int64_t)nx * ny));
        const double g0 = signed_freq(ix, nx), g1 = signed_freq(iy, ny), g2 = signed_freq(iz, nz);
        const cd v = in[i];
        if (KIND == MULT_ARRAY) {
            const double m = marr[i];
            out[i] = make_double2(m * v.x, m * v.y);
            continue;
        }
        if (KIND == MULT_GRADIENT) {
            const double ga = L.B[3 * alpha] * g0 + L.B[3 * alpha + 1] * g1 + L.B[3 * alpha + 2] * g2;
            const cd w = make_double2(-ga * v.y, ga * v.x);                      // i G_a v
            out[i] = accumulate ? make_double2(out[i].x + w.x, out[i].y + w.y) : w;
            continue;
        }
        const double c0 = L.B[0] * g0 + L.B[1] * g1 + L.B[2] * g2;
        const double c1 = L.B[3] * g0 + L.B[4] * g1 + L.B[5] * g2;
        const double c2 = L.B[6] * g0 + L.B[7] * g1 + L.B[8] * g2;
        const double G2 = c0 * c0 + c1 * c1 + c2 * c2;
        double m;
        if (KIND == MULT_KERKER) {            // mixing.jl:61-72: G^2 / (kTF^2 + G^2), enforce_real!, DC copied from dF
            const bool unpaired = ((nx % 2 == 0) && ix == nx / 2) || ((ny % 2 == 0) && iy == ny / 2) ||
                                  ((nz % 2 == 0) && iz == nz / 2);
            m = unpaired ? 0.0 : G2 / (p0 * p0 + G2);
            if (i == 0) m = 1.0;              // d_rho .+= mean(dF) - mean(d_rho): the G = 0 coefficient of dF survives
        } else if (KIND == MULT_DIELECTRIC) { // mixing.jl:161-171 with C0 = 1 - eps_r
            const double C0 = 1.0 - p1;
            m = (p0 * p0 - C0 * G2) / (p1 * p0 * p0 - C0 * G2);
            if (i == 0) m = 1.0;
        } else {                              // chi0models.jl:66-77: C0 kTF^2 G^2 / 4 pi / (kTF^2 - C0 G^2)
            const double C0 = 1.0 - p1;
            m = C0 * p0 * p0 * G2 / (4.0 * M_PI) / (p0 * p0 - C0 * G2);
        }
        out[i] = make_double2(m * v.x, m * v.y);
    }
}

__global__ __launch_bounds__(256) void k_sigma(int64_t n, const double* __restrict__ gx, const double* __restrict__ gy,
                                               const double* __restrict__ gz, double* __restrict__ sigma) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        sigma[i] = gx[i] * gx[i] + gy[i] * gy[i] + gz[i] * gz[i];
}

// out = a + scale * Re(c)
__global__ __launch_bounds__(256) void k_axpy_real(int64_t n, const double* __restrict__ a, double scale,
                                                   const cd* __restrict__ c, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = a[i] + scale * c[i].x;
}

int check_cube(dftk_mi_kblock* cube_kb, int64_t* N_out) {
    dftk_mi_basis* b = cube_kb->basis;
    const int64_t N = (int64_t)b->nx * b->ny * b->nz;
    if (cube_kb->n_G != N) {
        dftk_set_error("cube operation: the k-block must span the whole cube (n_G = %lld, N = %lld)",
                       (long long)cube_kb->n_G, (long long)N);
        return DFTK_MI_EINVAL;
    }
    *N_out = N;
    return 0;
}

Lat9 make_lat(const double* recip_h) {
    Lat9 L;
    for (int i = 0; i < 9; ++i) L.B[i] = recip_h ? recip_h[i] : 0.0;
    return L;
}
}  // namespace dftk_cube
using namespace dftk_cube;

int cube_ws_ensure(dftk_mi_basis* b, size_t bytes) {
    if (bytes <= b->dense_ws_bytes) return 0;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->dense_ws) HIPCHK(hipFree(b->dense_ws));
    b->dense_ws = nullptr;
    b->dense_ws_bytes = 0;
    HIPCHK(dftk_scratch_malloc(&b->dense_ws, bytes));
    b->dense_ws_bytes = bytes;
    return 0;
}

// c_out = unnormalised forward FFT of the real cube f (* g); tmp: one complex cube of scratch
int cube_forward_real(dftk_mi_kblock* cube_kb, const double* f, const double* g, cd* tmp, cd* c_out) {
    dftk_mi_basis* b = cube_kb->basis;
    const int64_t N = (int64_t)b->nx * b->ny * b->nz;
    hipLaunchKernelGGL(k_r2c, dim3(CUBE_BLOCKS), dim3(256), 0, b->stream, N, f, g, tmp);
    HIPCHK(hipGetLastError());
    return launch_fft_from_cube(cube_kb, tmp, c_out);
}

// out = i G_alpha c  (cartesian component alpha of B G), optionally accumulated into out
int cube_gradient_multiply(dftk_mi_kblock* cube_kb, const double* recip_h, int alpha, const cd* c, cd* out,
                           bool accumulate) {
    dftk_mi_basis* b = cube_kb->basis;
    hipLaunchKernelGGL(k_multiplier<MULT_GRADIENT>, dim3(CUBE_BLOCKS), dim3(256), 0, b->stream, b->nx, b->ny, b->nz,
                       make_lat(recip_h), 0.0, 0.0, alpha, (const double*)nullptr, c, out, accumulate ? 1 : 0);
    HIPCHK(hipGetLastError());
    return 0;
}

// out = scale * Re(backward FFT of c); tmp: one complex cube of scratch (c is preserved)
int cube_backward_real(dftk_mi_kblock* cube_kb, const cd* c, cd* tmp, double scale, double* out) {
    dftk_mi_basis* b = cube_kb->basis;
    const int64_t N = (int64_t)b->nx * b->ny * b->nz;
    CHK(launch_ifft_to_cube(cube_kb, c, tmp));
    hipLaunchKernelGGL(k_c2r, dim3(CUBE_BLOCKS), dim3(256), 0, b->stream, N, tmp, scale, out);
    HIPCHK(hipGetLastError());
    return 0;
}

int cube_sigma(dftk_mi_basis* b, int64_t N, const double* gx, const double* gy, const double* gz, double* sigma) {
    hipLaunchKernelGGL(k_sigma, dim3(CUBE_BLOCKS), dim3(256), 0, b->stream, N, gx, gy, gz, sigma);
    HIPCHK(hipGetLastError());
    return 0;
}

int cube_axpy_real(dftk_mi_basis* b, int64_t N, const double* a, double scale, const cd* c, double* out) {
    hipLaunchKernelGGL(k_axpy_real, dim3(CUBE_BLOCKS), dim3(256), 0, b->stream, N, a, scale, c, out);
    HIPCHK(hipGetLastError());
    return 0;
}

// symmetrize_rho(basis, rho; symmetries, do_lowpass) for one spin component (symmetry.jl:346-357)
int cube_symmetrize(dftk_mi_kblock* cube_kb, int n_sym, const int32_t* S_h, const double* tau_h, int do_lowpass,
                    const double* rho_in, double* rho_out) {
    int64_t N;
    CHK(check_cube(cube_kb, &N));
    dftk_mi_basis* b = cube_kb->basis;
    if (n_sym < 1 || n_sym > MAX_SYMM) {
        dftk_set_error("symmetrize_rho: %d symmetry operations (1 .. %d supported)", n_sym, MAX_SYMM);
        return DFTK_MI_EINVAL;
    }
    bool all_one = true;
    std::vector<int> invS(9 * (size_t)n_sym), S(9 * (size_t)n_sym);
    for (int s = 0; s < n_sym; ++s) {
        const int32_t* M = S_h + 9 * s;
        // integer inverse through the adjugate (|det S| = 1 for a lattice symmetry)
        const int det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                        M[2] * (M[3] * M[7] - M[4] * M[6]);
        if (det != 1 && det != -1) {
            dftk_set_error("symmetrize_rho: symmetry %d has det S = %d (must be +-1)", s, det);
            return DFTK_MI_EINVAL;
        }
        const int adj[9] = {M[4] * M[8] - M[5] * M[7], M[2] * M[7] - M[1] * M[8], M[1] * M[5] - M[2] * M[4],
                            M[5] * M[6] - M[3] * M[8], M[0] * M[8] - M[2] * M[6], M[2] * M[3] - M[0] * M[5],
                            M[3] * M[7] - M[4] * M[6], M[1] * M[6] - M[0] * M[7], M[0] * M[4] - M[1] * M[3]};
        for (int t = 0; t < 9; ++t) {
            invS[9 * s + t] = adj[t] * det;
            S[9 * s + t] = M[t];
        }
        const bool one = M[0] == 1 && M[4] == 1 && M[8] == 1 && M[1] == 0 && M[2] == 0 && M[3] == 0 && M[5] == 0 &&
                         M[6] == 0 && M[7] == 0 && tau_h[3 * s] == 0.0 && tau_h[3 * s + 1] == 0.0 && tau_h[3 * s + 2] == 0.0;
        all_one = all_one && one;
    }
    if (all_one) {                                           // all(isone, symmetries): the density is returned as is
        if (rho_out != rho_in)
            HIPCHK(hipMemcpyAsync(rho_out, rho_in, N * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        return 0;
    }
    const size_t tab = (size_t)n_sym * (18 * sizeof(int) + 3 * sizeof(double));
    // the tables stay on the device between calls (keyed by their content): no upload, no synchronisation per SCF step
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t bytes) {
        const unsigned char* c = reinterpret_cast<const unsigned char*>(p);
        for (size_t i = 0; i < bytes; ++i) key = (key ^ c[i]) * 1099511628211ull;
    };
    mix(invS.data(), invS.size() * sizeof(int));
    mix(S.data(), S.size() * sizeof(int));
    mix(tau_h, 3 * (size_t)n_sym * sizeof(double));
    if (!(b->symm_tab && b->symm_key == key && b->symm_n == n_sym)) {
        HIPCHK(hipStreamSynchronize(b->stream));
        if (b->symm_tab) HIPCHK(hipFree(b->symm_tab));
        b->symm_tab = nullptr;
        HIPCHK(hipMalloc(&b->symm_tab, tab + 64));
        double* t_tau = reinterpret_cast<double*>(b->symm_tab);
        int* t_invS = reinterpret_cast<int*>(t_tau + 3 * (size_t)n_sym);
        int* t_S = t_invS + 9 * (size_t)n_sym;
        HIPCHK(hipMemcpy(t_tau, tau_h, 3 * (size_t)n_sym * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(t_invS, invS.data(), 9 * (size_t)n_sym * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(t_S, S.data(), 9 * (size_t)n_sym * sizeof(int), hipMemcpyHostToDevice));
        b->symm_key = key;
        b->symm_n = n_sym;
    }
    CHK(cube_ws_ensure(b, 2 * (size_t)N * sizeof(cd)));
    cd* c1 = reinterpret_cast<cd*>(b->dense_ws);
    cd* c2 = c1 + N;
    const double* d_tau = reinterpret_cast<const double*>(b->symm_tab);
    const int* d_invS = reinterpret_cast<const int*>(d_tau + 3 * (size_t)n_sym);
    const int* d_S = d_invS + 9 * (size_t)n_sym;
    CHK(cube_forward_real(cube_kb, rho_in, nullptr, c1, c2));                       // c2 = F[rho]
    hipLaunchKernelGGL(k_symmetrize, dim3(CUBE_BLOCKS), dim3(256), 0, b->stream, b->nx, b->ny, b->nz, n_sym, d_invS, d_S,
                       d_tau, c2, c1, do_lowpass, 1.0 / (double)n_sym);
    HIPCHK(hipGetLastError());
    CHK(cube_backward_real(cube_kb, c1, c2, 1.0 / (double)N, rho_out));
    return 0;            // asynchronous on the basis' stream (header conventions)
}

// out = irfft(m(G) fft(f)) for the closed-form multipliers of mixing.jl / chi0models.jl or a given multiplier cube
int cube_fourier_filter(dftk_mi_kblock* cube_kb, int kind, const double* recip_h, double p0, double p1,
                        const double* mult_d, const double* f, double* out) {
    int64_t N;
    CHK(check_cube(cube_kb, &N));
    dftk_mi_basis* b = cube_kb->basis;
    CHK(cube_ws_ensure(b, 2 * (size_t)N * sizeof(cd)));
    cd* c1 = reinterpret_cast<cd*>(b->dense_ws);
    cd* c2 = c1 + N;
    CHK(cube_forward_real(cube_kb, f, nullptr, c1, c2));
    const Lat9 L = make_lat(recip_h);
    const dim3 g(CUBE_BLOCKS), t(256);
    switch (kind) {
        case MULT_KERKER:
            hipLaunchKernelGGL(k_multiplier<MULT_KERKER>, g, t, 0, b->stream, b->nx, b->ny, b->nz, L, p0, p1, 0, mult_d, c2,
                               c2, 0);
            break;
        case MULT_DIELECTRIC:
            hipLaunchKernelGGL(k_multiplier<MULT_DIELECTRIC>, g, t, 0, b->stream, b->nx, b->ny, b->nz, L, p0, p1, 0, mult_d,
                               c2, c2, 0);
            break;
        case MULT_CHI0_DIELECTRIC:
            hipLaunchKernelGGL(k_multiplier<MULT_CHI0_DIELECTRIC>, g, t, 0, b->stream, b->nx, b->ny, b->nz, L, p0, p1, 0,
                               mult_d, c2, c2, 0);
            break;
        case MULT_ARRAY:
            hipLaunchKernelGGL(k_multiplier<MULT_ARRAY>, g, t, 0, b->stream, b->nx, b->ny, b->nz, L, p0, p1, 0, mult_d, c2,
                               c2, 0);
            break;
        default:
            return DFTK_MI_EINVAL;
    }
    HIPCHK(hipGetLastError());
    CHK(cube_backward_real(cube_kb, c2, c1, 1.0 / (double)N, out));
    return 0;            // asynchronous on the basis' stream (header conventions)
}
