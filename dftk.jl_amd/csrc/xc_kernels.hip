// xc_kernels.hip -- the local-potential pipeline of energy_hamiltonian on the device:
//   Hartree   src/terms/hartree.jl:50-59   rho -> FFT -> * 4 pi / |G|^2 -> E_H = 1/2 Re <V_H(G), rho(G)> -> irfft
//   XC (LDA)  src/terms/xc.jl:84-160       e_xc(rho), v_xc(rho) point by point (Slater exchange, VWN5 / PW92
//                                          correlation: the closed forms libxc evaluates for lda_x, lda_c_vwn, lda_c_pw)
//   sum       src/terms/operators.jl:213-222   V = V_loc + V_H + v_xc, handed to the k-blocks as ONE potential
//   energies  src/terms/local.jl:15-16 (E_loc = sum rho V_loc dvol), xc.jl:113 (E_xc = sum e_xc dvol)
// One pass over the cube for XC + the sum + two energy reductions; the Poisson multiply and the Hartree energy
// ride on the Fourier-space pass between the two cube FFTs (the library's own pruned pipeline with a full "sphere").
// Reductions: one partial per workgroup, summed on the host in a fixed order (bitwise reproducible).
#include "common.h"
#include <cmath>
#include <vector>

namespace dftk_xc {   // (named: kernels of an anonymous namespace lose their names in rocprofv3 traces)
const int XC_BLOCKS = 1024;

__device__ __forceinline__ double wave_sum_d(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_real_to_complex(int64_t n, const double* __restrict__ x, cd* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = make_double2(x[i], 0.0);
}

// c <- green * c ; partial[block] = sum green |c_old|^2
__global__ __launch_bounds__(256) void k_poisson(int64_t n, cd* __restrict__ c, const double* __restrict__ green,
                                                 double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const cd v = c[i];
        const double g = green[i];
        acc += g * (v.x * v.x + v.y * v.y);
        c[i] = make_double2(g * v.x, g * v.y);
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// closed forms; eps = energy per particle, returns e = rho * eps and v = d e / d rho
__device__ __forceinline__ void lda_x(double rho, double& e, double& v) {
    const double cx = -0.73855876638202240588;   // -3/4 (3/pi)^(1/3)
    const double r13 = cbrt(rho);
    e = cx * rho * r13;
    v = (4.0 / 3.0) * cx * r13;
}
__device__ __forceinline__ void lda_c_vwn(double rho, double& e, double& v) {
    const double A = 0.0310907, b = 3.72744, c = 12.9352, x0 = -0.10498;
    const double rs = cbrt(3.0 / (4.0 * M_PI * rho));
    const double x = sqrt(rs);
    const double X = x * x + b * x + c, X0 = x0 * x0 + b * x0 + c;
    const double Q = sqrt(4.0 * c - b * b);
    const double at = atan(Q / (2.0 * x + b));
    const double eps = A * (log(x * x / X) + 2.0 * b / Q * at -
                            b * x0 / X0 * (log((x - x0) * (x - x0) / X) + 2.0 * (b + 2.0 * x0) / Q * at));
    const double dat = -2.0 * Q / (Q * Q + (2.0 * x + b) * (2.0 * x + b));
    const double deps_dx = A * (2.0 / x - (2.0 * x + b) / X + 2.0 * b / Q * dat -
                                b * x0 / X0 * (2.0 / (x - x0) - (2.0 * x + b) / X + 2.0 * (b + 2.0 * x0) / Q * dat));
    e = rho * eps;
    v = eps - rs / 3.0 * deps_dx / (2.0 * x);
}
__device__ __forceinline__ void lda_c_pw(double rho, double& e, double& v) {
    const double a = 0.031091, a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
    const double rs = cbrt(3.0 / (4.0 * M_PI * rho));
    const double sq = sqrt(rs);
    const double den = 2.0 * a * (b1 * sq + b2 * rs + b3 * rs * sq + b4 * rs * rs);
    const double lg = log1p(1.0 / den);
    const double eps = -2.0 * a * (1.0 + a1 * rs) * lg;
    const double dden = 2.0 * a * (b1 / (2.0 * sq) + b2 + 1.5 * b3 * sq + 2.0 * b4 * rs);
    const double deps = -2.0 * a * a1 * lg + 2.0 * a * (1.0 + a1 * rs) * dden / (den * den + den);
    e = rho * eps;
    v = eps - rs / 3.0 * deps;
}

// ---- GGA (PBE): e(rho, sigma) with forward-mode derivatives d/d rho, d/d sigma carried through the closed forms
// (libxc's gga_x_pbe / gga_c_pbe on lda_c_pw_mod; Perdew, Burke, Ernzerhof 1996).  A dual number (v, dr, ds) makes
// the potential terms exact derivatives of exactly the energy expression -- no hand-derived formulas.
struct D3 {
    double v, dr, ds;
};
__device__ __forceinline__ D3 dc(double c) { return D3{c, 0.0, 0.0}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.v + b.v, a.dr + b.dr, a.ds + b.ds}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.v - b.v, a.dr - b.dr, a.ds - b.ds}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) {
    return D3{a.v * b.v, a.dr * b.v + a.v * b.dr, a.ds * b.v + a.v * b.ds};
}
__device__ __forceinline__ D3 operator/(D3 a, D3 b) {
    const double q = a.v / b.v;
    return D3{q, (a.dr - q * b.dr) / b.v, (a.ds - q * b.ds) / b.v};
}
__device__ __forceinline__ D3 operator*(double c, D3 a) { return D3{c * a.v, c * a.dr, c * a.ds}; }
__device__ __forceinline__ D3 operator+(double c, D3 a) { return D3{c + a.v, a.dr, a.ds}; }
__device__ __forceinline__ D3 dchain(D3 a, double f, double df) { return D3{f, df * a.dr, df * a.ds}; }
__device__ __forceinline__ D3 dsqrt(D3 a) { const double r = sqrt(a.v); return dchain(a, r, 0.5 / r); }
__device__ __forceinline__ D3 dcbrt(D3 a) { const double r = cbrt(a.v); return dchain(a, r, r / (3.0 * a.v)); }
__device__ __forceinline__ D3 dlog1p(D3 a) { return dchain(a, log1p(a.v), 1.0 / (1.0 + a.v)); }
__device__ __forceinline__ D3 dexpm1(D3 a) { const double e = expm1(a.v); return dchain(a, e, e + 1.0); }

__device__ __forceinline__ D3 gga_x_pbe(D3 rho, D3 sigma) {
    const double kappa = 0.8040, mu = 0.2195149727645171;
    const double cx = -0.73855876638202240588;                       // -3/4 (3/pi)^(1/3)
    const D3 kf = dcbrt(3.0 * M_PI * M_PI * rho);
    const D3 s2 = sigma / (4.0 * (kf * kf * rho * rho));
    const D3 r13 = dcbrt(rho);
    return cx * (rho * r13) * ((1.0 + kappa) + (-kappa * kappa) * (dc(1.0) / (kappa + mu * s2)));
}
__device__ __forceinline__ D3 gga_c_pbe(D3 rho, D3 sigma) {
    const double beta = 0.06672455060314922, gamma = 0.031090690869654895;   // (1 - ln 2) / pi^2
    const double a = 0.0310907, a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
    const D3 rs = dcbrt(dc(3.0 / (4.0 * M_PI)) / rho);
    const D3 sq = dsqrt(rs);
    const D3 den = 2.0 * a * (b1 * sq + b2 * rs + b3 * (rs * sq) + b4 * (rs * rs));
    const D3 eps = (-2.0 * a) * ((1.0 + a1 * rs) * dlog1p(dc(1.0) / den));
    const D3 kf = dcbrt(3.0 * M_PI * M_PI * rho);
    const D3 t2 = (M_PI / 16.0) * (sigma / (kf * rho * rho));
    const D3 A = dc(beta / gamma) / dexpm1((-1.0 / gamma) * eps);
    const D3 f1 = t2 + A * (t2 * t2);
    const D3 H = gamma * dlog1p((beta / gamma) * (f1 / (1.0 + A * f1)));
    return rho * (eps + H);
}

// e, de/drho, de/dsigma per grid point; points with rho <= threshold contribute nothing (libxc-style threshold)
__global__ __launch_bounds__(256) void k_gga(int64_t n, const double* __restrict__ rho, const double* __restrict__ sigma,
                                             int fun_mask, double threshold, double* __restrict__ e,
                                             double* __restrict__ vrho, double* __restrict__ vsigma) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        D3 acc = dc(0.0);
        const double r = rho[i];
        if (r > threshold) {
            const D3 dr = D3{r, 1.0, 0.0}, dsg = D3{sigma[i], 0.0, 1.0};
            if (fun_mask & 8) acc = acc + gga_x_pbe(dr, dsg);
            if (fun_mask & 16) acc = acc + gga_c_pbe(dr, dsg);
        }
        e[i] = acc.v;
        vrho[i] = acc.dr;
        vsigma[i] = acc.ds;
    }
}
// ---- collinear spin, LDA: e(rho_up, rho_down) with forward-mode derivatives; the two derivative slots of D3 carry
// d/d rho_up and d/d rho_down here.  Closed forms as libxc's polarised lda_x (spin-scaling relation), lda_c_pw (PW92 eq. 8
// interpolation in zeta, f''(0) = 1.709921) and lda_xc_teter93 (Goedecker, Teter, Hutter 1996: Pade coefficients linear in
// f(zeta)).  The unpolarised lda_xc_teter93 is the same form at rho_up = rho_down.
__device__ __forceinline__ D3 dpow43(D3 a) { return a * dcbrt(a); }
__device__ __forceinline__ D3 spin_fzeta(D3 ra, D3 rb, D3 rt) {
    const D3 xa = 2.0 * (ra / rt), xb = 2.0 * (rb / rt);
    return (1.0 / (2.5198420997897464 - 2.0)) * (dpow43(xa) + dpow43(xb) + dc(-2.0));   // 2^(4/3) - 2
}
__device__ __forceinline__ D3 pw92_G(D3 rs, D3 sq, double A, double a1, double b1, double b2, double b3, double b4) {
    const D3 den = 2.0 * A * (b1 * sq + b2 * rs + b3 * (rs * sq) + b4 * (rs * rs));
    return (-2.0 * A) * ((1.0 + a1 * rs) * dlog1p(dc(1.0) / den));
}
__device__ __forceinline__ D3 lda_x_spin(D3 ra, D3 rb) {
    const double cx = -0.73855876638202240588 * 1.2599210498948732;   // -3/4 (3/pi)^(1/3) 2^(1/3)
    return cx * (dpow43(ra) + dpow43(rb));
}
__device__ __forceinline__ D3 lda_c_pw_spin(D3 ra, D3 rb) {
    const D3 rt = ra + rb;
    const D3 fz = spin_fzeta(ra, rb, rt);
    const D3 z = (ra - rb) / rt;
    const D3 z2 = z * z, z4 = z2 * z2;
    const D3 rs = dcbrt(dc(3.0 / (4.0 * M_PI)) / rt);
    const D3 sq = dsqrt(rs);
    const D3 e0 = pw92_G(rs, sq, 0.031091, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
    const D3 e1 = pw92_G(rs, sq, 0.015545, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
    const D3 mac = pw92_G(rs, sq, 0.016887, 0.11125, 10.357, 3.6231, 0.88026, 0.49671);   // = -alpha_c
    return rt * (e0 - (1.0 / 1.709921) * (mac * fz * (dc(1.0) - z4)) + (e1 - e0) * (fz * z4));
}
__device__ __forceinline__ D3 lda_xc_teter93_spin(D3 ra, D3 rb) {
    const double a[4] = {0.4581652932831429, 2.217058676663745, 0.7405551735357053, 0.01968227878617998};
    const double da[4] = {0.119086804055547, 0.6157402568883345, 0.1574201515892867, 0.003532336663397157};
    const double bb[4] = {1.0, 4.504130959426697, 1.110667363742916, 0.02359291751427506};
    const double db[4] = {0.0, 0.2673612973836267, 0.2052004607777787, 0.004200005045691381};
    const D3 rt = ra + rb;
    const D3 fz = spin_fzeta(ra, rb, rt);
    const D3 rs = dcbrt(dc(3.0 / (4.0 * M_PI)) / rt);
    const D3 num = (a[0] + da[0] * fz) + rs * ((a[1] + da[1] * fz) + rs * ((a[2] + da[2] * fz) + rs * (a[3] + da[3] * fz)));
    const D3 den = rs * ((bb[0] + db[0] * fz) + rs * ((bb[1] + db[1] * fz) + rs * ((bb[2] + db[2] * fz) + rs * (bb[3] + db[3] * fz))));
    return dc(-1.0) * (rt * (num / den));
}
// fun_mask bits: 1 lda_x, 4 lda_c_pw, 32 lda_xc_teter93
__device__ __forceinline__ D3 lda_spin_sum(double rho_up, double rho_dn, int fun_mask) {
    const double floor_ = 1e-20;                       // a spin channel is never evaluated below this density
    const D3 ra = D3{rho_up > floor_ ? rho_up : floor_, 1.0, 0.0}, rb = D3{rho_dn > floor_ ? rho_dn : floor_, 0.0, 1.0};
    D3 acc = dc(0.0);
    if (rho_up + rho_dn <= 2.0 * floor_) return acc;
    if (fun_mask & 1) acc = acc + lda_x_spin(ra, rb);
    if (fun_mask & 4) acc = acc + lda_c_pw_spin(ra, rb);
    if (fun_mask & 32) acc = acc + lda_xc_teter93_spin(ra, rb);
    return acc;
}

// V = V_loc + V_H + v_xc ; partials: [0] sum e_xc, [1] sum rho V_loc
__global__ __launch_bounds__(256) void k_xc_sum(int64_t n, const double* __restrict__ rho, const cd* __restrict__ vh_cube,
                                                double vh_scale, const double* __restrict__ vloc, int fun_mask,
                                                const double* __restrict__ e_extra, const double* __restrict__ v_extra,
                                                double* __restrict__ V, double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc_xc = 0.0, acc_loc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double r = rho[i];
        double e = 0.0, v = 0.0;
        if (fun_mask != 0 && r > 1e-300) {
            double ei, vi;
            if (fun_mask & 1) { lda_x(r, ei, vi); e += ei; v += vi; }
            if (fun_mask & 2) { lda_c_vwn(r, ei, vi); e += ei; v += vi; }
            if (fun_mask & 4) { lda_c_pw(r, ei, vi); e += ei; v += vi; }
            if (fun_mask & 32) {               // lda_xc_teter93: the polarised form at rho_up = rho_down = rho / 2
                const D3 t = lda_spin_sum(0.5 * r, 0.5 * r, 32);
                e += t.v;
                v += t.dr;
            }
        }
        if (e_extra) {                       // GGA part: e(rho, sigma) and v_rho - 2 div(v_sigma grad rho), precomputed
            e += e_extra[i];
            v += v_extra[i];
        }
        acc_xc += e;
        double tot = v;
        if (vloc) {
            const double vl = vloc[i];
            acc_loc += r * vl;
            tot += vl;
        }
        if (vh_cube) tot += vh_scale * vh_cube[i].x;
        if (V) V[i] = tot;
    }
    const double s0 = block_sum(acc_xc, sh);
    const double s1 = block_sum(acc_loc, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s0;
        partial[XC_BLOCKS + blockIdx.x] = s1;
    }
}
// rho_tot as a complex cube (input of the Hartree pass of a collinear model)
__global__ __launch_bounds__(256) void k_total_to_complex(int64_t n, const double* __restrict__ up, const double* __restrict__ dn,
                                                          cd* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = make_double2(up[i] + dn[i], 0.0);
}
// V_s = V_loc + V_H[rho_tot] + v_xc,s(rho_up, rho_down), s = up, down ; partials: [0] sum e_xc, [1] sum rho_tot V_loc
__global__ __launch_bounds__(256) void k_xc_sum_spin(int64_t n, const double* __restrict__ up, const double* __restrict__ dn,
                                                     const cd* __restrict__ vh_cube, double vh_scale,
                                                     const double* __restrict__ vloc, int fun_mask, double* __restrict__ V_up,
                                                     double* __restrict__ V_dn, double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc_xc = 0.0, acc_loc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double ra = up[i], rb = dn[i];
        const D3 e = lda_spin_sum(ra, rb, fun_mask);
        acc_xc += e.v;
        double common = 0.0;
        if (vloc) {
            const double vl = vloc[i];
            acc_loc += (ra + rb) * vl;
            common += vl;
        }
        if (vh_cube) common += vh_scale * vh_cube[i].x;
        if (V_up) {
            V_up[i] = common + e.dr;
            V_dn[i] = common + e.ds;
        }
    }
    const double s0 = block_sum(acc_xc, sh);
    const double s1 = block_sum(acc_loc, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s0;
        partial[XC_BLOCKS + blockIdx.x] = s1;
    }
}
}  // namespace dftk_xc
using namespace dftk_xc;

// out[i] = scale * Re(c[i])
__global__ __launch_bounds__(256) void k_xc_real_part_scaled(int64_t n, const cd* __restrict__ c, double scale, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = scale * c[i].x;
}
// out[a n + i] = v[i] * g[a n + i], a = 0, 1, 2 (as complex numbers)
__global__ __launch_bounds__(256) void k_product3_to_complex(int64_t n, const double* __restrict__ v, const double* __restrict__ g,
                                                             cd* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double vi = v[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) out[a * n + i] = make_double2(vi * g[a * n + i], 0.0);
    }
}

int xc_gga_pointwise(dftk_mi_basis* b, int64_t n, const double* rho, const double* sigma, int fun_mask,
                     double threshold, double* e, double* vrho, double* vsigma) {
    hipLaunchKernelGGL(k_gga, dim3(XC_BLOCKS), dim3(256), 0, b->stream, n, rho, sigma, fun_mask, threshold, e, vrho,
                       vsigma);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

// cube_kernels.hip
int cube_ws_ensure(dftk_mi_basis* b, size_t bytes);
int cube_forward_real(dftk_mi_kblock* cube_kb, const double* f, const double* g, cd* tmp, cd* c_out);
int cube_gradient_multiply(dftk_mi_kblock* cube_kb, const double* recip_h, int alpha, const cd* c, cd* out, bool accumulate);
int cube_backward_real(dftk_mi_kblock* cube_kb, const cd* c, cd* tmp, double scale, double* out);
int cube_sigma(dftk_mi_basis* b, int64_t N, const double* gx, const double* gy, const double* gz, double* sigma);
int cube_axpy_real(dftk_mi_basis* b, int64_t N, const double* a, double scale, const cd* c, double* out);

// Collinear-spin LDA pipeline: rho = (rho_up, rho_down), two cubes; Hartree of the TOTAL density, V_loc, and the spin-resolved
// XC potential summed into (V_up, V_down); energies = Hartree, Xc, AtomicLocal (rho_tot V_loc).
int local_potential_collinear(dftk_mi_kblock* cube_kb, const double* rho, const double* vloc, const double* green,
                              int fun_mask, double* V_out, double* energies_h) {
    dftk_mi_basis* b = cube_kb->basis;
    const int64_t N = (int64_t)b->nx * b->ny * b->nz;
    if (cube_kb->n_G != N) {
        dftk_set_error("local_potential_collinear: the k-block must span the whole cube (n_G = %lld, N = %lld)",
                       (long long)cube_kb->n_G, (long long)N);
        return DFTK_MI_EINVAL;
    }
    if (fun_mask & ~(1 | 4 | 32)) {
        dftk_set_error("local_potential_collinear: spin-polarised forms exist for lda_x, lda_c_pw, lda_xc_teter93 only "
                       "(mask %d)", fun_mask);
        return DFTK_MI_EINVAL;
    }
    CHK(cube_ws_ensure(b, 2 * (size_t)N * sizeof(cd) + 3 * XC_BLOCKS * sizeof(double)));
    cd* c1 = reinterpret_cast<cd*>(b->dense_ws);
    cd* c2 = c1 + N;
    double* partial = reinterpret_cast<double*>(c2 + N);
    const double *up = rho, *dn = rho + N;
    const cd* vh = nullptr;
    if (green) {
        hipLaunchKernelGGL(k_total_to_complex, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, up, dn, c1);
        CHK(launch_fft_from_cube(cube_kb, c1, c2));
        hipLaunchKernelGGL(k_poisson, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, c2, green, partial + 2 * XC_BLOCKS);
        CHK(launch_ifft_to_cube(cube_kb, c2, c1));
        vh = c1;
    }
    hipLaunchKernelGGL(k_xc_sum_spin, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, up, dn, vh, 1.0 / (double)N, vloc, fun_mask,
                       V_out, V_out ? V_out + N : (double*)nullptr, partial);
    HIPCHK(hipGetLastError());
    if (!energies_h) return 0;      // potential only: asynchronous, like every call that returns no host data
    std::vector<double> hp(3 * XC_BLOCKS, 0.0);
    CHK(host_fetch(b, hp.data(), partial, (green ? 3 : 2) * XC_BLOCKS * sizeof(double)));
    double s3[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < XC_BLOCKS; ++i) s3[k] += hp[(size_t)k * XC_BLOCKS + i];
    const double dvol = b->volume / (double)N;
    energies_h[0] = green ? 0.5 * b->volume / ((double)N * (double)N) * s3[2] : 0.0;
    energies_h[1] = s3[0] * dvol;
    energies_h[2] = s3[1] * dvol;
    return 0;
}

// cube_kb: a k-block whose "sphere" is the whole cube (mapping = 0 .. N-1), i.e. the library's cube FFT.
// fun_mask may carry LDA bits (1, 2, 4: point-wise in the final pass) and GGA bits (8, 16): for the latter
//   grad rho = irfft(i G_a fft(rho))            (LibxcDensities, xc.jl:356-409)
//   sigma = |grad rho|^2, (e, v_rho, v_sigma) = point-wise PBE forms (k_gga)
//   v_xc = v_rho - 2 div(v_sigma grad rho),  div f = irfft(sum_a i G_a fft(f_a))   (xc.jl:140-150, :576-584)
// with G in cartesian coordinates from recip_h (row-major recip_lattice).  8 cube FFTs in total.
int local_potential_lda(dftk_mi_kblock* cube_kb, const double* recip_h, const double* rho, const double* vloc,
                        const double* green, int fun_mask, double threshold, double* V_out, double* energies_h) {
    dftk_mi_basis* b = cube_kb->basis;
    const int64_t N = (int64_t)b->nx * b->ny * b->nz;
    if (cube_kb->n_G != N) {
        dftk_set_error("local_potential: the k-block must span the whole cube (n_G = %lld, N = %lld)",
                       (long long)cube_kb->n_G, (long long)N);
        return DFTK_MI_EINVAL;
    }
    const int gga_mask = fun_mask & 24;
    // complex cubes c1, c2 (+ three more and 7 real cubes for GGA) + reduction partials in the basis' dense workspace
    const size_t need = (gga_mask ? 5 : 2) * (size_t)N * sizeof(cd) + (gga_mask ? 7 : 0) * (size_t)N * sizeof(double) +
                        3 * XC_BLOCKS * sizeof(double);
    CHK(cube_ws_ensure(b, need));
    cd* c1 = reinterpret_cast<cd*>(b->dense_ws);
    cd* c2 = c1 + N;
    cd* g3 = gga_mask ? c2 + N : nullptr;          // three cubes behind one another (one FFT pipeline for the three)
    double* rbase = reinterpret_cast<double*>(c2 + N + (gga_mask ? 3 * N : 0));
    double* partial = rbase + (gga_mask ? 7 * N : 0);
    double *e_g = nullptr, *v_g = nullptr;
    const cd* vh = nullptr;
    std::vector<double> hp(3 * XC_BLOCKS, 0.0);
    // F[rho] (unnormalised) is computed ONCE: the gradient multipliers read it, then the Poisson kernel works on it in place.
    // The three components of grad rho and of v_sigma grad rho go through ONE transform pipeline each (three cubes per
    // launch) instead of three: 18 launches fewer per call -- on the 36^3 ... 30 x 30 x 120 cubes of the k-point workloads
    // every launch of this chain is ~6 us whatever it does.
    if (gga_mask) {
        double* grad[3] = {rbase, rbase + N, rbase + 2 * N};
        double* sigma = rbase + 3 * N;
        e_g = rbase + 4 * N;
        double* vrho = rbase + 5 * N;
        double* vsig = rbase + 6 * N;
        CHK(cube_forward_real(cube_kb, rho, nullptr, c1, c2));                         // c2 = F[rho]
        for (int a = 0; a < 3; ++a) CHK(cube_gradient_multiply(cube_kb, recip_h, a, c2, g3 + a * N, false));   // i G_a F[rho]
        if (green) {
            hipLaunchKernelGGL(k_poisson, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, c2, green, partial + 2 * XC_BLOCKS);
            CHK(launch_ifft_to_cube(cube_kb, c2, c1));                                 // c1 = N * V_H(r): kept until the final sum
            vh = c1;
        }
        CHK(launch_ifft_to_cube(cube_kb, g3, g3, 3));                                  // in place: N grad rho (complex cubes)
        hipLaunchKernelGGL(k_xc_real_part_scaled, dim3(XC_BLOCKS), dim3(256), 0, b->stream, 3 * N, (const cd*)g3, 1.0 / (double)N,
                           grad[0]);                                                   // the three gradients are adjacent
        CHK(cube_sigma(b, N, grad[0], grad[1], grad[2], sigma));
        hipLaunchKernelGGL(k_gga, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, rho, sigma, gga_mask, threshold, e_g, vrho,
                           vsig);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_product3_to_complex, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, (const double*)vsig,
                           (const double*)grad[0], g3);                                // v_sigma d_a rho, a = 0, 1, 2
        CHK(launch_fft_from_cube(cube_kb, g3, g3, 3));                                 // in place
        for (int a = 0; a < 3; ++a) CHK(cube_gradient_multiply(cube_kb, recip_h, a, g3 + a * N, c2, a > 0));   // c2 (+)= i G_a ...
        CHK(launch_ifft_to_cube(cube_kb, c2, g3));                                     // g3[0] = N div(...)
        v_g = sigma;                                                                   // (sigma is dead by now)
        CHK(cube_axpy_real(b, N, vrho, -2.0 / (double)N, g3, v_g));
    } else if (green) {
        hipLaunchKernelGGL(k_real_to_complex, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, rho, c1);
        CHK(launch_fft_from_cube(cube_kb, c1, c2));                       // c2 = F[rho] (unnormalised)
        hipLaunchKernelGGL(k_poisson, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, c2, green, partial + 2 * XC_BLOCKS);
        CHK(launch_ifft_to_cube(cube_kb, c2, c1));                        // c1 = N * V_H(r)
        vh = c1;
    }
    hipLaunchKernelGGL(k_xc_sum, dim3(XC_BLOCKS), dim3(256), 0, b->stream, N, rho, vh, 1.0 / (double)N, vloc,
                       fun_mask & (7 | 32), (const double*)e_g, (const double*)v_g, V_out, partial);
    HIPCHK(hipGetLastError());
    if (!energies_h) return 0;      // potential only: asynchronous (no fetch, no synchronisation)
    HIPCHK(hipMemcpyAsync(hp.data(), partial, (green ? 3 : 2) * XC_BLOCKS * sizeof(double), hipMemcpyDeviceToHost,
                          b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    double s[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < XC_BLOCKS; ++i) s[k] += hp[(size_t)k * XC_BLOCKS + i];
    const double dvol = b->volume / (double)N;
    energies_h[0] = green ? 0.5 * b->volume / ((double)N * (double)N) * s[2] : 0.0;   // Hartree
    energies_h[1] = s[0] * dvol;                                                        // Xc
    energies_h[2] = s[1] * dvol;                                                        // AtomicLocal
    return 0;
}
