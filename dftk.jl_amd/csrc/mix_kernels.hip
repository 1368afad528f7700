// mix_kernels.hip -- the density-sized solvers of the SCF glue behind the C ABI (remainder):
//   Anderson acceleration          src/scf/anderson.jl:36-130 (history on the device, the m x m least-squares problem on the
//                                  host from inner products: ONE reduction kernel + ONE fused update kernel per step)
//   chi0 mixing                    src/scf/mixing.jl:228-290: GMRES solve of (1 - chi0 vc) d_rho = dF with the RPA kernel
//                                  (hartree.jl:68-81), LdosModel / DielectricModel (chi0models.jl:21-80); restarted GMRES
//                                  as KrylovKit's linsolve (krylovdim 30, tol = max(1e-12, reltol |b|), zero start)
// Every vector is a cube (or two stacked cubes with collinear spin) of doubles in HBM.  These are launch-latency problems
// on the k-point workloads (36^3 cubes: every kernel is microseconds) and plain HBM streams on the large cells; what
// counts is launches and host synchronisations per call, so every pass fuses what the data flow allows and all inner
// products of a step travel to the host together: block partial sums are written straight into the basis' pinned,
// device-visible landing zone (b->h_fetch) and summed on the host in block order -- deterministic, one stream
// synchronisation per Krylov step / Anderson step, no device -> host blit.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

int cube_ws_ensure(dftk_mi_basis* b, size_t bytes);   // cube_kernels.hip

namespace dftk_mix {
const int MB = 128;          // blocks of every reduction kernel (partials are summed in block order)
const int MT = 256;
const int MAXK = 32;         // Krylov vectors / Anderson history entries a kernel takes by value

struct Lat9 {
    double B[9];
};
struct PtrTable {
    const double* p[MAXK];
    double c[MAXK];
};

__device__ __forceinline__ double m_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// block-wide sum (MT = 256 threads); result valid in every thread
__device__ __forceinline__ double m_block_sum(double v, double* sh /* [8] */) {
    v = m_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
// sum of the MB block partials part[blk * stride + q] in block order; every thread gets the result
__device__ __forceinline__ double m_sum_partials(const double* part, int stride, int q, double* sh) {
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < MB; ++i) t += part[(size_t)i * stride + q];
        sh[7] = t;
    }
    __syncthreads();
    return sh[7];
}
__device__ __forceinline__ int signed_freq(int i, int n) { return i <= (n - 1) / 2 ? i : i - n; }

// c = sum over the components of x (as a complex cube): the total density the Hartree kernel acts on
__global__ __launch_bounds__(MT) void k_total_r2c(int64_t N, int ncomp, const double* __restrict__ x, cd* __restrict__ c) {
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < N; i += (int64_t)gridDim.x * MT) {
        double t = x[i];
        if (ncomp == 2) t += x[N + i];
        c[i] = make_double2(t, 0.0);
    }
}
// c <- m .* c with a real multiplier cube (poisson Green's function), or out <- chi0_dielectric(G) c
__global__ __launch_bounds__(MT) void k_mult_array(int64_t N, const double* __restrict__ m, cd* __restrict__ c) {
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < N; i += (int64_t)gridDim.x * MT) {
        const double f = m[i];
        cd v = c[i];
        v.x *= f;
        v.y *= f;
        c[i] = v;
    }
}
__global__ __launch_bounds__(MT) void k_mult_chi0_dielectric(int nx, int ny, int nz, Lat9 L, double kTF, double eps_r,
                                                             const cd* __restrict__ in, cd* __restrict__ out) {
    const int64_t N = (int64_t)nx * ny * nz;
    const double C0 = 1.0 - eps_r;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < N; i += (int64_t)gridDim.x * MT) {
        const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((int64_t)nx * ny));
        const double g0 = signed_freq(ix, nx), g1 = signed_freq(iy, ny), g2 = signed_freq(iz, nz);
        const double c0 = L.B[0] * g0 + L.B[1] * g1 + L.B[2] * g2;
        const double c1 = L.B[3] * g0 + L.B[4] * g1 + L.B[5] * g2;
        const double c2 = L.B[6] * g0 + L.B[7] * g1 + L.B[8] * g2;
        const double G2 = c0 * c0 + c1 * c1 + c2 * c2;
        const double m = C0 * kTF * kTF * G2 / (4.0 * M_PI) / (kTF * kTF - C0 * G2);      // chi0models.jl:66-77
        const cd v = in[i];
        out[i] = make_double2(m * v.x, m * v.y);
    }
}
// dV = scale Re(c); block partials {sum dV, sum_comp sum ldos_comp dV}
__global__ __launch_bounds__(MT) void k_dv_partials(int64_t N, int ncomp, const cd* __restrict__ c, double scale,
                                                    const double* __restrict__ ldos, double* __restrict__ dV,
                                                    double* __restrict__ part) {
    __shared__ double sh[8];
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < N; i += (int64_t)gridDim.x * MT) {
        const double v = scale * c[i].x;
        dV[i] = v;
        s0 += v;
        if (ldos) {
            double l = ldos[i];
            if (ncomp == 2) l += ldos[N + i];
            s1 += l * v;
        }
    }
    const double t0 = m_block_sum(s0, sh), t1 = m_block_sum(s1, sh);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = t0;
        part[2 * blockIdx.x + 1] = t1;
    }
}
// out = x - chi0 (dV - mean dV) for the LDOS model (chi0models.jl:21-45: chi0 dV = ldos dEF - ldos dV with
// dEF = <ldos, dV> dvol / tdos) and / or the dielectric model (diel = its filtered dV, real part of a complex cube);
// block partials of sum(out) in part_out.  part_in: the partials of k_dv_partials.
__global__ __launch_bounds__(MT) void k_apply_chi0(int64_t N, int ncomp, const double* __restrict__ x,
                                                   const double* __restrict__ dV, const double* __restrict__ ldos,
                                                   const double* __restrict__ part_in, double dvol, double tdos,
                                                   const cd* __restrict__ diel, double diel_scale, double* __restrict__ out,
                                                   double* __restrict__ part_out) {
    __shared__ double sh[8];
    const double meanV = m_sum_partials(part_in, 2, 0, sh) / (double)N;
    double deF = 0.0;
    if (ldos) deF = m_sum_partials(part_in, 2, 1, sh) * dvol / tdos - meanV;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < N; i += (int64_t)gridDim.x * MT) {
        const double v = dV[i] - meanV;
        const double dl = diel ? diel_scale * diel[i].x : 0.0;
        for (int cpt = 0; cpt < ncomp; ++cpt) {
            double o = x[(int64_t)cpt * N + i];
            if (ldos) {
                const double l = ldos[(int64_t)cpt * N + i];
                o = o - (l * deF - l * v);          // drho + alpha (ldos dEF - ldos dV), alpha = -1
            }
            o -= dl;
            out[(int64_t)cpt * N + i] = o;
            s += o;
        }
    }
    const double t = m_block_sum(s, sh);
    if (threadIdx.x == 0) part_out[blockIdx.x] = t;
}
// w <- w - mean (mean from the MB partials of sum(w), when part_mean != null), then the block partials of <V_i, w>,
// i < nv, and of <w, w> at column nv: hpart[blk * (nv + 1) + i]
__global__ __launch_bounds__(MT) void k_center_dots(int64_t Nt, double* __restrict__ w, const double* __restrict__ part_mean,
                                                    int nv, PtrTable V, double* __restrict__ hpart) {
    __shared__ double sh[8];
    double mean = 0.0;
    if (part_mean) mean = m_sum_partials(part_mean, 1, 0, sh) / (double)Nt;
    double ww = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) {
        const double v = w[i] - mean;
        if (part_mean) w[i] = v;
        ww += v * v;
    }
    const double tw = m_block_sum(ww, sh);
    if (threadIdx.x == 0) hpart[(size_t)blockIdx.x * (nv + 1) + nv] = tw;
    __syncthreads();          // (this block's writes of w are visible to its own re-reads below)
    for (int k = 0; k < nv; ++k) {
        const double* vk = V.p[k];
        double s = 0.0;
        for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) s += vk[i] * w[i];
        const double t = m_block_sum(s, sh);
        if (threadIdx.x == 0) hpart[(size_t)blockIdx.x * (nv + 1) + k] = t;
    }
}
// w <- w - sum_k c_k V_k ; optionally vnext = w * scale_next
__global__ __launch_bounds__(MT) void k_axpy_multi(int64_t Nt, double* __restrict__ w, int nv, PtrTable V, double* __restrict__ vnext,
                                                   double scale_next) {
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) {
        double v = w[i];
        for (int k = 0; k < nv; ++k) v -= V.c[k] * V.p[k][i];
        w[i] = v;
        if (vnext) vnext[i] = v * scale_next;
    }
}
// out = a * x + b   (b: scalar), block partials of sum(out^2) (optional)
__global__ __launch_bounds__(MT) void k_scale_shift(int64_t Nt, const double* __restrict__ x, double a, double bsh,
                                                    double* __restrict__ out, double* __restrict__ part) {
    __shared__ double sh[8];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) {
        const double v = a * x[i] + bsh;
        out[i] = v;
        s += v * v;
    }
    if (part) {
        const double t = m_block_sum(s, sh);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
}
// block partials of {sum x, sum x^2, max |x|}
__global__ __launch_bounds__(MT) void k_sum_partials(int64_t Nt, const double* __restrict__ x, double* __restrict__ part) {
    __shared__ double sh[8];
    __shared__ double shm[MT];
    double s = 0.0, s2 = 0.0, mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) {
        const double v = x[i];
        s += v;
        s2 += v * v;
        mx = fmax(mx, fabs(v));
    }
    shm[threadIdx.x] = mx;
    const double t = m_block_sum(s, sh), t2 = m_block_sum(s2, sh);
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int i = 0; i < MT; ++i) m = fmax(m, shm[i]);
        part[3 * blockIdx.x] = t;
        part[3 * blockIdx.x + 1] = t2;
        part[3 * blockIdx.x + 2] = m;
    }
}
// r = b - a ; block partials of <r, r>
__global__ __launch_bounds__(MT) void k_residual_norm(int64_t Nt, const double* __restrict__ bvec, const double* __restrict__ a,
                                                      double* __restrict__ r, double* __restrict__ part) {
    __shared__ double sh[8];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) {
        const double v = bvec[i] - a[i];
        r[i] = v;
        s += v * v;
    }
    const double t = m_block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// Anderson: block partials of <R_k, pf> (k < nh) and <pf, pf> at column nh
__global__ __launch_bounds__(MT) void k_anderson_dots(int64_t Nt, const double* __restrict__ pf, int nh, PtrTable R,
                                                      double* __restrict__ hpart) {
    __shared__ double sh[8];
    for (int k = 0; k <= nh; ++k) {
        const double* rk = k < nh ? R.p[k] : pf;
        double s = 0.0;
        for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) s += rk[i] * pf[i];
        const double t = m_block_sum(s, sh);
        if (threadIdx.x == 0) hpart[(size_t)blockIdx.x * (nh + 1) + k] = t;
    }
}
// step scalars: block partials of <a, b> (column 0) and ||a - c||^2 (column 1); a null b / c leaves its column at zero
__global__ __launch_bounds__(MT) void k_step_sums(int64_t Nt, const double* __restrict__ a, const double* __restrict__ b,
                                                  const double* __restrict__ c, double* __restrict__ hpart) {
    __shared__ double sh[8];
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) {
        const double ai = a[i];
        if (b) s0 += ai * b[i];
        if (c) {
            const double d = ai - c[i];
            s1 += d * d;
        }
    }
    const double t0 = m_block_sum(s0, sh);
    const double t1 = m_block_sum(s1, sh);
    if (threadIdx.x == 0) {
        hpart[(size_t)blockIdx.x * 2] = t0;
        hpart[(size_t)blockIdx.x * 2 + 1] = t1;
    }
}
// Anderson: xn = c0 (x + alpha pf) + sum_k beta_k (X_k + alpha R_k); the pair (x, pf) goes into the history slot
__global__ __launch_bounds__(MT) void k_anderson_update(int64_t Nt, const double* __restrict__ x, const double* __restrict__ pf,
                                                        double alpha, double c0, int nh, PtrTable X, PtrTable R,
                                                        double* __restrict__ xn, double* __restrict__ slot_x,
                                                        double* __restrict__ slot_r) {
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < Nt; i += (int64_t)gridDim.x * MT) {
        const double xi = x[i], pi = pf[i];
        double v = c0 * (xi + alpha * pi);
        for (int k = 0; k < nh; ++k) v += X.c[k] * (X.p[k][i] + alpha * R.p[k][i]);
        xn[i] = v;
        if (slot_x) {
            slot_x[i] = xi;
            slot_r[i] = pi;
        }
    }
}

// ---- tiny dense helpers on the host (m <= 32) ----
// symmetric Jacobi eigen-decomposition: A (n x n, row-major, destroyed) -> eigenvalues w, eigenvectors Q (columns)
void sym_eig(int n, std::vector<double>& A, std::vector<double>& w, std::vector<double>& Q) {
    Q.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) Q[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) (i == j ? dg : off) += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        if (off <= 1e-32 * (dg + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double qkp = Q[(size_t)k * n + p], qkq = Q[(size_t)k * n + q];
                    Q[(size_t)k * n + p] = c * qkp - s * qkq;
                    Q[(size_t)k * n + q] = s * qkp + c * qkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

inline dim3 grid() { return dim3(MB); }

// sum the MB block partials of column q (stride `stride`) that a kernel left in the landing zone
inline double host_sum(const double* part, int stride, int q) {
    double t = 0.0;
    for (int i = 0; i < MB; ++i) t += part[(size_t)i * stride + q];
    return t;
}
}  // namespace dftk_mix
using namespace dftk_mix;

// The two cube-sized scalars an SCF step reads on the host besides the term energies -- int V_in rho_out (the Ritz-value form
// of the nonlocal energy, terms.py) and ||rho_out - rho_in||^2 (ScfConvergenceDensity, self_consistent_field.jl:229-236) --
// in ONE kernel and ONE synchronisation (they were two chains of torch launches with a fetch each).
extern "C" int dftk_mi_step_sums(dftk_mi_basis* b, int64_t n, const double* a_d, const double* b_d, const double* c_d,
                                 double* out_h) {
    if (!b || n < 1 || !a_d || (!b_d && !c_d) || !out_h) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    double* hpart = reinterpret_cast<double*>(b->h_fetch);
    hipLaunchKernelGGL(k_step_sums, grid(), dim3(MT), 0, b->stream, n, a_d, b_d, c_d, hpart);
    HIPCHK(hipGetLastError());
    CHK(host_wait(b));
    out_h[0] = host_sum(hpart, 2, 0);
    out_h[1] = host_sum(hpart, 2, 1);
    return 0;
}

// =============================================================================================== Anderson acceleration
struct dftk_mi_anderson {
    dftk_mi_basis* b;
    int64_t n;
    int m;
    double maxcond, errorfactor;
    double* buf;                       // (m + 1) slots of (x, r): 2 (m + 1) n doubles
    std::vector<int> slots;            // history order -> slot index
    std::vector<int> free_slots;
    std::vector<double> errors;        // ||r_i||
    std::vector<double> gram;          // gram[i * nh + j] = <r_i, r_j>, history order
};

extern "C" int dftk_mi_anderson_create(dftk_mi_basis* b, int64_t n, int m, double maxcond, double errorfactor,
                                       dftk_mi_anderson** out) {
    if (!b || n < 1 || m < 0 || m >= MAXK || !out) return DFTK_MI_EINVAL;
    HIPCHK(hipSetDevice(b->device));
    dftk_mi_anderson* a = new dftk_mi_anderson();
    a->b = b;
    a->n = n;
    a->m = m;
    a->maxcond = maxcond;
    a->errorfactor = errorfactor;
    a->buf = nullptr;
    if (m > 0) {
        if (dftk_scratch_malloc((void**)&a->buf, 2 * (size_t)(m + 1) * n * sizeof(double)) != hipSuccess) {
            delete a;
            dftk_set_error("anderson_create: cannot allocate %d history entries of %lld doubles", m + 1, (long long)n);
            return DFTK_MI_EHIP;
        }
        for (int i = m; i >= 0; --i) a->free_slots.push_back(i);
    }
    *out = a;
    return 0;
}
extern "C" int dftk_mi_anderson_destroy(dftk_mi_anderson* a) {
    if (!a) return 0;
    hipSetDevice(a->b->device);
    hipStreamSynchronize(a->b->stream);
    if (a->buf) hipFree(a->buf);
    delete a;
    return 0;
}
extern "C" int dftk_mi_anderson_reset(dftk_mi_anderson* a) {
    if (!a) return DFTK_MI_EINVAL;
    for (int s : a->slots) a->free_slots.push_back(s);
    a->slots.clear();
    a->errors.clear();
    a->gram.clear();
    return 0;
}
extern "C" int dftk_mi_anderson_history(const dftk_mi_anderson* a) { return a ? (int)a->slots.size() : -1; }

static void anderson_delete(dftk_mi_anderson* a, const std::vector<int>& idxs) {
    const int nh = (int)a->slots.size();
    std::vector<char> gone(nh, 0);
    for (int i : idxs) gone[i] = 1;
    std::vector<int> keep;
    for (int i = 0; i < nh; ++i)
        if (!gone[i])
            keep.push_back(i);
        else
            a->free_slots.push_back(a->slots[i]);
    std::vector<int> slots;
    std::vector<double> errors, gram(keep.size() * keep.size());
    for (size_t i = 0; i < keep.size(); ++i) {
        slots.push_back(a->slots[keep[i]]);
        errors.push_back(a->errors[keep[i]]);
        for (size_t j = 0; j < keep.size(); ++j) gram[i * keep.size() + j] = a->gram[(size_t)keep[i] * nh + keep[j]];
    }
    a->slots.swap(slots);
    a->errors.swap(errors);
    a->gram.swap(gram);
}

// x_next = Anderson(x, alpha, Pf)  (anderson.jl:81-130; ScfAndersonDensitySolver, scf_solvers.jl:85-98)
extern "C" int dftk_mi_anderson_step(dftk_mi_anderson* a, const double* x_d, double alpha, const double* pf_d, double* xn_d,
                                     int* n_history) {
    if (!a || !x_d || !pf_d || !xn_d || xn_d == x_d || xn_d == pf_d) return DFTK_MI_EINVAL;
    dftk_mi_basis* b = a->b;
    HIPCHK(hipSetDevice(b->device));
    const int64_t n = a->n;
    PtrTable X{}, R{};
    if (a->m == 0) {
        hipLaunchKernelGGL(k_anderson_update, grid(), dim3(MT), 0, b->stream, n, x_d, pf_d, alpha, 1.0, 0, X, R, xn_d,
                           (double*)nullptr, (double*)nullptr);
        HIPCHK(hipGetLastError());
        return 0;
    }
    auto slot_x = [&](int s) { return a->buf + 2 * (size_t)s * n; };
    auto slot_r = [&](int s) { return a->buf + (2 * (size_t)s + 1) * n; };
    int nh = (int)a->slots.size();
    // <r_i, pf> for the whole history and <pf, pf>: one kernel, partials into the pinned landing zone, one synchronisation
    for (int k = 0; k < nh; ++k) R.p[k] = slot_r(a->slots[k]);
    double* hpart = reinterpret_cast<double*>(b->h_fetch);
    hipLaunchKernelGGL(k_anderson_dots, grid(), dim3(MT), 0, b->stream, n, pf_d, nh, R, hpart);
    HIPCHK(hipGetLastError());
    CHK(host_wait(b));
    std::vector<double> rp(nh);
    for (int k = 0; k < nh; ++k) rp[k] = host_sum(hpart, nh + 1, k);
    const double pp = host_sum(hpart, nh + 1, nh);
    if (!std::isfinite(pp)) {
        dftk_set_error("anderson_step: non-finite preconditioned residual");
        return DFTK_MI_NUM_NONFINITE;
    }
    auto push = [&](const std::vector<double>& row) {
        // (x, pf) enter the history: written into a free slot by the update kernel below
        const int s = a->free_slots.back();
        a->free_slots.pop_back();
        const int n0 = (int)a->slots.size();
        std::vector<double> g((size_t)(n0 + 1) * (n0 + 1), 0.0);
        for (int i = 0; i < n0; ++i)
            for (int j = 0; j < n0; ++j) g[(size_t)i * (n0 + 1) + j] = a->gram[(size_t)i * n0 + j];
        for (int i = 0; i < n0; ++i) g[(size_t)i * (n0 + 1) + n0] = g[(size_t)n0 * (n0 + 1) + i] = row[i];
        g[(size_t)n0 * (n0 + 1) + n0] = pp;
        a->gram.swap(g);
        a->slots.push_back(s);
        a->errors.push_back(std::sqrt(std::max(pp, 0.0)));
        return s;
    };
    if (nh == 0) {
        const int s = push(rp);
        hipLaunchKernelGGL(k_anderson_update, grid(), dim3(MT), 0, b->stream, n, x_d, pf_d, alpha, 1.0, 0, X, R, xn_d, slot_x(s),
                           slot_r(s));
        HIPCHK(hipGetLastError());
        if (n_history) *n_history = (int)a->slots.size();
        return 0;
    }
    // drop entries whose error is more than errorfactor times the smallest one (the newest entry is never dropped)
    const double err_n = std::sqrt(std::max(pp, 0.0));
    double min_error = err_n;
    for (double e : a->errors) min_error = std::min(min_error, e);
    std::vector<int> drop;
    for (int i = 0; i + 1 < nh; ++i)
        if (a->errors[i] > a->errorfactor * min_error) drop.push_back(i);
    if (!drop.empty()) {
        std::vector<double> rp2;
        for (int i = 0; i < nh; ++i)
            if (std::find(drop.begin(), drop.end(), i) == drop.end()) rp2.push_back(rp[i]);
        anderson_delete(a, drop);
        rp.swap(rp2);
        nh = (int)a->slots.size();
    }
    // M[:, j] = r_j - pf  =>  G = M'M, bvec = M'pf; the oldest-by-error entries go while cond(R) = sqrt(cond(G)) > maxcond
    std::vector<int> keep(nh);
    for (int i = 0; i < nh; ++i) keep[i] = i;
    std::vector<double> G, bv, w, Q;
    for (;;) {
        const int k = (int)keep.size();
        G.assign((size_t)k * k, 0.0);
        bv.assign(k, 0.0);
        for (int i = 0; i < k; ++i) {
            bv[i] = rp[keep[i]] - pp;
            for (int j = 0; j < k; ++j) G[(size_t)i * k + j] = a->gram[(size_t)keep[i] * nh + keep[j]] - rp[keep[i]] - rp[keep[j]] + pp;
        }
        std::vector<double> Gc = G;
        sym_eig(k, Gc, w, Q);
        double wmax = 0.0, wmin = 1e300;
        for (double v : w) {
            wmax = std::max(wmax, v);
            wmin = std::min(wmin, v);
        }
        const double condR = wmax > 0 ? std::sqrt(std::max(wmax, 0.0) / std::max(wmin, 1e-300)) : 1.0;
        if (k > 1 && condR > a->maxcond) {
            int worst = 0;
            for (int i = 1; i + 1 < k; ++i)
                if (a->errors[keep[i]] > a->errors[keep[worst]]) worst = i;
            keep.erase(keep.begin() + worst);
            continue;
        }
        break;
    }
    if ((int)keep.size() < nh) {
        std::vector<int> gone;
        std::vector<double> rp2;
        for (int i = 0; i < nh; ++i)
            if (std::find(keep.begin(), keep.end(), i) == keep.end())
                gone.push_back(i);
            else
                rp2.push_back(rp[i]);
        anderson_delete(a, gone);
        rp.swap(rp2);
        nh = (int)a->slots.size();
    }
    // betas = -lstsq(G, bvec): minimum-norm solution through the eigen-decomposition (singular values below
    // eps * k * s_max are treated as zero, numpy.linalg.lstsq's default cut-off)
    const int k = nh;
    std::vector<double> betas(k, 0.0);
    {
        double smax = 0.0;
        for (double v : w) smax = std::max(smax, std::fabs(v));
        const double cut = 2.220446049250313e-16 * k * smax;
        for (int e = 0; e < k; ++e) {
            if (!(std::fabs(w[e]) > cut)) continue;
            double qb = 0.0;
            for (int i = 0; i < k; ++i) qb += Q[(size_t)i * k + e] * bv[i];
            for (int i = 0; i < k; ++i) betas[i] -= Q[(size_t)i * k + e] * qb / w[e];
        }
    }
    double sb = 0.0;
    for (int i = 0; i < k; ++i) {
        sb += betas[i];
        X.p[i] = slot_x(a->slots[i]);
        R.p[i] = slot_r(a->slots[i]);
        X.c[i] = betas[i];
    }
    // a full history makes room first (the oldest entry leaves AFTER it has been used, as the reference's push! does)
    int s_new;
    if ((int)a->slots.size() >= a->m + 1 || a->free_slots.empty()) {
        dftk_set_error("anderson_step: history bookkeeping error");
        return DFTK_MI_EINVAL;
    }
    s_new = push(rp);
    hipLaunchKernelGGL(k_anderson_update, grid(), dim3(MT), 0, b->stream, n, x_d, pf_d, alpha, 1.0 - sb, k, X, R, xn_d,
                       slot_x(s_new), slot_r(s_new));
    HIPCHK(hipGetLastError());
    if ((int)a->slots.size() > a->m) anderson_delete(a, std::vector<int>{0});
    if (n_history) *n_history = (int)a->slots.size();
    return 0;
}

// ========================================================================================================= chi0 mixing
namespace {
struct MixWork {
    dftk_mi_kblock* kb;
    dftk_mi_basis* b;
    int64_t N, Nt;
    int ncomp;
    const double* poisson;
    const double* ldos;
    double dvol, tdos;
    bool dielectric;
    double kTF, eps_r;
    Lat9 L;
    cd *c1, *c2, *c3;
    double* dV;
    double* part;        // device partials (MB x 3 doubles)
    int n_applies = 0;
};

// out = eps' x = x - chi0 vc x, mean not yet removed; block partials of sum(out) are left in w.part + 2 MB
int apply_dielectric_adjoint(MixWork& w, const double* x, double* out) {
    dftk_mi_basis* b = w.b;
    const dim3 g = grid(), t(MT);
    w.n_applies += 1;
    hipLaunchKernelGGL(k_total_r2c, g, t, 0, b->stream, w.N, w.ncomp, x, w.c1);
    HIPCHK(hipGetLastError());
    if (w.poisson) {
        CHK(launch_fft_from_cube(w.kb, w.c1, w.c2));
        hipLaunchKernelGGL(k_mult_array, g, t, 0, b->stream, w.N, w.poisson, w.c2);
        HIPCHK(hipGetLastError());
        CHK(launch_ifft_to_cube(w.kb, w.c2, w.c1));
        hipLaunchKernelGGL(k_dv_partials, g, t, 0, b->stream, w.N, w.ncomp, (const cd*)w.c1, 1.0 / (double)w.N, w.ldos, w.dV, w.part);
    } else {
        HIPCHK(hipMemsetAsync(w.c1, 0, (size_t)w.N * sizeof(cd), b->stream));
        hipLaunchKernelGGL(k_dv_partials, g, t, 0, b->stream, w.N, w.ncomp, (const cd*)w.c1, 0.0, w.ldos, w.dV, w.part);
    }
    HIPCHK(hipGetLastError());
    const cd* diel = nullptr;
    if (w.dielectric && w.poisson) {
        // the dielectric model acts on dV through its Fourier multiplier (zero at G = 0: the mean of dV does not matter)
        hipLaunchKernelGGL(k_mult_chi0_dielectric, g, t, 0, b->stream, b->nx, b->ny, b->nz, w.L, w.kTF, w.eps_r, (const cd*)w.c2, w.c3);
        HIPCHK(hipGetLastError());
        CHK(launch_ifft_to_cube(w.kb, w.c3, w.c1));
        diel = w.c1;
    }
    hipLaunchKernelGGL(k_apply_chi0, g, t, 0, b->stream, w.N, w.ncomp, x, (const double*)w.dV, w.ldos, (const double*)w.part, w.dvol,
                       w.tdos, diel, 1.0 / (double)w.N, out, w.part + 2 * MB);
    HIPCHK(hipGetLastError());
    return 0;
}
}  // namespace

// d_rho = (1 - chi0 vc)^-1 dF by restarted GMRES (mixing.jl:241-290).  One host synchronisation per Krylov step.
extern "C" int dftk_mi_chi0_mix(dftk_mi_kblock* cube_kb, int n_comp, const double* recip_lattice_h, const double* poisson_d,
                                const double* ldos_d, double dvol, int dielectric, double kTF, double eps_r,
                                const double* dF_d, double reltol, int krylovdim, int maxiter, double* drho_d, int* n_applies,
                                int* converged) {
    if (!cube_kb || cube_kb->sh_comm || (n_comp != 1 && n_comp != 2) || !dF_d || !drho_d || !(reltol > 0) || krylovdim < 1 ||
        krylovdim >= MAXK || maxiter < 1 || (dielectric && !recip_lattice_h))
        return DFTK_MI_EINVAL;
    dftk_mi_basis* b = cube_kb->basis;
    HIPCHK(hipSetDevice(b->device));
    const int64_t N = (int64_t)b->nx * b->ny * b->nz;
    if (cube_kb->n_G != N) {
        dftk_set_error("chi0_mix: the k-block must span the whole cube");
        return DFTK_MI_EINVAL;
    }
    const int64_t Nt = N * n_comp;
    if (n_applies) *n_applies = 0;
    if (converged) *converged = 1;
    const dim3 g = grid(), t(MT);
    double* hpart = reinterpret_cast<double*>(b->h_fetch);
    // workspace: three complex cubes, dV, partials, then the vectors of the solver: b, x, r / w, and the Krylov basis
    // (krylovdim + 1 vectors; the buffer grows geometrically -- a solve that ends in 2-3 steps must not reserve 31 cubes)
    const size_t fixed = 3 * (size_t)N * sizeof(cd) + (size_t)N * sizeof(double) + 4 * MB * sizeof(double) + 256;
    int vcap = std::min(krylovdim + 1, 8);
    auto total_bytes = [&](int cap) { return fixed + (size_t)(3 + cap) * Nt * sizeof(double); };
    CHK(cube_ws_ensure(b, total_bytes(vcap)));
    MixWork w;
    auto bind = [&]() {
        char* p = reinterpret_cast<char*>(b->dense_ws);
        w.c1 = reinterpret_cast<cd*>(p);
        w.c2 = w.c1 + N;
        w.c3 = w.c2 + N;
        w.dV = reinterpret_cast<double*>(w.c3 + N);
        w.part = w.dV + N;
    };
    bind();
    w.kb = cube_kb;
    w.b = b;
    w.N = N;
    w.Nt = Nt;
    w.ncomp = n_comp;
    w.poisson = poisson_d;
    w.ldos = ldos_d;
    w.dvol = dvol;
    w.tdos = 0.0;
    w.dielectric = dielectric != 0 && (1.0 - eps_r) != 0.0;
    w.kTF = kTF;
    w.eps_r = eps_r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) w.L.B[3 * i + j] = recip_lattice_h ? recip_lattice_h[i + 3 * j] : 0.0;
    auto vec = [&](int i) { return w.part + 4 * MB + 32 + (size_t)i * Nt; };   // 0: b, 1: x, 2: w / r, 3 ...: Krylov basis
    // ---- first fetch: sum / max of the LDOS (is the model alive?) and the mean and norm of dF ----
    hipLaunchKernelGGL(k_sum_partials, g, t, 0, b->stream, Nt, dF_d, hpart);
    if (ldos_d) hipLaunchKernelGGL(k_sum_partials, g, t, 0, b->stream, Nt, ldos_d, hpart + 3 * MB);
    HIPCHK(hipGetLastError());
    CHK(host_wait(b));
    const double sumF = host_sum(hpart, 3, 0);
    if (ldos_d) {
        double amax = 0.0;
        for (int i = 0; i < MB; ++i) amax = std::max(amax, hpart[3 * MB + 3 * i + 2]);
        const double total = host_sum(hpart + 3 * MB, 3, 0);
        if (!(amax >= std::sqrt(2.220446049250313e-16))) {
            w.ldos = nullptr;                                   // chi0models.jl:32: no LDOS -> the term does not exist
        } else {
            w.tdos = total * dvol;
        }
    }
    if (!std::isfinite(sumF)) {
        dftk_set_error("chi0_mix: non-finite input");
        return DFTK_MI_NUM_NONFINITE;
    }
    if (!w.ldos && !w.dielectric) {                             // "do not bother running GMRES": simple mixing
        if (drho_d != dF_d) HIPCHK(hipMemcpyAsync(drho_d, dF_d, Nt * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        return 0;
    }
    const double dc = sumF / (double)Nt;
    // b = dF - mean(dF), ||b||
    hipLaunchKernelGGL(k_scale_shift, g, t, 0, b->stream, Nt, dF_d, 1.0, -dc, vec(0), hpart);
    HIPCHK(hipMemsetAsync(vec(1), 0, Nt * sizeof(double), b->stream));
    HIPCHK(hipGetLastError());
    CHK(host_wait(b));
    const double nb = std::sqrt(std::max(host_sum(hpart, 1, 0), 0.0));
    const double tol = std::max(1e-12, reltol * nb);
    double beta = nb;
    const double* rsrc = vec(0);      // the current residual: b itself while x = 0
    bool ok = beta <= tol;
    std::vector<double> H((size_t)(krylovdim + 1) * krylovdim), gv(krylovdim + 1), cs(krylovdim), sn(krylovdim), y(krylovdim);
    auto Hat = [&](int i, int j) -> double& { return H[(size_t)i * krylovdim + j]; };
    for (int cycle = 0; cycle < maxiter && !ok; ++cycle) {
        // V_0 = r / beta
        hipLaunchKernelGGL(k_scale_shift, g, t, 0, b->stream, Nt, rsrc, 1.0 / beta, 0.0, vec(3), (double*)nullptr);
        HIPCHK(hipGetLastError());
        std::fill(H.begin(), H.end(), 0.0);
        std::fill(gv.begin(), gv.end(), 0.0);
        gv[0] = beta;
        int k_used = 0;
        for (int k = 0; k < krylovdim; ++k) {
            if (k + 2 > vcap) {
                // grow the Krylov basis: new buffer, everything allocated so far moves with one device copy
                const int ncap = std::min(krylovdim + 1, 2 * vcap);
                const size_t old_bytes = total_bytes(vcap);
                void* old = b->dense_ws;
                CHK(host_wait(b));
                void* nw = nullptr;
                HIPCHK(dftk_scratch_malloc(&nw, total_bytes(ncap)));
                HIPCHK(hipMemcpyAsync(nw, old, old_bytes, hipMemcpyDeviceToDevice, b->stream));
                CHK(host_wait(b));
                HIPCHK(hipFree(old));
                b->dense_ws = nw;
                b->dense_ws_bytes = total_bytes(ncap);
                vcap = ncap;
                bind();
            }
            double* wv = vec(2);
            CHK(apply_dielectric_adjoint(w, vec(3 + k), wv));
            // w -= mean(w); h = V' w and <w, w>: one kernel, one fetch (classical Gram-Schmidt; the norm of the remainder
            // follows from Pythagoras and is recomputed only when cancellation has eaten its digits)
            PtrTable V{};
            for (int i = 0; i <= k; ++i) V.p[i] = vec(3 + i);
            hipLaunchKernelGGL(k_center_dots, g, t, 0, b->stream, Nt, wv, (const double*)(w.part + 2 * MB), k + 1, V, hpart);
            HIPCHK(hipGetLastError());
            CHK(host_wait(b));
            std::vector<double> h(k + 1);
            for (int i = 0; i <= k; ++i) h[i] = host_sum(hpart, k + 2, i);
            double ww = host_sum(hpart, k + 2, k + 1);
            if (!std::isfinite(ww)) {
                dftk_set_error("chi0_mix: non-finite Krylov vector");
                return DFTK_MI_NUM_NONFINITE;
            }
            double hh = 0.0;
            for (double v : h) hh += v * v;
            double rest = ww - hh;
            bool updated = false;
            if (rest < 1e-2 * ww) {
                // cancellation: a second pass of Gram-Schmidt ("twice is enough") and the remainder's own norm
                for (int i = 0; i <= k; ++i) V.c[i] = h[i];
                hipLaunchKernelGGL(k_axpy_multi, g, t, 0, b->stream, Nt, wv, k + 1, V, (double*)nullptr, 0.0);
                hipLaunchKernelGGL(k_center_dots, g, t, 0, b->stream, Nt, wv, (const double*)nullptr, k + 1, V, hpart);
                HIPCHK(hipGetLastError());
                CHK(host_wait(b));
                double h2h2 = 0.0;
                std::vector<double> h2(k + 1);
                for (int i = 0; i <= k; ++i) {
                    h2[i] = host_sum(hpart, k + 2, i);
                    h2h2 += h2[i] * h2[i];
                    h[i] += h2[i];
                }
                const double ww2 = host_sum(hpart, k + 2, k + 1);
                for (int i = 0; i <= k; ++i) V.c[i] = h2[i];
                rest = ww2 - h2h2;
                if (rest < 1e-2 * ww2) {
                    hipLaunchKernelGGL(k_axpy_multi, g, t, 0, b->stream, Nt, wv, k + 1, V, (double*)nullptr, 0.0);
                    hipLaunchKernelGGL(k_center_dots, g, t, 0, b->stream, Nt, wv, (const double*)nullptr, 0, V, hpart);
                    HIPCHK(hipGetLastError());
                    CHK(host_wait(b));
                    rest = host_sum(hpart, 1, 0);
                    updated = true;
                }
            } else {
                for (int i = 0; i <= k; ++i) V.c[i] = h[i];
            }
            for (int i = 0; i <= k; ++i) Hat(i, k) = h[i];
            Hat(k + 1, k) = std::sqrt(std::max(rest, 0.0));
            for (int j = 0; j < k; ++j) {
                const double tt = cs[j] * Hat(j, k) + sn[j] * Hat(j + 1, k);
                Hat(j + 1, k) = -sn[j] * Hat(j, k) + cs[j] * Hat(j + 1, k);
                Hat(j, k) = tt;
            }
            const double d = std::hypot(Hat(k, k), Hat(k + 1, k));
            if (d == 0.0) {
                cs[k] = 1.0;
                sn[k] = 0.0;
            } else {
                cs[k] = Hat(k, k) / d;
                sn[k] = Hat(k + 1, k) / d;
            }
            Hat(k, k) = d;
            const double hk1 = Hat(k + 1, k);
            Hat(k + 1, k) = 0.0;
            gv[k + 1] = -sn[k] * gv[k];
            gv[k] = cs[k] * gv[k];
            k_used = k + 1;
            const bool last = std::fabs(gv[k + 1]) <= tol || hk1 == 0.0 || k + 1 == krylovdim;
            if (!last) {
                // V_{k+1} = (w - V h) / hk1 in the same pass as the subtraction
                if (updated)
                    hipLaunchKernelGGL(k_scale_shift, g, t, 0, b->stream, Nt, (const double*)wv, 1.0 / hk1, 0.0, vec(3 + k + 1),
                                       (double*)nullptr);
                else
                    hipLaunchKernelGGL(k_axpy_multi, g, t, 0, b->stream, Nt, wv, k + 1, V, vec(3 + k + 1), 1.0 / hk1);
                HIPCHK(hipGetLastError());
            }
            if (std::fabs(gv[k + 1]) <= tol || hk1 == 0.0) break;
        }
        // y = triu(H) \ g ; x += V y
        for (int i = k_used - 1; i >= 0; --i) {
            double s = gv[i];
            for (int j = i + 1; j < k_used; ++j) s -= Hat(i, j) * y[j];
            y[i] = s / Hat(i, i);
        }
        {
            PtrTable V{};
            for (int i = 0; i < k_used; ++i) {
                V.p[i] = vec(3 + i);
                V.c[i] = -y[i];
            }
            hipLaunchKernelGGL(k_axpy_multi, g, t, 0, b->stream, Nt, vec(1), k_used, V, (double*)nullptr, 0.0);
            HIPCHK(hipGetLastError());
        }
        // |g[k_used]| is the residual norm of the minimiser while the basis is orthonormal (the second pass maintains that);
        // the TRUE residual is measured at every restart and before an accepted exit of a long cycle
        if (std::fabs(gv[k_used]) <= tol && k_used <= 8) {
            beta = std::fabs(gv[k_used]);
            ok = true;
            break;
        }
        CHK(apply_dielectric_adjoint(w, vec(1), vec(2)));
        {
            PtrTable V{};
            hipLaunchKernelGGL(k_center_dots, g, t, 0, b->stream, Nt, vec(2), (const double*)(w.part + 2 * MB), 0, V, hpart);
            hipLaunchKernelGGL(k_residual_norm, g, t, 0, b->stream, Nt, (const double*)vec(0), (const double*)vec(2), vec(2), hpart);
            HIPCHK(hipGetLastError());
            CHK(host_wait(b));
            beta = std::sqrt(std::max(host_sum(hpart, 1, 0), 0.0));
        }
        rsrc = vec(2);
        ok = beta <= tol;
    }
    // d_rho = x + mean(dF)
    hipLaunchKernelGGL(k_scale_shift, g, t, 0, b->stream, Nt, (const double*)vec(1), 1.0, dc, drho_d, (double*)nullptr);
    HIPCHK(hipGetLastError());
    if (n_applies) *n_applies = w.n_applies;
    if (converged) *converged = ok ? 1 : 0;
    return host_wait(b);          // (the workspace is shared with the other cube operations of this basis)
}
