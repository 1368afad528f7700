// gamma_kernels.hip -- real-symmetric orbitals of a Gamma-point block (gfx950 only).
//
// At k = 0 the Kohn-Sham Hamiltonian of the reference's models (real local potential, src/terms/local.jl; HGH
// projectors = Fourier transforms of REAL functions, src/terms/nonlocal.jl:176-220) commutes with complex
// conjugation in real space, so its eigenvectors can be chosen as REAL fields: psi(-G) = conj(psi(G)).  The reference
// keeps general complex orbitals there (src/ has no Gamma special case); this file is an EXTENSION that restricts
// LOBPCG to that invariant subspace -- eigenvalues, density and energies are unchanged (the restricted operator is
// a real symmetric matrix with the same spectrum and multiplicities), the orbitals differ from the reference's by
// the unitary mixing inside degenerate subspaces / global phases that are arbitrary there as well.
//
// HALF-SPHERE FORMAT of a real-symmetric vector x: row 0 = x(G = 0) (real; imaginary part stored as exactly 0),
// row j > 0 = sqrt(2) x(G_j) for one representative G_j of every pair {G, -G}.  With this scaling the plain
// real dot product of the 2 n_half real numbers equals the full inner product <x, y>, so every n_G-long product
// of LOBPCG becomes a REAL GEMM (zgemm flag DFTK_MI_GEMM_REAL: two instead of three real matrix-core products per
// stored complex entry) over HALF as many rows: a third of the flops of the 3M complex product; the dense
// 3M x 3M algebra sees real symmetric matrices (stored as complex with zero imaginary parts).
// H psi: two half-format bands a, b travel through ONE complex FFT pipeline as the full-sphere vector a + i b
// (the local potential is real and the kinetic factor even in G, so H_loc (a + i b) = H_loc a + i H_loc b with both
// parts real-symmetric) and are separated again by W(G) +/- conj(W(-G)).
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <unordered_map>
#include <vector>

// pair tables: g[j] / mg[j] = sphere row of G_j / -G_j; j = 0 is G = 0.  Pairs in ascending order of the first row.
int gamma_tables_host(int nx, int ny, int nz, int64_t n_G, const int64_t* mapping, int64_t* n_half_out, int32_t* g,
                      int32_t* mg) {
    if (n_G < 1 || mapping[0] != 0) {
        dftk_set_error("gamma_real: the sphere must contain G = 0 as its first entry");
        return DFTK_MI_EINVAL;
    }
    std::unordered_map<int64_t, int32_t> row;
    row.reserve((size_t)n_G * 2);
    for (int64_t c = 0; c < n_G; ++c) row[mapping[c]] = (int32_t)c;
    int64_t nh = 0;
    for (int64_t c = 0; c < n_G; ++c) {
        const int64_t lin = mapping[c];
        const int ix = (int)(lin % nx), iy = (int)((lin / nx) % ny), iz = (int)(lin / ((int64_t)nx * ny));
        const int64_t mlin = (int64_t)((nx - ix) % nx) + (int64_t)nx * (((ny - iy) % ny) + (int64_t)ny * ((nz - iz) % nz));
        auto it = row.find(mlin);
        if (it == row.end()) {
            dftk_set_error("gamma_real: the sphere is not inversion symmetric (row %lld has no -G partner)", (long long)c);
            return DFTK_MI_EINVAL;
        }
        const int32_t pc = it->second;
        if (pc < c) continue;            // pair already listed from its first member
        if (pc == c && c != 0) {          // a Nyquist point is its own partner: not a valid orbital sphere
            dftk_set_error("gamma_real: row %lld is its own inversion partner (Nyquist frequency inside the sphere)", (long long)c);
            return DFTK_MI_EINVAL;
        }
        if (g) {
            g[nh] = (int32_t)c;
            mg[nh] = pc;
        }
        nh += 1;
    }
    if (2 * nh - 1 != n_G) {
        dftk_set_error("gamma_real: inconsistent pairing (%lld pairs for %lld rows)", (long long)nh, (long long)n_G);
        return DFTK_MI_EINVAL;
    }
    *n_half_out = nh;
    return 0;
}

#define GR_SQRT2 1.4142135623730951
#define GR_ISQRT2 0.70710678118654752

// The format conversions are pure HBM streams (one 16-byte element per pair member): every thread handles GR_UNR pairs and
// issues all of its loads before the first use -- one element per thread kept 8-32 KB per CU in flight and reached
// 1.3-2.2 TB/s (tools/ew_bench.py).  A workgroup of 256 threads takes GR_UNR * 256 consecutive pairs of one column.
#define GR_UNR 4
#define GR_ROWS (256 * GR_UNR)
__global__ __launch_bounds__(256) void k_gr_compress(int64_t nh, const int* __restrict__ g, const int* __restrict__ mg,
                                                     const cd* __restrict__ X, int64_t ldx, cd* __restrict__ H, int64_t ldh) {
    const int64_t j0 = (int64_t)blockIdx.x * GR_ROWS + threadIdx.x;
    const cd* x = X + (int64_t)blockIdx.y * ldx;
    cd* h = H + (int64_t)blockIdx.y * ldh;
    int ig[GR_UNR], im[GR_UNR];
    cd a[GR_UNR], b[GR_UNR];
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t j = j0 + u * 256;
        ig[u] = j < nh ? g[j] : 0;
        im[u] = j < nh ? mg[j] : 0;
    }
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        a[u] = x[ig[u]];
        b[u] = x[im[u]];
    }
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t j = j0 + u * 256;
        const double s = j ? GR_ISQRT2 : 0.5;            // sqrt(2) * 1/2 (symmetric part), row 0: 1/2 (a == b)
        if (j < nh) h[j] = make_double2(s * (a[u].x + b[u].x), s * (a[u].y - b[u].y));
    }
}

// compress with the phase alignment of the LOBPCG entry: every column is first rotated by exp(-i phi),
// exp(2 i phi) = s / |s|, s = sum_G x(G) x(-G) -- the global phase that maximises its real-symmetric part (a real field
// times any phase becomes +- itself: nothing is lost, e.g. when orbitals of a complex iteration are handed over).  A
// column that is already real-symmetric has s > 0, phi = 0 exactly, and is compressed bit for bit as by k_gr_compress.
// One workgroup of 1024 threads per column: reduction pass over the pairs, then the write pass.
#define GRA_NT 1024
__global__ __launch_bounds__(GRA_NT) void k_gr_compress_aligned(int64_t nh, const int* __restrict__ g,
                                                                const int* __restrict__ mg, const cd* __restrict__ X,
                                                                int64_t ldx, cd* __restrict__ H, int64_t ldh) {
    __shared__ double sh[2][GRA_NT / 64];
    __shared__ double s_cs[2];
    const cd* x = X + (int64_t)blockIdx.x * ldx;
    cd* h = H + (int64_t)blockIdx.x * ldh;
    double sr = 0.0, si = 0.0;
    for (int64_t j0 = threadIdx.x; j0 < nh; j0 += GRA_NT * GR_UNR) {
        int ig[GR_UNR], im[GR_UNR];
        cd a[GR_UNR], b[GR_UNR];
#pragma unroll
        for (int u = 0; u < GR_UNR; ++u) {
            const int64_t j = j0 + u * GRA_NT;
            ig[u] = j < nh ? g[j] : 0;
            im[u] = j < nh ? mg[j] : 0;
        }
#pragma unroll
        for (int u = 0; u < GR_UNR; ++u) {
            a[u] = x[ig[u]];
            b[u] = x[im[u]];
        }
#pragma unroll
        for (int u = 0; u < GR_UNR; ++u) {
            const int64_t j = j0 + u * GRA_NT;
            const double w = j < nh ? (j ? 2.0 : 1.0) : 0.0;
            sr += w * (a[u].x * b[u].x - a[u].y * b[u].y);
            si += w * (a[u].x * b[u].y + a[u].y * b[u].x);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_down(sr, off, 64);
        si += __shfl_down(si, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = sr;
        sh[1][threadIdx.x >> 6] = si;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tr = 0.0, ti = 0.0;
#pragma unroll
        for (int w = 0; w < GRA_NT / 64; ++w) {
            tr += sh[0][w];
            ti += sh[1][w];
        }
        double c = 1.0, sn = 0.0;
        if ((tr != 0.0 || ti != 0.0) && isfinite(tr) && isfinite(ti) && !(ti == 0.0 && tr > 0.0)) {
            const double phi = 0.5 * atan2(ti, tr);
            c = cos(phi);
            sn = sin(phi);
        }
        s_cs[0] = c;
        s_cs[1] = sn;
    }
    __syncthreads();
    const double c = s_cs[0], sn = s_cs[1];
    const bool rot = sn != 0.0 || c != 1.0;
    for (int64_t j0 = threadIdx.x; j0 < nh; j0 += GRA_NT * GR_UNR) {
        int ig[GR_UNR], im[GR_UNR];
        cd a[GR_UNR], b[GR_UNR];
#pragma unroll
        for (int u = 0; u < GR_UNR; ++u) {
            const int64_t j = j0 + u * GRA_NT;
            ig[u] = j < nh ? g[j] : 0;
            im[u] = j < nh ? mg[j] : 0;
        }
#pragma unroll
        for (int u = 0; u < GR_UNR; ++u) {
            a[u] = x[ig[u]];
            b[u] = x[im[u]];
        }
#pragma unroll
        for (int u = 0; u < GR_UNR; ++u) {
            const int64_t j = j0 + u * GRA_NT;
            cd aa = a[u], bb = b[u];
            if (rot) {      // x * exp(-i phi)
                aa = make_double2(aa.x * c + aa.y * sn, aa.y * c - aa.x * sn);
                bb = make_double2(bb.x * c + bb.y * sn, bb.y * c - bb.x * sn);
            }
            const double s = j ? GR_ISQRT2 : 0.5;
            if (j < nh) h[j] = make_double2(s * (aa.x + bb.x), s * (aa.y - bb.y));
        }
    }
}

__global__ __launch_bounds__(256) void k_gr_expand(int64_t nh, const int* __restrict__ g, const int* __restrict__ mg,
                                                   const cd* __restrict__ H, int64_t ldh, cd* __restrict__ X, int64_t ldx) {
    const int64_t j0 = (int64_t)blockIdx.x * GR_ROWS + threadIdx.x;
    cd* x = X + (int64_t)blockIdx.y * ldx;
    const cd* hc = H + (int64_t)blockIdx.y * ldh;
    int ig[GR_UNR], im[GR_UNR];
    cd h[GR_UNR];
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t j = j0 + u * 256;
        const bool in = j < nh;
        ig[u] = in ? g[j] : 0;
        im[u] = in ? mg[j] : 0;
        h[u] = in ? hc[j] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t j = j0 + u * 256;
        if (j >= nh) continue;
        if (j == 0) {
            x[ig[u]] = make_double2(h[u].x, 0.0);
            continue;
        }
        x[ig[u]] = make_double2(GR_ISQRT2 * h[u].x, GR_ISQRT2 * h[u].y);
        x[im[u]] = make_double2(GR_ISQRT2 * h[u].x, -GR_ISQRT2 * h[u].y);
    }
}

// Z[:, p] = full-sphere image of (a + i b), a = H[:, 2p], b = H[:, 2p + 1] (b = 0 past the last band)
__global__ __launch_bounds__(256) void k_gr_pack(int64_t nh, int nb, const int* __restrict__ g, const int* __restrict__ mg,
                                                 const cd* __restrict__ H, int64_t ldh, cd* __restrict__ Z, int64_t ldz) {
    const int64_t j0 = (int64_t)blockIdx.x * GR_ROWS + threadIdx.x;
    const int p = blockIdx.y;
    const bool has_b = 2 * p + 1 < nb;
    const cd* ha = H + (int64_t)(2 * p) * ldh;
    const cd* hb = H + (int64_t)(has_b ? 2 * p + 1 : 2 * p) * ldh;
    cd* z = Z + (int64_t)p * ldz;
    int ig[GR_UNR], im[GR_UNR];
    cd a[GR_UNR], b[GR_UNR];
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t j = j0 + u * 256;
        const bool in = j < nh;
        ig[u] = in ? g[j] : 0;
        im[u] = in ? mg[j] : 0;
        a[u] = in ? ha[j] : make_double2(0.0, 0.0);
        b[u] = (in && has_b) ? hb[j] : make_double2(0.0, 0.0);
    }
    const double s = GR_ISQRT2;
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t j = j0 + u * 256;
        if (j >= nh) continue;
        if (j == 0) {
            z[ig[u]] = make_double2(a[u].x, b[u].x);
            continue;
        }
        z[ig[u]] = make_double2(s * (a[u].x - b[u].y), s * (a[u].y + b[u].x));       // a + i b
        z[im[u]] = make_double2(s * (a[u].x + b[u].y), s * (b[u].x - a[u].y));      // conj(a) + i conj(b)
    }
}

// inverse of k_gr_pack on the pipeline's output W: A = (W(G) + conj W(-G)) / 2, B = (W(G) - conj W(-G)) / (2i)
__global__ __launch_bounds__(256) void k_gr_unpack(int64_t nh, int nb, const int* __restrict__ g, const int* __restrict__ mg,
                                                   const cd* __restrict__ W, int64_t ldw, cd* __restrict__ H, int64_t ldh) {
    const int64_t j0 = (int64_t)blockIdx.x * GR_ROWS + threadIdx.x;
    const int p = blockIdx.y;
    const cd* w = W + (int64_t)p * ldw;
    int ig[GR_UNR], im[GR_UNR];
    cd wg[GR_UNR], wm[GR_UNR];
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t j = j0 + u * 256;
        ig[u] = j < nh ? g[j] : 0;
        im[u] = j < nh ? mg[j] : 0;
    }
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        wg[u] = w[ig[u]];
        wm[u] = w[im[u]];
    }
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t j = j0 + u * 256;
        if (j >= nh) continue;
        const double s = j ? GR_ISQRT2 : 0.5;            // sqrt(2) / 2, row 0: 1 / 2 (wg == wm)
        H[j + (int64_t)(2 * p) * ldh] = make_double2(s * (wg[u].x + wm[u].x), s * (wg[u].y - wm[u].y));
        if (2 * p + 1 < nb) H[j + (int64_t)(2 * p + 1) * ldh] = make_double2(s * (wg[u].y + wm[u].y), s * (wm[u].x - wg[u].x));
    }
}

// Z[:, p] = X[:, 2p] + i X[:, 2p + 1] on the FULL sphere (density of real-symmetric orbitals: two bands per transform)
__global__ __launch_bounds__(256) void k_gr_pack_full(int64_t n, int nb, const cd* __restrict__ X, int64_t ldx,
                                                      cd* __restrict__ Z, int64_t ldz) {
    const int64_t i0 = (int64_t)blockIdx.x * GR_ROWS + threadIdx.x;
    const int p = blockIdx.y;
    const bool has_b = 2 * p + 1 < nb;
    const cd* xa = X + (int64_t)(2 * p) * ldx;
    const cd* xb = X + (int64_t)(has_b ? 2 * p + 1 : 2 * p) * ldx;
    cd a[GR_UNR], b[GR_UNR];
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t i = i0 + u * 256;
        a[u] = i < n ? xa[i] : make_double2(0.0, 0.0);
        b[u] = (i < n && has_b) ? xb[i] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < GR_UNR; ++u) {
        const int64_t i = i0 + u * 256;
        if (i < n) Z[i + (int64_t)p * ldz] = make_double2(a[u].x - b[u].y, a[u].y + b[u].x);
    }
}

// Ph[j, c] = s_j P[g_j, c]; out[0] = max |P[mg_j, c] - conj(P[g_j, c])|, out[1] = max |P| (bit patterns of
// non-negative doubles order like integers -> atomicMax on the 64-bit image)
__global__ __launch_bounds__(256) void k_gr_gather_P(int64_t nh, const int* __restrict__ g, const int* __restrict__ mg,
                                                     const cd* __restrict__ P, int64_t ldP, cd* __restrict__ Ph, int64_t ldh,
                                                     unsigned long long* __restrict__ out) {
    __shared__ double sh[2][256];
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double asym = 0.0, mag = 0.0;
    if (j < nh) {
        const cd* pc = P + (int64_t)blockIdx.y * ldP;
        const cd a = pc[g[j]], b = pc[mg[j]];
        asym = hypot(b.x - a.x, b.y + a.y);
        mag = hypot(a.x, a.y);
        Ph[j + (int64_t)blockIdx.y * ldh] = j ? make_double2(GR_SQRT2 * a.x, GR_SQRT2 * a.y) : make_double2(a.x, 0.0);
    }
    // one pair of atomics per workgroup (NaNs must not get lost in the max: they compare false, so carry them as +inf)
    sh[0][threadIdx.x] = asym == asym ? asym : INFINITY;
    sh[1][threadIdx.x] = mag == mag ? mag : INFINITY;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] = fmax(sh[0][threadIdx.x], sh[0][threadIdx.x + s]);
            sh[1][threadIdx.x] = fmax(sh[1][threadIdx.x], sh[1][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(out, (unsigned long long)__double_as_longlong(sh[0][0]));
        atomicMax(out + 1, (unsigned long long)__double_as_longlong(sh[1][0]));
    }
}

static dim3 gr_grid(int64_t rows, int cols) { return dim3((unsigned)((rows + 255) / 256), (unsigned)cols); }
// (the kernels that take GR_UNR pairs per thread)
static dim3 gr_grid_unr(int64_t rows, int cols) { return dim3((unsigned)((rows + GR_ROWS - 1) / GR_ROWS), (unsigned)cols); }

int gamma_enable(dftk_mi_kblock* kb, int on) {
    dftk_mi_basis* b = kb->basis;
    if (!on) {
        if (kb->gr) kb->gr->on = false;
        return 0;
    }
    if (kb->gr && kb->gr->d_g) {
        kb->gr->on = true;
        return 0;
    }
    if (!kb->h_mapping || !kb->h_kin) return DFTK_MI_EINVAL;
    std::vector<int32_t> g((size_t)(kb->n_G + 1) / 2 + 1), mg(g.size());
    int64_t nh = 0;
    CHK(gamma_tables_host(b->nx, b->ny, b->nz, kb->n_G, kb->h_mapping->data(), &nh, g.data(), mg.data()));
    std::vector<double> kin((size_t)nh);
    for (int64_t j = 0; j < nh; ++j) {
        const double a = (*kb->h_kin)[g[j]], c = (*kb->h_kin)[mg[j]];
        if (std::fabs(a - c) > 1e-12 * (1.0 + std::fabs(a))) {
            dftk_set_error("gamma_real: kinetic energies of G and -G differ (k != 0?)");
            return DFTK_MI_EINVAL;
        }
        kin[j] = a;
    }
    GammaReal* gr = kb->gr ? kb->gr : new GammaReal();   // (gamma_density may have created the scratch holder)
    gr->n_half = nh;
    HIPCHK(hipMalloc((void**)&gr->d_g, nh * sizeof(int)));
    HIPCHK(hipMalloc((void**)&gr->d_mg, nh * sizeof(int)));
    HIPCHK(hipMalloc((void**)&gr->d_kin_half, nh * sizeof(double)));
    HIPCHK(hipMemcpy(gr->d_g, g.data(), nh * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(gr->d_mg, mg.data(), nh * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(gr->d_kin_half, kin.data(), nh * sizeof(double), hipMemcpyHostToDevice));
    // plane-wave sharded block (set_shard comes first): the half-format rows are split evenly over the ranks
    gr->half_rows.clear();
    if (kb->sh_comm) {
        const int p = comm_size(kb->sh_comm);
        if (nh < p) {
            dftk_set_error("gamma_real: more ranks than half-sphere rows");
            if (!kb->gr) delete gr;
            return DFTK_MI_EINVAL;
        }
        gr->half_rows.assign(p + 1, 0);
        const int64_t base = nh / p, rem = nh % p;
        for (int r = 0; r < p; ++r) gr->half_rows[r + 1] = gr->half_rows[r] + base + (r < rem ? 1 : 0);
    }
    gr->on = true;
    kb->gr = gr;
    return 0;
}

int64_t gamma_local_rows(const dftk_mi_kblock* kb) {
    const GammaReal* gr = kb->gr;
    if (!kb->sh_comm) return gr->n_half;
    const int r = comm_rank(kb->sh_comm);
    return gr->half_rows[r + 1] - gr->half_rows[r];
}
int64_t gamma_row0(const dftk_mi_kblock* kb) { return kb->sh_comm ? kb->gr->half_rows[comm_rank(kb->sh_comm)] : 0; }

void gamma_destroy(GammaReal* gr) {
    if (!gr) return;
    void* ptrs[] = {gr->d_g, gr->d_mg, gr->d_kin_half, gr->P_half, gr->buf};
    for (void* p : ptrs)
        if (p) hipFree(p);
    delete gr;
}

int gamma_ensure_buf(dftk_mi_kblock* kb, size_t elems) {
    GammaReal* gr = kb->gr;
    const size_t need = elems * sizeof(cd);
    if (need <= gr->buf_bytes) return 0;
    HIPCHK(hipStreamSynchronize(kb->basis->stream));
    if (gr->buf) HIPCHK(hipFree(gr->buf));
    gr->buf = nullptr;
    gr->buf_bytes = 0;
    HIPCHK(dftk_scratch_malloc((void**)&gr->buf, need));
    gr->buf_bytes = need;
    return 0;
}

int gamma_compress(dftk_mi_kblock* kb, int m, const cd* X, int64_t ldx, cd* H, int64_t ldh) {
    if (m <= 0) return 0;
    GammaReal* gr = kb->gr;
    ProfScope prof_scope(kb->basis, PROF_EW, 48.0 * (double)gr->n_half * m);
    hipLaunchKernelGGL(k_gr_compress, gr_grid_unr(gr->n_half, m), dim3(256), 0, kb->basis->stream, gr->n_half, gr->d_g,
                       gr->d_mg, X, ldx, H, ldh);
    HIPCHK(hipGetLastError());
    return 0;
}

int gamma_compress_aligned(dftk_mi_kblock* kb, int m, const cd* X, int64_t ldx, cd* H, int64_t ldh) {
    if (m <= 0) return 0;
    GammaReal* gr = kb->gr;
    ProfScope prof_scope(kb->basis, PROF_EW, 64.0 * (double)gr->n_half * m);
    hipLaunchKernelGGL(k_gr_compress_aligned, dim3(m), dim3(GRA_NT), 0, kb->basis->stream, gr->n_half, gr->d_g, gr->d_mg, X, ldx,
                       H, ldh);
    HIPCHK(hipGetLastError());
    return 0;
}

int gamma_expand(dftk_mi_kblock* kb, int m, const cd* H, int64_t ldh, cd* X, int64_t ldx) {
    if (m <= 0) return 0;
    GammaReal* gr = kb->gr;
    ProfScope prof_scope(kb->basis, PROF_EW, 48.0 * (double)gr->n_half * m);
    hipLaunchKernelGGL(k_gr_expand, gr_grid_unr(gr->n_half, m), dim3(256), 0, kb->basis->stream, gr->n_half, gr->d_g,
                       gr->d_mg, H, ldh, X, ldx);
    HIPCHK(hipGetLastError());
    return 0;
}

// half-format projectors (built on first use after dftk_mi_kblock_set_projectors); refuses projectors that are
// not Fourier transforms of real functions
int gamma_gather_P(dftk_mi_kblock* kb, int ncols, const cd* P, int64_t ldP, cd* Ph, int64_t ldh, double* asym_mag_h) {
    GammaReal* gr = kb->gr;
    dftk_mi_basis* b = kb->basis;
    unsigned long long* d_out = reinterpret_cast<unsigned long long*>(b->d_scalars);
    HIPCHK(hipMemsetAsync(d_out, 0, 2 * sizeof(unsigned long long), b->stream));
    if (ncols > 0) {
        hipLaunchKernelGGL(k_gr_gather_P, gr_grid(gr->n_half, ncols), dim3(256), 0, b->stream, gr->n_half, gr->d_g,
                           gr->d_mg, P, ldP, Ph, ldh, d_out);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(asym_mag_h, d_out, 2 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

static int gr_projectors(dftk_mi_kblock* kb) {
    GammaReal* gr = kb->gr;
    if (gr->P_src == kb->P && gr->P_n_p == kb->n_p && gr->P_half) return 0;
    if (kb->sh_comm) return gamma_projectors_sharded(kb);
    dftk_mi_basis* b = kb->basis;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (gr->P_half) HIPCHK(hipFree(gr->P_half));
    gr->P_half = nullptr;
    HIPCHK(hipMalloc((void**)&gr->P_half, (size_t)gr->n_half * kb->n_p * sizeof(cd)));
    double h[2];
    CHK(gamma_gather_P(kb, kb->n_p, kb->P, kb->ldP, gr->P_half, gr->n_half, h));
    if (!(std::isfinite(h[1]) && h[0] <= 1e-10 * (h[1] > 0 ? h[1] : 1.0))) {
        dftk_set_error("gamma_real: the projectors are not real-symmetric (max |P(-G) - conj P(G)| = %.3e, max |P| = %.3e)",
                       h[0], h[1]);
        HIPCHK(hipFree(gr->P_half));
        gr->P_half = nullptr;
        return DFTK_MI_EINVAL;
    }
    gr->P_src = kb->P;
    gr->P_n_p = kb->n_p;
    return 0;
}

int gamma_pack_pairs(dftk_mi_kblock* kb, int nb, const cd* H, int64_t ldh, cd* Z, int64_t ldz) {
    if (nb <= 0) return 0;
    GammaReal* gr = kb->gr;
    ProfScope prof_scope(kb->basis, PROF_EW, 32.0 * (double)gr->n_half * nb);
    hipLaunchKernelGGL(k_gr_pack, gr_grid_unr(gr->n_half, (nb + 1) / 2), dim3(256), 0, kb->basis->stream, gr->n_half, nb,
                       gr->d_g, gr->d_mg, H, ldh, Z, ldz);
    HIPCHK(hipGetLastError());
    return 0;
}
int gamma_unpack_pairs(dftk_mi_kblock* kb, int nb, const cd* W, int64_t ldw, cd* H, int64_t ldh) {
    if (nb <= 0) return 0;
    GammaReal* gr = kb->gr;
    ProfScope prof_scope(kb->basis, PROF_EW, 32.0 * (double)gr->n_half * nb);
    hipLaunchKernelGGL(k_gr_unpack, gr_grid_unr(gr->n_half, (nb + 1) / 2), dim3(256), 0, kb->basis->stream, gr->n_half, nb,
                       gr->d_g, gr->d_mg, W, ldw, H, ldh);
    HIPCHK(hipGetLastError());
    return 0;
}
int gamma_pack_full(dftk_mi_kblock* kb, int nb, const cd* X, int64_t ldx, cd* Z, int64_t ldz) {
    if (nb <= 0) return 0;
    ProfScope prof_scope(kb->basis, PROF_EW, 32.0 * (double)kb->n_G * nb);
    hipLaunchKernelGGL(k_gr_pack_full, gr_grid_unr(kb->n_G, (nb + 1) / 2), dim3(256), 0, kb->basis->stream, kb->n_G, nb, X,
                       ldx, Z, ldz);
    HIPCHK(hipGetLastError());
    return 0;
}

// H psi in the half-sphere format (which: bit 0 local, 1 kinetic, 2 nonlocal as dftk_mi_apply_H_parts)
int gamma_apply_H(dftk_mi_kblock* kb, int which, int nb, const cd* psi, int64_t ldpsi, cd* Hpsi, int64_t ldH) {
    if (nb <= 0) return 0;
    if (kb->sh_comm) return gamma_apply_H_sharded(kb, which, nb, psi, ldpsi, Hpsi, ldH);
    GammaReal* gr = kb->gr;
    dftk_mi_basis* b = kb->basis;
    const int nb2 = (nb + 1) / 2;
    const bool local = (which & 1) && kb->d_Vs != nullptr;
    const bool kinetic = which & 2;
    const int slot = prof_begin(b, PROF_APPLY_H, (double)nb);
    struct G {
        dftk_mi_basis* b;
        int s;
        ~G() { prof_end(b, s); }
    } guard{b, slot};
    CHK(gamma_ensure_buf(kb, 2 * (size_t)kb->n_G * nb2));
    cd* Z = gr->buf;
    cd* W = gr->buf + (size_t)kb->n_G * nb2;
    // (a plane-wave-sharded run transposes the slab of nb half-format bands to whole bands and back around this pipeline)
    prof_count(b, PROF_A2A_MODEL, 16.0 * (double)gr->n_half * nb);   // slabs -> bands
    prof_count(b, PROF_A2A_MODEL, 16.0 * (double)gr->n_half * nb);   // bands -> slabs
    if ((which & 4) && kb->n_p > 0) prof_count(b, PROF_AR_MODEL, 16.0 * (double)kb->n_p * nb);   // P' psi partial sums
    CHK(gamma_pack_pairs(kb, nb, psi, ldpsi, Z, kb->n_G));
    CHK(launch_local_apply(kb, nb2, Z, kb->n_G, W, kb->n_G, kinetic, local));
    CHK(gamma_unpack_pairs(kb, nb, W, kb->n_G, Hpsi, ldH));
    if ((which & 4) && kb->n_p > 0) {
        CHK(gr_projectors(kb));
        CHK(apply_nonlocal_rows(kb, nb, gr->P_half, gr->n_half, gr->n_half, psi, ldpsi, Hpsi, ldH, true,
                                DFTK_MI_GEMM_REAL, nullptr));
    }
    return 0;
}

// rho += sum_band w |IFFT psi|^2 for REAL-SYMMETRIC columns in the full-sphere layout: bands 2p, 2p + 1 share one
// transform (the real part of the transformed pair is band 2p, the imaginary part band 2p + 1)
int gamma_density(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, const double* w_h, double* rho) {
    if (nb <= 0) return 0;
    if (!kb->gr) {                       // only the scratch buffer of the structure is needed here
        kb->gr = new GammaReal();
        kb->gr->on = false;
    }
    if (kb->sh_comm) return gamma_density_sharded(kb, nb, psi, ldpsi, w_h, rho);
    return gamma_density_bands(kb, nb, psi, ldpsi, w_h, rho);
}

// whole bands on this rank (full-sphere layout)
int gamma_density_bands(dftk_mi_kblock* kb, int nb, const cd* psi, int64_t ldpsi, const double* w_h, double* rho) {
    if (nb <= 0) return 0;
    const int nb2 = (nb + 1) / 2;
    CHK(gamma_ensure_buf(kb, (size_t)kb->n_G * nb2));
    CHK(gamma_pack_full(kb, nb, psi, ldpsi, kb->gr->buf, kb->n_G));
    std::vector<double> wre(nb2), wim(nb2);
    for (int p = 0; p < nb2; ++p) {
        wre[p] = w_h[2 * p];
        wim[p] = (2 * p + 1 < nb) ? w_h[2 * p + 1] : 0.0;
    }
    return launch_density(kb, nb2, kb->gr->buf, kb->n_G, wre.data(), rho, wim.data());
}
