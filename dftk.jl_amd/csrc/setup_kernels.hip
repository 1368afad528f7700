// setup_kernels.hip -- basis / term set-up behind the ABI:
//   Kpoint sphere enumeration      src/Kpoint.jl:20-41 (+ kinetic multipliers, src/terms/kinetic.jl:31-35)
//   build_projection_vectors       src/terms/nonlocal.jl:166-244 for HGH pseudopotentials
//                                  (radial parts src/pseudo/PspHgh.jl:140-164, real solid harmonics
//                                   src/common/spherical_harmonics.jl:31-66)
// The reference runs both as host loops (O(N) and O(n_G n_p), minutes for 1000-electron cells).
// Here the sphere is one pass over the cube in index order on the host (native, no cube-sized temporaries) and the
// projector matrix is written by one device kernel, 16 B per element, straight into the caller's P.
#include "common.h"
#include <cmath>
#include <vector>

// ------------------------------------------------------------------------------------------------ sphere (host)
// G_axis(n)[i]: [0 .. floor((n-1)/2), -ceil((n-1)/2) .. -1]   (src/fft.jl:24-31)
static inline int g_of(int i, int n) { return i <= (n - 1) / 2 ? i : i - n; }

#pragma clang fp contract(off)
int sphere_enumerate_host(int nx, int ny, int nz, const double* B /* recip_lattice, column-major 3x3 */,
                          const double* k, double Ecut, int64_t cap, int64_t* n_G_out, int64_t* mapping0,
                          double* kinetic, int32_t* G_out) {
    int64_t n = 0;
    for (int iz = 0; iz < nz; ++iz) {
        const double gz = (double)g_of(iz, nz) + k[2];
        for (int iy = 0; iy < ny; ++iy) {
            const double gy = (double)g_of(iy, ny) + k[1];
            for (int ix = 0; ix < nx; ++ix) {
                const double gx = (double)g_of(ix, nx) + k[0];
                // B (G + k) spelled out column by column, the same operation order as the host mirror / the oracle
                double s = 0.0;
                for (int c = 0; c < 3; ++c) {
                    const double q = gx * B[c + 0] + gy * B[c + 3] + gz * B[c + 6];
                    s += q * q;
                }
                const double kin = s / 2;
                if (kin <= Ecut) {
                    if (n < cap) {
                        if (mapping0) mapping0[n] = (int64_t)ix + (int64_t)nx * ((int64_t)iy + (int64_t)ny * iz);
                        if (kinetic) kinetic[n] = kin;
                        if (G_out) {
                            G_out[3 * n + 0] = g_of(ix, nx);
                            G_out[3 * n + 1] = g_of(iy, ny);
                            G_out[3 * n + 2] = g_of(iz, nz);
                        }
                    }
                    n += 1;
                }
            }
        }
    }
    *n_G_out = n;
    if (n > cap && (mapping0 || kinetic || G_out)) {
        dftk_set_error("sphere has %lld plane waves, the buffers hold %lld", (long long)n, (long long)cap);
        return DFTK_MI_EINVAL;
    }
    return 0;
}
#pragma clang fp contract(on)

// ------------------------------------------------------------------------------------------------ projectors (device)
struct ProjCol {       // one column of P
    double rx, ry, rz;   // atom position (reduced)
    double rp;           // r_l of the species
    int l, m, i;         // angular momentum, magnetic index, radial index (1-based)
};

__device__ __forceinline__ double hgh_radial(int l, int i, double rp, double p) {
    // eval_psp_projector_fourier (PspHgh.jl:140-164), includes the division by p^l
    const double t2 = (p * rp) * (p * rp);
    const double common = 4.0 * pow(M_PI, 1.25) * sqrt(ldexp(1.0, l + 1) * rp * rp * rp) * exp(-t2 / 2.0);
    switch (l * 4 + i) {
        case 0 * 4 + 1: return common;
        case 0 * 4 + 2: return common * (2.0 / sqrt(15.0)) * (3.0 - t2);
        case 0 * 4 + 3: return common * (4.0 / (3.0 * sqrt(105.0))) * (15.0 - 10.0 * t2 + t2 * t2);
        case 1 * 4 + 1: return common * (rp / sqrt(3.0));
        case 1 * 4 + 2: return common * (2.0 * rp / sqrt(105.0)) * (5.0 - t2);
        case 1 * 4 + 3: return common * (4.0 * rp / (3.0 * sqrt(1155.0))) * (35.0 - 14.0 * t2 + t2 * t2);
        case 2 * 4 + 1: return common * (rp * rp / sqrt(15.0));
        case 2 * 4 + 2: return common * (2.0 * rp * rp / (3.0 * sqrt(105.0))) * (7.0 - t2);
        case 3 * 4 + 1: return common * (rp * rp * rp / sqrt(105.0));
        default: return nan("");
    }
}

__device__ __forceinline__ double solid_harmonic(int l, int m, double x, double y, double z) {
    // r^l Y_lm, real form (spherical_harmonics.jl:31-66)
    const double pi = M_PI;
    if (l == 0) return sqrt(1.0 / (4.0 * pi));
    if (l == 1) return sqrt(3.0 / (4.0 * pi)) * (m == -1 ? y : (m == 0 ? z : x));
    if (l == 2) {
        switch (m) {
            case -2: return sqrt(15.0 / (4.0 * pi)) * x * y;
            case -1: return sqrt(15.0 / (4.0 * pi)) * y * z;
            case 0: return sqrt(5.0 / (16.0 * pi)) * (2.0 * z * z - x * x - y * y);
            case 1: return sqrt(15.0 / (4.0 * pi)) * x * z;
            default: return sqrt(15.0 / (16.0 * pi)) * (x * x - y * y);
        }
    }
    switch (m) {
        case -3: return sqrt(35.0 / (32.0 * pi)) * (3.0 * x * x - y * y) * y;
        case -2: return sqrt(105.0 / (4.0 * pi)) * x * y * z;
        case -1: return sqrt(21.0 / (32.0 * pi)) * y * (4.0 * z * z - x * x - y * y);
        case 0: return sqrt(7.0 / (16.0 * pi)) * z * (2.0 * z * z - 3.0 * x * x - 3.0 * y * y);
        case 1: return sqrt(21.0 / (32.0 * pi)) * x * (4.0 * z * z - x * x - y * y);
        case 2: return sqrt(105.0 / (16.0 * pi)) * (x * x - y * y) * z;
        default: return sqrt(35.0 / (32.0 * pi)) * (x * x - 3.0 * y * y) * x;
    }
}

struct Mat3 {
    double b[9];   // column-major
};

// P[g, c] = radial_{l,i}(|q|) * Y_lm(q) * (-i)^l / sqrt(Omega) * exp(-2 pi i (G + k).r),  q = B (G + k)
__global__ __launch_bounds__(256) void k_build_projectors(int64_t n_rows, int n_cols, const int32_t* __restrict__ G,
                                                          Mat3 B, double kx, double ky, double kz, double inv_sqrt_vol,
                                                          const ProjCol* __restrict__ cols, cd* __restrict__ P,
                                                          int64_t ldP) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_rows) return;
    const double px = (double)G[3 * g + 0] + kx, py = (double)G[3 * g + 1] + ky, pz = (double)G[3 * g + 2] + kz;
    const double qx = px * B.b[0] + py * B.b[3] + pz * B.b[6];
    const double qy = px * B.b[1] + py * B.b[4] + pz * B.b[7];
    const double qz = px * B.b[2] + py * B.b[5] + pz * B.b[8];
    const double qn = sqrt(qx * qx + qy * qy + qz * qz);
    for (int c = blockIdx.y; c < n_cols; c += gridDim.y) {
        const ProjCol pc = cols[c];
        double f = hgh_radial(pc.l, pc.i, pc.rp, qn) * solid_harmonic(pc.l, pc.m, qx, qy, qz) * inv_sqrt_vol;
        // (-i)^l : 1, -i, -1, i
        double fr, fi;
        switch (pc.l & 3) {
            case 0: fr = f; fi = 0.0; break;
            case 1: fr = 0.0; fi = -f; break;
            case 2: fr = -f; fi = 0.0; break;
            default: fr = 0.0; fi = f; break;
        }
        const double ph = -2.0 * M_PI * (px * pc.rx + py * pc.ry + pz * pc.rz);
        double sn, cs;
        sincos(ph, &sn, &cs);
        P[(int64_t)c * ldP + g] = make_double2(fr * cs - fi * sn, fr * sn + fi * cs);
    }
}

// columns: atom-major in the caller's order; within an atom (l, m, i): offset_l + n_l (m + l) + (i - 1)
// (nonlocal.jl:205-244).  rp_h / nproj_h: 4 entries per species (l = 0..3).
int build_projectors_hgh(dftk_mi_basis* b, int64_t n_rows, const int32_t* G_d, const double* recip_h, const double* k_h,
                         double volume, int n_species, const double* rp_h, const int* nproj_h, int n_atoms,
                         const int* species_of_atom_h, const double* positions_h, cd* P_d, int64_t ldP, int* n_p_out) {
    std::vector<ProjCol> cols;
    for (int a = 0; a < n_atoms; ++a) {
        const int s = species_of_atom_h[a];
        if (s < 0 || s >= n_species) return DFTK_MI_EINVAL;
        for (int l = 0; l < 4; ++l) {
            const int nl = nproj_h[4 * s + l];
            if (nl < 0 || nl > 3 || (l == 2 && nl > 2) || (l == 3 && nl > 1)) {
                dftk_set_error("HGH projector l=%d with %d radial functions is not tabulated", l, nl);
                return DFTK_MI_EINVAL;
            }
            for (int m = -l; m <= l; ++m)
                for (int i = 1; i <= nl; ++i)
                    cols.push_back(ProjCol{positions_h[3 * a], positions_h[3 * a + 1], positions_h[3 * a + 2],
                                           rp_h[4 * s + l], l, m, i});
        }
    }
    *n_p_out = (int)cols.size();
    if (cols.empty() || !P_d) return 0;
    if (ldP < n_rows) return DFTK_MI_EINVAL;
    CHK(ensure_ws(b, cols.size() * sizeof(ProjCol)));
    HIPCHK(hipMemcpyAsync(b->ws, cols.data(), cols.size() * sizeof(ProjCol), hipMemcpyHostToDevice, b->stream));
    Mat3 B;
    for (int i = 0; i < 9; ++i) B.b[i] = recip_h[i];
    const unsigned gy = (unsigned)std::min<size_t>(cols.size(), 64);
    hipLaunchKernelGGL(k_build_projectors, dim3((unsigned)((n_rows + 255) / 256), gy), dim3(256), 0, b->stream, n_rows,
                       (int)cols.size(), G_d, B, k_h[0], k_h[1], k_h[2], 1.0 / sqrt(volume),
                       reinterpret_cast<const ProjCol*>(b->ws), P_d, ldP);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));   // cols (host vector) and b->ws are reused by the next call
    return 0;
}

// ------------------------------------------------------------------------------------------------ atomic superpositions
// f(G) = sum_species ff_s(|G|) sum_{a in s} e^{-2 pi i G.r_a} / sqrt(Omega) on the whole cube, entries whose -G partner
// is not on the grid zeroed (enforce_real!, symmetry.jl:318-337), then the inverse cube FFT:
//   kind 0: compute_local_potential (src/terms/local.jl:108-138) with the HGH local form factor
//           (eval_psp_local_fourier, src/pseudo/PspHgh.jl:110-124); params = {rloc, Zion, c1, c2, c3, c4}
//   kind 1: Gaussian valence-density superposition (src/density_methods.jl:111-125,158-181,236-244);
//           params = {decay length, valence charge}
struct AtomPar {
    double rx, ry, rz;
    int species;
};
__global__ __launch_bounds__(256) void k_atomic_sum(int nx, int ny, int nz, Mat3 B, int kind, int n_atoms,
                                                    const AtomPar* __restrict__ atoms, const double* __restrict__ par,
                                                    double inv_sqrt_vol, cd* __restrict__ out) {
    const int64_t N = (int64_t)nx * ny * nz;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N) return;
    const int ix = (int)(idx % nx), iy = (int)((idx / nx) % ny), iz = (int)(idx / ((int64_t)nx * ny));
    const bool unpaired = ((nx % 2 == 0) && ix == nx / 2) || ((ny % 2 == 0) && iy == ny / 2) || ((nz % 2 == 0) && iz == nz / 2);
    if (unpaired) {
        out[idx] = make_double2(0.0, 0.0);
        return;
    }
    const double gx = (double)(ix <= (nx - 1) / 2 ? ix : ix - nx), gy = (double)(iy <= (ny - 1) / 2 ? iy : iy - ny),
                 gz = (double)(iz <= (nz - 1) / 2 ? iz : iz - nz);
    const double qx = gx * B.b[0] + gy * B.b[3] + gz * B.b[6];
    const double qy = gx * B.b[1] + gy * B.b[4] + gz * B.b[7];
    const double qz = gx * B.b[2] + gy * B.b[5] + gz * B.b[8];
    const double p = sqrt(qx * qx + qy * qy + qz * qz);
    double re = 0.0, im = 0.0;
    int cur = -1;
    double ff = 0.0;
    for (int a = 0; a < n_atoms; ++a) {
        const AtomPar at = atoms[a];
        if (at.species != cur) {      // atoms arrive grouped by species: one form-factor evaluation per species
            cur = at.species;
            const double* q = par + 8 * cur;
            if (kind == 0) {
                const double rloc = q[0], Zion = q[1];
                const double t2 = (p * rloc) * (p * rloc);
                if (t2 > 0.0) {
                    const double P = q[2] + q[3] * (3.0 - t2) + q[4] * (15.0 - 10.0 * t2 + t2 * t2) +
                                     q[5] * (105.0 - 105.0 * t2 + 21.0 * t2 * t2 - t2 * t2 * t2);
                    ff = 4.0 * M_PI * rloc * rloc * (-Zion + sqrt(M_PI / 2.0) * rloc * t2 * P) * exp(-t2 / 2.0) / t2;
                } else {
                    ff = 0.0;                  // compensating background
                }
            } else {
                const double x = p * q[0];
                ff = q[1] * exp(-x * x);
            }
            ff *= inv_sqrt_vol;
        }
        double sn, cs;
        sincos(-2.0 * M_PI * (gx * at.rx + gy * at.ry + gz * at.rz), &sn, &cs);
        re += ff * cs;
        im += ff * sn;
    }
    out[idx] = make_double2(re, im);
}

__global__ __launch_bounds__(256) void k_real_part_scaled(int64_t n, const cd* __restrict__ c, double scale,
                                                          double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = scale * c[i].x;
}

int atomic_superposition(dftk_mi_kblock* cube_kb, int kind, const double* recip_h, int n_species, const double* par_h,
                         int n_atoms, const int* species_of_atom_h, const double* positions_h, double* out_d) {
    dftk_mi_basis* b = cube_kb->basis;
    const int64_t N = (int64_t)b->nx * b->ny * b->nz;
    if (cube_kb->n_G != N) {
        dftk_set_error("atomic_superposition: the k-block must span the whole cube");
        return DFTK_MI_EINVAL;
    }
    std::vector<AtomPar> atoms(n_atoms);
    for (int a = 0; a < n_atoms; ++a) {
        if (species_of_atom_h[a] < 0 || species_of_atom_h[a] >= n_species) return DFTK_MI_EINVAL;
        if (a > 0 && species_of_atom_h[a] < species_of_atom_h[a - 1]) {
            dftk_set_error("atomic_superposition: atoms must be grouped by species");
            return DFTK_MI_EINVAL;
        }
        atoms[a] = AtomPar{positions_h[3 * a], positions_h[3 * a + 1], positions_h[3 * a + 2], species_of_atom_h[a]};
    }
    const size_t need = 2 * (size_t)N * sizeof(cd);
    if (need > b->dense_ws_bytes) {
        HIPCHK(hipStreamSynchronize(b->stream));
        if (b->dense_ws) HIPCHK(hipFree(b->dense_ws));
        b->dense_ws = nullptr;
        b->dense_ws_bytes = 0;
        HIPCHK(dftk_scratch_malloc(&b->dense_ws, need));
        b->dense_ws_bytes = need;
    }
    cd* c1 = reinterpret_cast<cd*>(b->dense_ws);
    cd* c2 = c1 + N;
    const size_t tab = (size_t)n_atoms * sizeof(AtomPar) + (size_t)n_species * 8 * sizeof(double);
    CHK(ensure_ws(b, tab));
    AtomPar* d_atoms = reinterpret_cast<AtomPar*>(b->ws);
    double* d_par = reinterpret_cast<double*>(d_atoms + n_atoms);
    HIPCHK(hipMemcpyAsync(d_atoms, atoms.data(), (size_t)n_atoms * sizeof(AtomPar), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(d_par, par_h, (size_t)n_species * 8 * sizeof(double), hipMemcpyHostToDevice, b->stream));
    Mat3 B;
    for (int i = 0; i < 9; ++i) B.b[i] = recip_h[i];
    hipLaunchKernelGGL(k_atomic_sum, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, b->stream, b->nx, b->ny, b->nz, B,
                       kind, n_atoms, d_atoms, d_par, 1.0 / sqrt(b->volume), c1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));      // host tables and b->ws are free again (the FFT below reuses ws-free paths)
    CHK(launch_ifft_to_cube(cube_kb, c1, c2));
    hipLaunchKernelGGL(k_real_part_scaled, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, b->stream, N, c2,
                       1.0 / sqrt(b->volume), out_d);   // ifft_normalization (fft.jl:87)
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
