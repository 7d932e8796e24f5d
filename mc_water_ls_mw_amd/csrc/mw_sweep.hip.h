// mw_sweep.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine:
// device-resident Monte Carlo driver: mc_water_translation, mc_volume,
// mc_update_wl_bins, mc_lattice_switch of mc_moves.F90 for many independent walkers.
#pragma once

#include "mw_common.hip.h"
#include "mw_full_energy.hip.h"
#include "mw_move_energy.hip.h"

namespace mw {

// =====================================================================================
// Device-resident translation-move driver: mc_water_translation
// (mc_moves.F90:966-1213) with eta_weight (:893-964) and mu_to_bin (:2187-2215), for many
// independent walkers at once.  One wavefront per walker runs its Markov chain move after
// move: pick a molecule, draw the displacement in the active lattice, map it through
// fractional coordinates into the partner lattice (:1042-1066), fused old/new local energy in
// each lattice (move_energy_wave), update the order parameter mu and the multicanonical
// weights' contribution, accept or revert (:1145-1209).  The caller-side bookkeeping of
// model_energy (:1013-1016,1087,1190) is done here on the per-box energies.
// Random numbers: Philox4x32-10, counter (move lo, move hi, walker, call), key = seed -- the same
// stream as the oracle's mwo_move_uniforms.
//   grid = walkers in the launch, block = 64
// =====================================================================================
struct SweepParams {
    double beta, max_trans;
    double r_pos, a_pos, r_neg, a_neg, mu_lo, mu_hi;
    int nlat, nbins, eta_interp, start_bin, end_bin, pad;
    // the rest of a translation-only mc_cycle (all off by default)
    int record, samplerun, always_switch, npt;      // mc_update_wl_bins active / fixed weights / switch after every move / ensemble
    double av_binwidth, wl_factor, log_unbiased_norm, pressure;
    double transP, dv_max;                          // move-type threshold (mc_moves.F90:157-166), max cell-element change
    // leshift (userparams.f90:41): ref_enthalpy(1) - ref_enthalpy(2), 0 when off (main.f90:146-150,173; mc_moves.F90:1371,1567-1584)
    double dref;
    // wl_swetnam (mc_moves.F90:1636-1653): the increment follows the histogram's r.m.s. deviation from flat, move by move
    int swetnam, dd;                                // dd: parallel_strategy = 'dd' (window per walker, mc_moves.F90:181-210,659-709)
    double wl_alpha, orig_wl_factor, mu_min, mu_max;
    int eq_cycles, in_window;                       // dd: equilibration length (cycles); in_window: this walker's flag (filled per walker)
    // the reference's -DMINU build (mc_moves.F90:1119-1140,1168-1170,1385-1401,1426-1429): an accepted move also takes the
    // walker to the lattice of lower enthalpy; ref1/ref2 = ref_enthalpy(1:2) under leshift, 0 otherwise
    int minu, pad_minu;
    double ref1, ref2;
};

// The MINU branch of both move types: the lattice the move would end in; diffkT rewritten with the switch's terms if it differs.
// E = trial energies, V = trial volumes, Eb / Vb = energy and volume of the CURRENT lattice before the move.
__device__ __forceinline__ int dev_minu_branch(const SweepParams& sp, int ls, double E1, double E2, double V1, double V2,
                                               double Eb, double Vb, bool vol_terms, int N, double new_eta, double old_eta,
                                               double& diffkT)
{
    const double h1 = E1 + sp.pressure * V1 - sp.ref1, h2 = E2 + sp.pressure * V2 - sp.ref2;   // minloc, :1122-1126
    const int lsn = h2 < h1 ? 2 : 1;
    if (lsn != ls) {
        const double En = lsn == 1 ? E1 : E2, Vn = lsn == 1 ? V1 : V2;
        double d;
        if (vol_terms) d = sp.beta * En - sp.beta * Eb + sp.beta * sp.pressure * (Vn - Vb) - (double)N * log(Vn / Vb) + new_eta - old_eta;   // :1131-1133,1396-1397
        else           d = sp.beta * En - sp.beta * Eb + new_eta - old_eta;                                                                  // :1135
        if (sp.ref1 != 0.0 || sp.ref2 != 0.0)                                                                                               // leshift, :1134,1136,1398
            d = d - sp.beta * (lsn == 1 ? sp.ref1 : sp.ref2) + sp.beta * (ls == 1 ? sp.ref1 : sp.ref2);
        diffkT = d;
    }
    return lsn;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ int dev_mu_to_bin(const SweepParams& g, double mu)          // mc_moves.F90:2187-2215
{
    if (fabs(mu) <= 0.5) return g.nbins / 2 + 1;
    if (mu > 0.0) {
        const double arg = 1.0 - (mu - 0.5) * (1.0 - g.r_pos) / g.a_pos;
        return g.nbins / 2 + 2 + (int)(log(arg) / log(g.r_pos));
    }
    const double arg = 1.0 - (fabs(mu) - 0.5) * (1.0 - g.r_neg) / g.a_neg;
    return g.nbins / 2 - (int)(log(arg) / log(g.r_neg));
}

__device__ __forceinline__ double dev_eta_weight(const SweepParams& g, const double* weight,
                                                 const double* __restrict__ mu_bin, const double* __restrict__ binwidth,
                                                 double mu)                                // mc_moves.F90:893-964
{
    // 'dd' walkers that have not reached their window yet carry no weight: the reference returns here without
    // assigning the function result (:913); 0 is what its comment asks for ("don't want to penalise walkers")
    if (!g.in_window) return 0.0;
    if (mu < g.mu_lo || mu > g.mu_hi) return 1.7976931348623157e308;                       // huge(1.0_dp)
    const int k = dev_mu_to_bin(g, mu);
    const double* w = weight - 1; const double* mb = mu_bin - 1; const double* bw = binwidth - 1;   // 1-based views
    if (!g.eta_interp) return w[k];
    if (k == g.start_bin) return w[k] + (mu - mb[k]) * (2.0 * (w[k + 1] - w[k]) / (bw[k] + bw[k + 1]));
    if (k == g.end_bin)   return w[k] + (mu - mb[k]) * (2.0 * (w[k] - w[k - 1]) / (bw[k] + bw[k - 1]));
    if (mu > mb[k])       return w[k] + (mu - mb[k]) * (2.0 * (w[k + 1] - w[k]) / (bw[k] + bw[k + 1]));
    return w[k - 1] + (mu - mb[k - 1]) * (2.0 * (w[k] - w[k - 1]) / (bw[k] + bw[k - 1]));
}

#define MW_HM(m, r, c) ((m)[((c) - 1) * 3 + ((r) - 1)])     // Fortran (r,c) of a column-major 3x3
__device__ __forceinline__ void dev_recipmatrix(const double* __restrict__ h, double rc[9])   // util.f90:43-77
{
    MW_HM(rc,1,1) = MW_HM(h,2,2)*MW_HM(h,3,3) - MW_HM(h,2,3)*MW_HM(h,3,2);
    MW_HM(rc,1,2) = MW_HM(h,2,3)*MW_HM(h,3,1) - MW_HM(h,2,1)*MW_HM(h,3,3);
    MW_HM(rc,1,3) = MW_HM(h,2,1)*MW_HM(h,3,2) - MW_HM(h,2,2)*MW_HM(h,3,1);
    MW_HM(rc,2,1) = MW_HM(h,1,3)*MW_HM(h,3,2) - MW_HM(h,1,2)*MW_HM(h,3,3);
    MW_HM(rc,2,2) = MW_HM(h,1,1)*MW_HM(h,3,3) - MW_HM(h,1,3)*MW_HM(h,3,1);
    MW_HM(rc,2,3) = MW_HM(h,1,2)*MW_HM(h,3,1) - MW_HM(h,1,1)*MW_HM(h,3,2);
    MW_HM(rc,3,1) = MW_HM(h,1,2)*MW_HM(h,2,3) - MW_HM(h,1,3)*MW_HM(h,2,2);
    MW_HM(rc,3,2) = MW_HM(h,1,3)*MW_HM(h,2,1) - MW_HM(h,1,1)*MW_HM(h,2,3);
    MW_HM(rc,3,3) = MW_HM(h,1,1)*MW_HM(h,2,2) - MW_HM(h,1,2)*MW_HM(h,2,1);
    const double vol = MW_HM(h,1,1)*MW_HM(rc,1,1) + MW_HM(h,1,2)*MW_HM(rc,1,2) + MW_HM(h,1,3)*MW_HM(rc,1,3);
    const double f = 2.0 * 3.141592653589793238462643383279502884197 / vol;
#pragma unroll
    for (int i = 0; i < 9; ++i) rc[i] *= f;
}

// -------------------------------------------------------------------------------------
// Volume move of one walker by its wavefront: mc_volume (mc_moves.F90:1216-1534; ref_ljr,
// which only chain synchronisation reads, is not carried).  Rare (probability ~1/N per move), so it is an
// out-of-line function: one symmetric hmatrix element of both lattices changes, every position is rescaled
// through fractional coordinates (lanes over molecules), image vectors are rebuilt on the device in the
// reference's order and arithmetic, and the full-box energies are recomputed by the wavefront WITH THE
// EXISTING LISTS (atom_energy over the slot-major list); on rejection everything is put back the way the
// reference does it (positions mapped back through the NEW reciprocal matrix, :1413-1506).
// -------------------------------------------------------------------------------------
struct VolCtx {
    double* pos_g;            // global positions of the walker's first box
    double* spos;             // LDS positions [L][N][3] or nullptr
    double* shmat;            // LDS hmatrix   [2][9]
    double* srecip;           // LDS recip     [2][9]
    double* svol;             // LDS volume    [2]
    double* siv;              // LDS image vectors [L][ivcap][3]
    int* sniv;                // LDS nivect    [2]
    double* hmat_g;           // global mirrors of the above, walker's first box
    double* vol_g;
    double* ivect_g;
    int* nivect_g;
    const uint32_t* list_g;   // slot-major list (columns in k_list_order's order), walker's first box
    const int* order_g;       // molecule of each column
    const int* nns_g;         // row length of each column
    const int* cmax_g;        // longest row per group of 64 columns
    uint32_t* queue;          // this lane's column of an LDS queue [kQCap][64]
    int N, S, ivcap, L;
};

// compute_ivects (molint.F90:174-217) for one lattice, lanes over vectors; returns nivect or -1
__device__ __forceinline__ int dev_compute_ivects(const double* __restrict__ h, double* __restrict__ siv_l,
                                                  double* __restrict__ iv_g, int ivcap, int lane)
{
#pragma clang fp contract(off)
    const double rc = kSmallA * kSigma;
    const int im = (int)floor(rc / sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2])) + 1;       // :189-191
    const int jm = (int)floor(rc / sqrt(h[3] * h[3] + h[4] * h[4] + h[5] * h[5])) + 1;
    const int km = (int)floor(rc / sqrt(h[6] * h[6] + h[7] * h[7] + h[8] * h[8])) + 1;
    const int w1 = 2 * jm + 1, w2 = 2 * km + 1;
    const int n = (2 * im + 1) * w1 * w2;                                                    // :193
    if (n > ivcap) return -1;
    const int central = (im * w1 + jm) * w2 + km;
    for (int k = lane; k < n; k += 64) {
        double vx = 0.0, vy = 0.0, vz = 0.0;                                                 // :197 central cell first
        if (k > 0) {
            const int lin = (k - 1 < central) ? k - 1 : k;                                   // loop order of :200-213
            const int kc = lin % w2 - km, jc = (lin / w2) % w1 - jm, ic = lin / (w2 * w1) - im;
            const double sx0 = (double)ic * h[0], sx1 = (double)ic * h[1], sx2 = (double)ic * h[2];
            const double sy0 = (double)jc * h[3], sy1 = (double)jc * h[4], sy2 = (double)jc * h[5];
            const double sz0 = (double)kc * h[6], sz1 = (double)kc * h[7], sz2 = (double)kc * h[8];
            vx = (sx0 + sy0) + sz0; vy = (sx1 + sy1) + sz1; vz = (sx2 + sy2) + sz2;          // :208
        }
        siv_l[3 * k] = vx; siv_l[3 * k + 1] = vy; siv_l[3 * k + 2] = vz;
        iv_g[3 * k] = vx; iv_g[3 * k + 1] = vy; iv_g[3 * k + 2] = vz;
    }
    return n;
}

__device__ __forceinline__ double dev_det3(const double* m)                                   // util.f90:16-41
{
    double det = MW_HM(m,1,1) * (MW_HM(m,2,2) * MW_HM(m,3,3) - MW_HM(m,2,3) * MW_HM(m,3,2));
    det = det - MW_HM(m,1,2) * (MW_HM(m,2,1) * MW_HM(m,3,3) - MW_HM(m,2,3) * MW_HM(m,3,1));
    det = det + MW_HM(m,1,3) * (MW_HM(m,2,1) * MW_HM(m,3,2) - MW_HM(m,2,2) * MW_HM(m,3,1));
    return det;
}

// ljr += (H_new * (recip . ljr / 2 pi) - ljr), lanes over molecules (mc_moves.F90:1288-1316)
__device__ __forceinline__ void dev_rescale(const VolCtx& c, int l, const double* recip, const double* hnew, int lane)
{
    const double invPi = 1.0 / 3.141592653589793238462643383279502884197;
    double* Pg = c.pos_g + (size_t)l * c.N * 3;
    double* Ps = c.spos ? c.spos + (size_t)l * c.N * 3 : nullptr;
    for (int i = lane; i < c.N; i += 64) {
        const double* p = Ps ? Ps + 3 * i : Pg + 3 * i;
        const double o0 = p[0], o1 = p[1], o2 = p[2];
        double s0 = MW_HM(recip,1,1) * o0 + MW_HM(recip,2,1) * o1 + MW_HM(recip,3,1) * o2;
        double s1 = MW_HM(recip,1,2) * o0 + MW_HM(recip,2,2) * o1 + MW_HM(recip,3,2) * o2;
        double s2 = MW_HM(recip,1,3) * o0 + MW_HM(recip,2,3) * o1 + MW_HM(recip,3,3) * o2;
        s0 = s0 * 0.5 * invPi; s1 = s1 * 0.5 * invPi; s2 = s2 * 0.5 * invPi;
        double t0 = MW_HM(hnew,1,1) * s0 + MW_HM(hnew,1,2) * s1 + MW_HM(hnew,1,3) * s2;
        double t1 = MW_HM(hnew,2,1) * s0 + MW_HM(hnew,2,2) * s1 + MW_HM(hnew,2,3) * s2;
        double t2 = MW_HM(hnew,3,1) * s0 + MW_HM(hnew,3,2) * s1 + MW_HM(hnew,3,3) * s2;
        t0 = t0 - o0; t1 = t1 - o1; t2 = t2 - o2;
        const double n0 = o0 + t0, n1 = o1 + t1, n2 = o2 + t2;
        Pg[3 * i] = n0; Pg[3 * i + 1] = n1; Pg[3 * i + 2] = n2;
        if (Ps) { Ps[3 * i] = n0; Ps[3 * i + 1] = n1; Ps[3 * i + 2] = n2; }
    }
}

// compute_model_energy of lattice l by one wavefront (value in every lane)
__device__ __forceinline__ double dev_wave_model_energy(const VolCtx& c, int l, int lane)
{
    const double* Pg = c.pos_g + (size_t)l * c.N * 3;
    const double* Ps = c.spos ? c.spos + (size_t)l * c.N * 3 : nullptr;
    const double* IVl = c.siv + (size_t)l * c.ivcap * 3;
    const uint32_t* Lg = c.list_g + (size_t)l * c.S * c.N;
    const int* ORD = c.order_g + (size_t)l * c.N;
    const int* NNS = c.nns_g + (size_t)l * c.N;
    const int* CM = c.cmax_g + (size_t)l * ((c.N + 63) >> 6);
    auto getiv = [&](int k, double& x, double& y, double& z) { x = IVl[3 * k]; y = IVl[3 * k + 1]; z = IVl[3 * k + 2]; };
    auto getpos = [&](int j, double& x, double& y, double& z) {
        const double* p = Ps ? Ps + 3 * (size_t)j : Pg + 3 * (size_t)j;
        x = p[0]; y = p[1]; z = p[2];
    };
    double esum = 0.0;
    const ListRsrc rs = list_rsrc(Lg, c.N, c.S);
    uint32_t cur[8];
    int n_cur = 0, mol = 0;
    uint32_t col = kNoColumn;
    if (lane < c.N) { col = (uint32_t)lane * 4u; n_cur = NNS[lane]; mol = ORD[lane]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) cur[u] = list_load(rs, col, u, c.N, c.S);
    for (int base = 0; base < c.N; base += 64) {                 // wave-uniform: one group of 64 list columns per pass
        const bool act = col != kNoColumn;
        const int tn = base + 64 + lane;
        uint32_t col_next = kNoColumn;
        int n_next = 0, mol_next = 0;
        if (tn < c.N) { col_next = (uint32_t)tn * 4u; n_next = NNS[tn]; mol_next = ORD[tn]; }
        const int cm = __builtin_amdgcn_readfirstlane(CM[base >> 6]);
        AtomSum a = atom_energy<64, false>(rs, col, col_next, mol, act ? (n_cur & 0xff) : 0, cm & 0xff, cm >> 8, c.N, c.S, c.queue, getpos, getiv, cur);
        if (act) esum += a.e;
        n_cur = n_next; mol = mol_next; col = col_next;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off, 64);
    return esum;
}

__device__ __forceinline__
int volume_move_wave(const VolCtx& c, const SweepParams& sp, const double* weight, const double* __restrict__ mu_bin,
                     const double* __restrict__ binwidth, double u0, double u1, double u2, double u3,
                     int& ls, double& ls_mu, double men[2], int lane)
{
    const int L = c.L, N = c.N;
    double backup_e[2] = {men[0], men[1]}, old_vol[2] = {c.svol[0], c.svol[1]};
    double old_h[2][9], recip_used[2][9];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int t = 0; t < 9; ++t) { old_h[l][t] = c.shmat[l * 9 + t]; recip_used[l][t] = c.srecip[l * 9 + t]; }
    __builtin_amdgcn_wave_barrier();
    const int idim = (int)(u0 * 3.0) + 1, jdim = (int)(u1 * 3.0) + 1;                          // :1269-1272
    const double dh = (2.0 * u2 - 1.0) * sp.dv_max;                                             // :1276
    if (lane == 0) {
        for (int l = 0; l < L; ++l) {                                                           // :1281-1282
            MW_HM(c.shmat + 9 * l, idim, jdim) = MW_HM(c.shmat + 9 * l, idim, jdim) + dh;
            if (idim != jdim) MW_HM(c.shmat + 9 * l, jdim, idim) = MW_HM(c.shmat + 9 * l, jdim, idim) + dh;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double new_e[2] = {0.0, 0.0};
    int bad = 0, nresc = 0;                                                                     // nresc: lattices rescaled so far
    for (int l = 0; l < L; ++l) {                                                               // :1285-1358
        dev_rescale(c, l, recip_used[l], c.shmat + 9 * l, lane);
        nresc = l + 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int niv = dev_compute_ivects(c.shmat + 9 * l, c.siv + (size_t)l * c.ivcap * 3,
                                           c.ivect_g + (size_t)l * c.ivcap * 3, c.ivcap, lane);
        if (lane == 0) {
            c.svol[l] = fabs(dev_det3(c.shmat + 9 * l));
            double rcp[9];
            dev_recipmatrix(c.shmat + 9 * l, rcp);
#pragma unroll
            for (int t = 0; t < 9; ++t) c.srecip[l * 9 + t] = rcp[t];
            if (niv >= 0) { c.sniv[l] = niv; c.nivect_g[l] = niv; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // The new cell needs more image vectors than there is room for: the move counts as rejected (and is
        // flagged); the lattices rescaled so far -- this one included, its reciprocal matrix is in place -- go back.
        if (niv < 0) { bad = 1; break; }
        new_e[l] = dev_wave_model_energy(c, l, lane);
    }
    int ok = 0;
    if (!bad) {
        men[0] = new_e[0]; men[1] = new_e[1];
        const double dE = (ls == 1 ? new_e[0] - backup_e[0] : new_e[1] - backup_e[1]);          // :1361
        const double Vls = ls == 1 ? c.svol[0] : c.svol[1], Vold = ls == 1 ? old_vol[0] : old_vol[1];
        double old_eta = 0.0, new_eta = 0.0;
        if (L == 2) {                                                                            // :1363-1371
            old_eta = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);
            double mu = (men[0] + sp.pressure * c.svol[0]) - (men[1] + sp.pressure * c.svol[1]);
            mu = mu - sp.dref;                                                                   // :1371 (leshift)
            mu = mu * sp.beta - (double)N * log(c.svol[0] / c.svol[1]);
            ls_mu = mu;
            new_eta = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);
        }
        double diffkT = sp.beta * dE + new_eta - old_eta + sp.beta * sp.pressure * (Vls - Vold)
                        - (double)N * log(Vls / Vold);                                           // :1381-1382
        int minu_ls = ls;
        if (sp.minu && L == 2)                                                                   // :1385-1401
            minu_ls = dev_minu_branch(sp, ls, men[0], men[1], c.svol[0], c.svol[1], ls == 1 ? backup_e[0] : backup_e[1], Vold, true, N,
                                      new_eta, old_eta, diffkT);
        double cmp = exp(-diffkT);
        cmp = cmp > 1.0 ? 1.0 : cmp;
        ok = u3 < cmp ? 1 : 0;                                                                   // :1410
        if (ok) ls = minu_ls;                                                                    // :1426-1429
    }
    if (!ok) {                                                                                   // :1426-1530
        double recip_new[2][9];
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int t = 0; t < 9; ++t) recip_new[l][t] = c.srecip[l * 9 + t];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            for (int l = 0; l < L; ++l) {
                c.svol[l] = old_vol[l];
                for (int t = 0; t < 9; ++t) { c.shmat[l * 9 + t] = old_h[l][t]; c.srecip[l * 9 + t] = recip_used[l][t]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int l = 0; l < (bad ? nresc : L); ++l) {
            dev_rescale(c, l, recip_new[l], c.shmat + 9 * l, lane);                              // back through the NEW recip
            const int niv = dev_compute_ivects(c.shmat + 9 * l, c.siv + (size_t)l * c.ivcap * 3,
                                               c.ivect_g + (size_t)l * c.ivcap * 3, c.ivcap, lane);   // :1510-1512
            if (lane == 0 && niv > 0) { c.sniv[l] = niv; c.nivect_g[l] = niv; }
        }
        men[0] = backup_e[0]; men[1] = backup_e[1];                                              // :1514
        if (L == 2) {                                                                            // :1516-1520
            double mu = (men[0] + sp.pressure * c.svol[0]) - (men[1] + sp.pressure * c.svol[1]);
            mu = mu - sp.dref;                                                                   // :1526 (leshift)
            mu = mu * sp.beta - (double)N * log(c.svol[0] / c.svol[1]);
            ls_mu = mu;
        }
    }
    if (lane == 0) {                                   // global mirrors of the cell
        for (int l = 0; l < L; ++l) {
            for (int t = 0; t < 9; ++t) c.hmat_g[l * 9 + t] = c.shmat[l * 9 + t];
            c.vol_g[l] = c.svol[l];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return bad ? -1 : ok;
}

// Per-walker tables (two lattices only): weight / histogram / unbiased_hist [walker][nbins]; every walker
// reads its OWN weights in eta_weight, so Wang-Landau updates stay local until the host synchronises them
// (comms_allreduce_eta/hist/uhist semantics, WalkerComms).
// Two wavefronts per SIMD are the design point (one walker per wavefront, 512 VGPRs per SIMD lane): the build with volume
// moves must stay within 256 VGPRs -- at 264 it ran ONE wavefront per SIMD and the NPT farm lost a third of its rate.
template <bool LDSPOS, bool LDSLIST, bool WITHVOL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_sweep_translation(double* pos, double* hmat, double* ivect,
                         int* nivect, const uint32_t* __restrict__ listm, const uint32_t* __restrict__ list,
                         const int* __restrict__ nn, const int* __restrict__ order, const int* __restrict__ nns,
                         const int* __restrict__ cmax, double* __restrict__ energy,
                         int* __restrict__ wls, double* __restrict__ wmu, unsigned long long* __restrict__ wacc,
                         unsigned long long* __restrict__ wswitch, double* __restrict__ wshift,
                         SweepParams sp, double* wweight, double* whist, double* wuhist,
                         const double* __restrict__ mu_bin_g, const double* __restrict__ binwidth_g,
                         double* volume, unsigned long long* __restrict__ wvol, int* __restrict__ wflag,
                         int N, int S, int ivcap, int nmoves, unsigned long long seed, unsigned long long move0,
                         int walker0, double* __restrict__ mvlog, int rstride,
                         const double* __restrict__ wwin, double* __restrict__ wfac, double* __restrict__ wsum,
                         int* __restrict__ winflag, const double* __restrict__ wstep)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ WaveScratch ws;
    __shared__ uint32_t squeue[WITHVOL ? (kQCap + 1) * 64 : 1];   // in-range queue of the volume move's full-box energy
    __shared__ double shmat[2][9], svol[2];      // the walker's cells: volume moves change them in place
    __shared__ int sniv[2];
    const int lane = threadIdx.x;
    const int wlk = walker0 + blockIdx.x;
    const int L = sp.nlat;
    const int box0 = wlk * L;
    // per-walker pieces of the parameter block: its window of the overlap parameter ('dd': mc_moves.F90:659-709; else
    // the whole range), whether it has reached that window, its Wang-Landau increment and Swetnam's visit total
    if (wwin) {
        sp.start_bin = (int)wwin[4 * (size_t)wlk]; sp.end_bin = (int)wwin[4 * (size_t)wlk + 1];
        sp.mu_lo = wwin[4 * (size_t)wlk + 2]; sp.mu_hi = wwin[4 * (size_t)wlk + 3];
    }
    sp.in_window = sp.dd ? winflag[wlk] : 1;                              // mc_moves.F90:112,872
    if (wstep) { sp.max_trans = wstep[2 * (size_t)wlk]; sp.dv_max = wstep[2 * (size_t)wlk + 1]; }   // per-walker step sizes (equilibration tuning, :1729-1732)
    double wlf = L == 2 ? wfac[wlk] : 0.0;                                // wl_factor of this walker (:1615,1677)
    double sumh = L == 2 ? wsum[wlk] : 0.0;                               // sumhist (:94,1638)
    const double invPi = 1.0 / 3.141592653589793238462643383279502884197;

    // image vectors of the walker's lattices in LDS: siv[l][ivcap][3]; with LDSPOS (small systems) the walker's
    // positions live there too for the whole launch -- spos[l][N][3] -- and every gather is an LDS read
    double* siv = smem;
    double* spos = smem + (size_t)L * ivcap * 3;
    for (int l = 0; l < L; ++l) {
        const int niv = nivect[box0 + l];
        for (int t = lane; t < niv * 3; t += 64) siv[(size_t)l * ivcap * 3 + t] = ivect[(size_t)(box0 + l) * ivcap * 3 + t];
        if (LDSPOS) {
            const double* Pg = pos + (size_t)(box0 + l) * N * 3;
            for (int t = lane; t < 3 * N; t += 64) spos[(size_t)l * N * 3 + t] = Pg[t];
        }
    }
    // LDSLIST (the reference's own system sizes, ~48 molecules): list rows (`rstride` entries each: the longest row of
    // any box, rounded up to 4, at most 32) and row lengths too, so that nothing in the move loop waits on global
    // memory.  (A read of slots rstride..31 of a row lands in the next row or the row lengths: masked by the caller.)
    uint32_t* srow = reinterpret_cast<uint32_t*>(spos + (LDSPOS ? (size_t)L * N * 3 : 0));
    int* snn = reinterpret_cast<int*>(srow + (LDSLIST ? (size_t)L * N * rstride : 0));
    if (LDSLIST) {
        for (int l = 0; l < L; ++l) {
            const uint32_t* LMg = listm + (size_t)(box0 + l) * N * kRow;
            for (int t = lane; t < N * rstride; t += 64) srow[(size_t)l * N * rstride + t] = LMg[(size_t)(t / rstride) * kRow + (t % rstride)];
            for (int t = lane; t < N; t += 64) snn[l * N + t] = nn[(size_t)(box0 + l) * N + t];
        }
    }
    // two lattices: this walker's weight table and the (shared) bin centres / widths, so that eta_weight and the
    // Wang-Landau update after every move are LDS arithmetic; the weights go back to the walker's table at the end
    double* sweight = reinterpret_cast<double*>(snn + (LDSLIST ? (size_t)L * N : 0));
    double* smub = sweight + sp.nbins;
    double* sbw = smub + sp.nbins;
    if (L == 2) {
        for (int t = lane; t < sp.nbins; t += 64) {
            sweight[t] = wweight[(size_t)wlk * sp.nbins + t];
            smub[t] = mu_bin_g[t];
            sbw[t] = binwidth_g[t];
        }
    }
    const double* mu_bin = smub;
    const double* binwidth = sbw;
    __shared__ double srecip[2][9];          // recip_matrix(:,:,ils) of the walker's lattices
    if (lane == 0) {
        for (int l = 0; l < L; ++l) {
            double rcp[9];
            dev_recipmatrix(hmat + (size_t)(box0 + l) * 9, rcp);
#pragma unroll
            for (int t = 0; t < 9; ++t) { srecip[l][t] = rcp[t]; shmat[l][t] = hmat[(size_t)(box0 + l) * 9 + t]; }
            svol[l] = volume[box0 + l];
            sniv[l] = nivect[box0 + l];
        }
    }
    __syncthreads();
    VolCtx vc;
    vc.pos_g = pos + (size_t)box0 * N * 3; vc.spos = LDSPOS ? spos : nullptr;
    vc.shmat = &shmat[0][0]; vc.srecip = &srecip[0][0]; vc.svol = svol; vc.siv = siv; vc.sniv = sniv;
    vc.hmat_g = hmat + (size_t)box0 * 9; vc.vol_g = volume + box0; vc.ivect_g = ivect + (size_t)box0 * ivcap * 3;
    vc.nivect_g = nivect + box0; vc.list_g = list + (size_t)box0 * S * N;
    vc.order_g = order + (size_t)box0 * N; vc.nns_g = nns + (size_t)box0 * N; vc.cmax_g = cmax + (size_t)box0 * ((N + 63) >> 6);
    vc.queue = squeue + lane; vc.N = N; vc.S = S; vc.ivcap = ivcap; vc.L = L;
    unsigned long long nvol_try = 0, nvol_acc = 0;
    int flag = 0;

    // this walker's weight table (read by eta_weight, updated by mc_update_wl_bins) and histograms
    double* weight = sweight;                // (LDS copy; only two-lattice runs read or update it)
    double* hist = whist + (size_t)wlk * sp.nbins;
    double* uhist = wuhist + (size_t)wlk * sp.nbins;
    unsigned long long nsw = 0;
    double gauge = 0.0;                      // total of the minima subtracted from this walker's weights (:1682-1685)

    int ls = wls[wlk];                       // active lattice, 1-based
    double ls_mu = wmu[wlk];
    double men[2] = {energy[box0], L == 2 ? energy[box0 + 1] : 0.0};
    unsigned long long acc = 0;

    for (int mv = 0; mv < nmoves; ++mv) {
        // six uniforms: lanes 0..2 run one Philox call each, the values are broadcast
        double ua = 0.0, ub = 0.0;
        if (lane < 4) {
            const unsigned long long m = move0 + (unsigned long long)mv;
            uint32_t c[4] = {(uint32_t)m, (uint32_t)(m >> 32), (uint32_t)wlk, (uint32_t)lane};
            philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            ua = u53(c[0], c[1]); ub = u53(c[2], c[3]);
        }
        const double u0 = readlane_f64(ua, 0), u1 = readlane_f64(ub, 0), u2 = readlane_f64(ua, 1);
        const double u3 = readlane_f64(ub, 1), u4 = readlane_f64(ua, 2), u5 = readlane_f64(ub, 2);
        const double u6 = readlane_f64(ua, 3);          // lattice-switch variate (mc_moves.F90:1576)
        const double u7 = readlane_f64(ub, 3);          // move type (mc_moves.F90:226)
        int cyc = 0;                                                   // mc_cycle_num of this move ('dd' only)
        if (sp.dd) {
            const unsigned long long mg = move0 + (unsigned long long)mv;
            cyc = (int)(mg / (unsigned long long)N) + 1;
            if (mg % (unsigned long long)N == 0ull) {                  // top of a cycle: the equilibration check of mc_cycle (:181-210)
                if (cyc < sp.eq_cycles) sp.in_window = (ls_mu > sp.mu_lo && ls_mu < sp.mu_hi) ? 1 : 0;
                else if (cyc == sp.eq_cycles) { if (!sp.in_window) flag |= 2; }     // "Not all walkers have reached their designated window"
                else sp.in_window = 1;                                 // a restart
            }
        }
        const bool is_volume = WITHVOL && !(u7 < sp.transP);    // WITHVOL = false: translation-only build, no call, lean registers
        bool ok = false;
        double eo[2] = {0.0, 0.0}, en[2] = {0.0, 0.0}, diffkT = 0.0;
        int imol = 0;
        if (is_volume) {                                                          // mc_moves.F90:232-235
            int rv = 0;
            if constexpr (WITHVOL) rv = volume_move_wave(vc, sp, weight, mu_bin, binwidth, u0, u1, u2, u3, ls, ls_mu, men, lane);
            ++nvol_try;
            if (rv == 1) ++nvol_acc;
            if (rv < 0) flag |= 1;
            ok = rv == 1;
        } else {
        const int lsn = L == 2 ? 3 - ls : 1;
        imol = (int)(u0 * (double)N) + 1;                                        // mc_moves.F90:1001-1002
        imol = imol > N ? N : imol;
        const int i = imol - 1;
        double x = 2.0 * u1 - 1.0, y = 2.0 * u2 - 1.0, z = 2.0 * u3 - 1.0;        // :1021-1027
        const double norm = 1.0 / sqrt(x * x + y * y + z * z);                    // :1029
        x *= norm; y *= norm; z *= norm;
        const double r = u4 * 2.0 - 1.0;                                          // :1035
        x = x * sp.max_trans * r; y = y * sp.max_trans * r; z = z * sp.max_trans * r;
        const double* rc = srecip[ls - 1];
        double sx = MW_HM(rc,1,1) * x + MW_HM(rc,2,1) * y + MW_HM(rc,3,1) * z;    // :1042-1050
        double sy = MW_HM(rc,1,2) * x + MW_HM(rc,2,2) * y + MW_HM(rc,3,2) * z;
        double sz = MW_HM(rc,1,3) * x + MW_HM(rc,2,3) * y + MW_HM(rc,3,3) * z;
        sx = sx * 0.5 * invPi; sy = sy * 0.5 * invPi; sz = sz * 0.5 * invPi;      // :1052-1054
        double tv[2][3] = {{x, y, z}, {x, y, z}};                                  // move in the active lattice
        if (L == 2) {                                                             // :1061-1067
            const double* hn = shmat[lsn - 1];
            const double mx = MW_HM(hn,1,1) * sx + MW_HM(hn,1,2) * sy + MW_HM(hn,1,3) * sz;
            const double my = MW_HM(hn,2,1) * sx + MW_HM(hn,2,2) * sy + MW_HM(hn,2,3) * sz;
            const double mz = MW_HM(hn,3,1) * sx + MW_HM(hn,3,2) * sy + MW_HM(hn,3,3) * sz;
            if (lsn == 1) { tv[0][0] = mx; tv[0][1] = my; tv[0][2] = mz; }         // static indices only
            else          { tv[1][0] = mx; tv[1][1] = my; tv[1][2] = mz; }
        }

        double pn[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
#pragma unroll
        for (int l = 0; l < 2; ++l) if (l < L) {                                  // :1007-1018, 1076-1092
            const double* P = pos + (size_t)(box0 + l) * N * 3;
            const uint32_t* LM = listm + (size_t)(box0 + l) * N * kRow;
            const int* NN = nn + (size_t)(box0 + l) * N;
            const double* IVl = siv + (size_t)l * ivcap * 3;
            auto getiv = [&](int k, double& a, double& b, double& c) { a = IVl[3 * k]; b = IVl[3 * k + 1]; c = IVl[3 * k + 2]; };
            const double* Pl = LDSPOS ? (spos + (size_t)l * N * 3) : P;
            auto getpos = [&](int j, double& a, double& b, double& c) { const double* p = Pl + 3 * (size_t)j; a = p[0]; b = p[1]; c = p[2]; };
            double xo, yo, zo;
            getpos(i, xo, yo, zo);
            pn[l][0] = xo + tv[l][0]; pn[l][1] = yo + tv[l][1]; pn[l][2] = zo + tv[l][2];   // :1079
            const uint32_t* SR = srow + (size_t)l * N * rstride;
            const int* SN = snn + l * N;
            auto row = [&](int jx, int sl) { return LDSLIST ? SR[jx * rstride + sl] : LM[(size_t)jx * kRow + sl]; };
            auto nnof = [&](int jx) { return LDSLIST ? SN[jx] : NN[jx]; };
            MoveRes res;
            const bool fast = move_energy_wave(getpos, getiv, row, nnof, &ws, sniv[l], i, nnof(i), row(i, lane & 31), xo, yo, zo,
                                               pn[l][0], pn[l][1], pn[l][2], lane, res);
            if (!fast) {
                Override none; none.idx = -1; none.x = none.y = none.z = 0.0;
                Override tr; tr.idx = i; tr.x = pn[l][0]; tr.y = pn[l][1]; tr.z = pn[l][2];
                res.eo = local_energy_wave(P, ivect + (size_t)(box0 + l) * ivcap * 3, LM, NN, i, none, none, lane, res.io, res.so);
                res.en = local_energy_wave(P, ivect + (size_t)(box0 + l) * ivcap * 3, LM, NN, i, tr, none, lane, res.in_, res.sn);
            }
            eo[l] = res.eo; en[l] = res.en;
        }
        const double dE0 = en[0] - eo[0], dE1 = en[1] - eo[1];                    // :1090
        const double bk0 = men[0], bk1 = men[1];                                  // :1013
        men[0] = (men[0] - eo[0]) + en[0];                                        // :1016,1087
        men[1] = (men[1] - eo[1]) + en[1];
        int minu_ls = ls;
        if (L == 1) {
            diffkT = sp.beta * dE0;                                               // :1106
        } else {
            const double eta_old = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);   // :1112-1116
            ls_mu = ls_mu + (dE0 - dE1) * sp.beta;
            const double eta_new = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);
            diffkT = (ls == 1 ? dE0 : dE1) * sp.beta + eta_new - eta_old;
            if (sp.minu)                                                          // :1119-1140
                minu_ls = dev_minu_branch(sp, ls, men[0], men[1], svol[0], svol[1], ls == 1 ? bk0 : bk1, ls == 1 ? svol[0] : svol[1],
                                          sp.npt != 0, N, eta_new, eta_old, diffkT);
        }
        double pacc = exp(-diffkT);
        pacc = pacc > 1.0 ? 1.0 : pacc;
        ok = u5 < pacc;                                                           // :1145-1146 (false for NaN)
        if (ok) {
            ++acc;
            ls = minu_ls;                                                         // :1168-1170
            if (lane == 0) {
#pragma unroll
                for (int l = 0; l < 2; ++l) if (l < L) {
                    double* P = pos + ((size_t)(box0 + l) * N + i) * 3;
                    P[0] = pn[l][0]; P[1] = pn[l][1]; P[2] = pn[l][2];
                    if (LDSPOS) {
                        double* S = spos + ((size_t)l * N + i) * 3;
                        S[0] = pn[l][0]; S[1] = pn[l][1]; S[2] = pn[l][2];
                    }
                }
            }
        } else {                                                                  // :1182-1195
            men[0] = bk0; men[1] = bk1;
            if (L == 2) ls_mu = ls_mu - (dE0 - dE1) * sp.beta;
        }
        }   // translation
        // the next move of this wavefront must see the committed position (and the weights written below)
        int sw = 0;
        if (L == 2 && sp.record) {                                                // mc_update_wl_bins, :1597-1689
            const int k = dev_mu_to_bin(sp, ls_mu);
            if (k >= 1 && k <= sp.nbins) {
                const double bwk = binwidth[k - 1];
                if (sp.samplerun) {
                    const double etaw = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);
                    if (lane == 0) {
                        hist[k - 1] = hist[k - 1] + sp.av_binwidth / bwk;                        // :1621
                        uhist[k - 1] = uhist[k - 1] + (sp.av_binwidth / bwk) * exp(etaw - sp.log_unbiased_norm);   // :1627-1629
                    }
                } else {
                    if (sp.swetnam) {                                             // :1636-1653
                        sumh = sumh + 1.0;
                        double acc = 0.0;
                        const double span = sp.mu_max - sp.mu_min - 1.0;
                        for (int b = lane; b < sp.nbins; b += 64) {
                            const double hb = hist[b] + (b == k - 1 ? sp.av_binwidth / bwk : 0.0);    // this move's visit is already counted (:1621)
                            const double dev = hb * binwidth[b] / sumh - binwidth[b] / span;
                            acc += dev * dev;
                        }
                        acc = readlane_f64(dpp_wave_sum(acc), 63);
                        double f = sqrt(acc / (double)sp.nbins);
                        f = log(f) * sp.wl_alpha * (double)sp.nbins;
                        wlf = f < sp.orig_wl_factor ? f : sp.orig_wl_factor;
                    }
                    // weight(k) += av_binwidth*wl_factor/binwidth(k) -- whichever bin k is (:1680); then the minimum over the
                    // walker's window is subtracted inside the window (:1682-1685; with 'dd' windows k may lie outside)
                    const double inc = sp.av_binwidth * wlf / bwk;
                    double mn = 1.7976931348623157e308;
                    for (int b = sp.start_bin - 1 + lane; b < sp.end_bin; b += 64) {
                        double w = weight[b];
                        if (b == k - 1) w = w + inc;
                        mn = w < mn ? w : mn;
                    }
                    mn = readlane_f64(dpp_wave_min(mn), 63);
                    for (int b = sp.start_bin - 1 + lane; b < sp.end_bin; b += 64) {
                        double w = weight[b];
                        if (b == k - 1) w = w + inc;
                        weight[b] = w - mn;
                    }
                    if (lane == 0 && (k < sp.start_bin || k > sp.end_bin)) weight[k - 1] = weight[k - 1] + inc;
                    gauge += mn;
                    if (lane == 0) hist[k - 1] = hist[k - 1] + sp.av_binwidth / bwk;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
        if (L == 2 && sp.always_switch && !(sp.dd && cyc < sp.eq_cycles)) {      // mc_lattice_switch, :1536-1594 (:243-248: not while a 'dd' run equilibrates)
            const int lsw = 3 - ls;
            const double eta_w = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);
            const double deta = eta_w - eta_w;                                    // new_eta - old_eta, :1557-1558
            const double Els = ls == 1 ? men[0] : men[1], Elsn = ls == 1 ? men[1] : men[0];
            const double V1 = svol[0], V2 = svol[1];
            const double Vls = ls == 1 ? V1 : V2, Vlsn = ls == 1 ? V2 : V1;
            double dk;
            if (sp.npt) dk = sp.beta * Elsn - sp.beta * Els + sp.beta * sp.pressure * (Vlsn - Vls) - (double)N * log(Vlsn / Vls) + deta;
            else        dk = sp.beta * Elsn - sp.beta * Els + deta;
            dk = dk + (ls == 1 ? sp.beta * sp.dref : -(sp.beta * sp.dref));       // leshift: - beta ref(lsn) + beta ref(ls), :1567,1572
            double cmp = exp(-dk);
            cmp = cmp > 1.0 ? 1.0 : cmp;
            if (u6 < cmp) {
                double mu = (men[0] + sp.pressure * V1) - (men[1] + sp.pressure * V2);          // :1581-1583
                mu = mu - sp.dref;                                                              // :1584 (leshift)
                mu = mu * sp.beta - (double)N * log(V1 / V2);
                ls_mu = mu; ls = lsw; sw = 1; ++nsw;
            }
        }
        if (mvlog && lane == 0) {
            double* q = mvlog + ((size_t)blockIdx.x * nmoves + mv) * 8;
            if (is_volume) { eo[0] = men[0]; en[0] = svol[0]; eo[1] = L == 2 ? men[1] : 0.0; en[1] = L == 2 ? svol[1] : 0.0; }
            q[0] = (double)imol; q[1] = (ok ? 1.0 : 0.0) + 2.0 * sw + (is_volume ? 4.0 : 0.0); q[2] = eo[0]; q[3] = en[0]; q[4] = eo[1]; q[5] = en[1]; q[6] = ls_mu; q[7] = diffkT;
        }
        // the next move of this wavefront must see the committed position
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (L == 2 && sp.record && !sp.samplerun)
        for (int t = lane; t < sp.nbins; t += 64) wweight[(size_t)wlk * sp.nbins + t] = sweight[t];
    if (lane == 0) {
        wls[wlk] = ls; wmu[wlk] = ls_mu; wacc[wlk] += acc; wswitch[wlk] += nsw; wshift[wlk] += gauge;
        wvol[2 * wlk] += nvol_try; wvol[2 * wlk + 1] += nvol_acc;
        if (flag) wflag[wlk] |= flag;                       // bit 0: image-vector table outgrown, bit 1: 'dd' walker not in its window at eq_mc_cycles
        if (L == 2) { wfac[wlk] = wlf; wsum[wlk] = sumh; }
        if (sp.dd) winflag[wlk] = sp.in_window;
        energy[box0] = men[0];
        if (L == 2) energy[box0 + 1] = men[1];
    }
}

// =====================================================================================
// The exchange step's host-side sums, on the device: a farm holds one table per walker, and the synchronisation
// needs  sum over walkers of (table + gauge shift - last synchronised table)  per bin -- 3 x nbins numbers -- and
// afterwards the same synchronised row in every walker.  Fixed summation order (chunks of walkers, then the
// chunks in order), so the result does not depend on scheduling.
//   k_tables_partial: grid = chunks of kTableChunk walkers, thread = bin;  k_tables_final: thread = bin
// =====================================================================================
constexpr int kTableChunk = 64;

__global__ __launch_bounds__(128)
void k_tables_partial(const double* __restrict__ tab, const double* __restrict__ shift, const double* __restrict__ last,
                      double* __restrict__ partial, int nbins, int w0, int count)
{
    const int c = blockIdx.x;
    const int wa = w0 + c * kTableChunk, wb = min(w0 + count, wa + kTableChunk);
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) {
        const double l = last[b];
        double s = 0.0;
        for (int w = wa; w < wb; ++w) s += (tab[(size_t)w * nbins + b] + (shift ? shift[w] : 0.0)) - l;
        partial[(size_t)c * nbins + b] = s;
    }
}

__global__ __launch_bounds__(128)
void k_tables_final(const double* __restrict__ partial, double* __restrict__ out, int nbins, int nchunks)
{
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) {
        double s = 0.0;
        for (int c = 0; c < nchunks; ++c) s += partial[(size_t)c * nbins + b];
        out[b] = s;
    }
}

__global__ __launch_bounds__(128)
void k_tables_broadcast(double* __restrict__ tab, const double* __restrict__ row, int nbins, int w0)
{
    double* t = tab + (size_t)(w0 + blockIdx.x) * nbins;
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) t[b] = row[b];
}

}  // namespace mw
