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
// independent walkers at once.  One workgroup per walker -- one wavefront per lattice -- runs its
// Markov chain move after move: pick a molecule, draw the displacement in the active lattice, map
// it through fractional coordinates into the partner lattice (:1042-1066), fused old/new local
// energy in each lattice (move_energy_wave, the lattices side by side), update the order parameter
// mu and the multicanonical weights' contribution, accept or revert (:1145-1209).  The caller-side
// bookkeeping of model_energy (:1013-1016,1087,1190) is done here on the per-box energies.
// Random numbers: Philox4x32-10, counter (move lo, move hi, walker, call), key = seed -- the same
// stream as the oracle's mwo_move_uniforms.
//   grid = walkers in the launch, block = 64 x lattices
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

// -------------------------------------------------------------------------------------
// Order parameter -> bin -> weight, LANE-PARALLEL.  These are scalar computations of the host program (a log, three
// divisions, an exp per call) and a wavefront has no scalar double-precision unit: evaluated one after the other by all
// 64 lanes they cost more vector instructions per move than one lattice's whole energy evaluation (profiles/r03a: 2200
// VALU instructions per two-lattice move with the Wang-Landau update and a switch attempt, 1170 of them energy).  Here
// every value of mu a move needs -- the trial value, the value a rejection restores, the current one -- sits in its own
// lane and ONE instruction stream serves them all; likewise the move's exponentials.
// -------------------------------------------------------------------------------------
struct MuGridDev {                       // per walker, wave-uniform
    double c_pos, c_neg, ilr_pos, ilr_neg, mu_lo, mu_hi;       // c = (1 - r) / a, ilr = 1 / log(r) of the two geometric bin progressions
    int nbins, start_bin, end_bin, eta_interp, in_window;
};

// mc_moves.F90:2187-2215, one mu per lane: bin = nbins/2 + 2 + int(log(1 - (mu - 0.5)(1 - r)/a) / log r) on the positive side.  The
// two divisions are multiplications by per-walker constants and the logarithm is the engine's own (mw_common.hip.h) -- 45 vector
// instructions where the expression as written costs 220; the bin differs from the reference's only for a mu within ~1e-15
// (relative) of a bin boundary, where two libm implementations differ as well.
__device__ __forceinline__ int lane_mu_to_bin(const MuGridDev& g, double mu)
{
    const double a = fabs(mu);
    const bool pos = mu > 0.0;
    const double c = pos ? g.c_pos : g.c_neg, ilr = pos ? g.ilr_pos : g.ilr_neg;
    const double arg = __builtin_fma(-(a - 0.5), c, 1.0);
    const int q = (int)(fast_log_pos(arg) * ilr);
    return a <= 0.5 ? g.nbins / 2 + 1 : (pos ? g.nbins / 2 + 2 + q : g.nbins / 2 - q);
}

// eta_weight (mc_moves.F90:893-964) for one mu per lane, bin k already known; w / mb / bw: 0-based tables in LDS.
// The four interpolation branches of the reference are one expression with selected indices:
//   eta = w(base) + (mu - mu_bin(base)) * 2 (w(hi) - w(lo)) / (binwidth(hi) + binwidth(lo)),   hi = lo + 1
__device__ __forceinline__ double lane_eta(const MuGridDev& g, const double* w, const double* __restrict__ mb,
                                           const double* __restrict__ bw, double mu, int k)
{
    const int nb = g.nbins;
    const int kc = k < 1 ? 1 : (k > nb ? nb : k);               // (a bin outside the table only with mu outside the range: not used then)
    const bool up = (kc == g.start_bin) || (kc != g.end_bin && mu > mb[kc - 1]);
    int hi = up ? kc + 1 : kc;
    hi = hi > nb ? nb : (hi < 2 ? 2 : hi);
    const int lo = hi - 1;
    const int base = (up || kc == g.end_bin) ? kc : (kc > 1 ? kc - 1 : 1);
    double val = w[kc - 1];
    if (g.eta_interp) val = w[base - 1] + (mu - mb[base - 1]) * (2.0 * (w[hi - 1] - w[lo - 1]) / (bw[hi - 1] + bw[lo - 1]));
    // 'dd' walkers that have not reached their window yet carry no weight: the reference returns there without
    // assigning the function result (:913); 0 is what its comment asks for ("don't want to penalise walkers")
    val = (mu < g.mu_lo || mu > g.mu_hi) ? 1.7976931348623157e308 : val;      // huge(1.0_dp)
    return g.in_window ? val : 0.0;
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

// Diagnostic build only (-DMW_SWEEP_STAMPS, tools/sweep_stamps.py): cycles of walker 0's first wavefront per phase of a round,
// summed over the launch into g_sweep_stamps[0..15] (mw_move_energy.hip.h holds the array and the stages of one evaluation).
#ifdef MW_SWEEP_STAMPS
#define MW_SW_NOW() ((blockIdx.x == 0 && wv == 0) ? (unsigned long long)clock64() : 0ull)
#define MW_SW_ACC(k, d) do { if (blockIdx.x == 0 && wv == 0 && lane == 0) g_sweep_stamps[k] += (d); } while (0)
#else
#define MW_SW_NOW() 0ull
#define MW_SW_ACC(k, d) do { } while (0)
#endif

#ifndef MW_BIG_WHEN
#define MW_BIG_WHEN ((SPEC > 1) ? 1 : 0)      // when the moments of walkers in global memory are asked for (move_energy_mom_wave: WHEN)
#endif
constexpr int kSweepQCap = 9;                             // in-range queue of a volume move's full-box energy: sized to fit the scratch record
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
    double* sbk;              // LDS backup of a volume move's old cells [2][27]
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
    uint32_t* queue;          // this lane's column of an LDS queue [kSweepQCap + 1][64]
    const unsigned short* srow;   // LDS list rows [L][N][rstride] (16-bit entries) and row lengths [L][N] of walkers entirely in LDS, else nullptr
    const unsigned char* snn;
    int rstride;
    double* mom_trial;        // where a volume move leaves the moments of the trial cell, [L][N][kMomStride], or nullptr
    unsigned* inmask;         // SPLIT builds (volume_move_wg): per molecule the row slots in range, [L][N]
    double* rec;              // ... and the in-range neighbours' {dx, dy, dz, 1/r, e1, g}, [L][N][kSplitQ][6]
    int N, S, ivcap, L;
};
constexpr int kSplitQ = 12;   // in-range neighbours per molecule the split evaluation has room for (more: the one-wavefront routine)

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
// (LDSPOS at compile time: a pointer chosen at run time between LDS and global memory makes every access through it a FLAT one --
//  the full-box energy's position gathers, hundreds per volume move, among them)
template <bool LDSPOS>
__device__ __forceinline__ void dev_rescale(const VolCtx& c, int l, const double* recip, const double* hnew, int lane)
{
    const double invPi = 1.0 / 3.141592653589793238462643383279502884197;
    double* Pg = c.pos_g + (size_t)l * c.N * 3;
    double* Ps = c.spos + (size_t)l * c.N * 3;
    for (int i = lane; i < c.N; i += 64) {
        const double* p = LDSPOS ? Ps + 3 * i : Pg + 3 * i;
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
        if constexpr (LDSPOS) { Ps[3 * i] = n0; Ps[3 * i + 1] = n1; Ps[3 * i + 2] = n2; }
    }
}

// compute_model_energy of lattice l by one wavefront (value in every lane).  `mom_l` (walkers entirely in LDS only): every
// molecule's moments too, [N][kMomStride] -- what the translations' moment path reads (move_energy_mom_wave)
// (BATCH4: the distance tests' gathers four at a time -- for the look-ahead builds, which have the registers)
template <bool LDSPOS, bool BATCH4 = false>
__device__ __forceinline__ double dev_wave_model_energy(const VolCtx& c, int l, int lane, double* __restrict__ mom_l = nullptr)
{
    const double* Pg = c.pos_g + (size_t)l * c.N * 3;
    const double* Ps = c.spos + (size_t)l * c.N * 3;
    const double* IVl = c.siv + (size_t)l * c.ivcap * 3;
    const uint32_t* Lg = c.list_g + (size_t)l * c.S * c.N;
    const int* ORD = c.order_g + (size_t)l * c.N;
    const int* NNS = c.nns_g + (size_t)l * c.N;
    const int* CM = c.cmax_g + (size_t)l * ((c.N + 63) >> 6);
    auto getiv = [&](int k, double& x, double& y, double& z) { x = IVl[3 * k]; y = IVl[3 * k + 1]; z = IVl[3 * k + 2]; };
    auto getpos = [&](int j, double& x, double& y, double& z) {
        const double* p = LDSPOS ? Ps + 3 * (size_t)j : Pg + 3 * (size_t)j;
        x = p[0]; y = p[1]; z = p[2];
    };
    double esum = 0.0;
    if (c.srow) {
        // a walker entirely in LDS: its rows are there too (molecule-major, 16-bit entries) -- one lane per molecule, no list read
        // from global memory (the slot-major list cost three dependent global round trips of ~1.5 us each: 9 of a volume move's 19 us)
        uint32_t cur[8];
        for (int base = 0; base < c.N; base += 64) {
            const int mol = base + lane;
            const bool act = mol < c.N;
            const int n = act ? (int)c.snn[l * c.N + mol] : 0;
            const int nmax = __builtin_amdgcn_readfirstlane(wave_max_i(n));
            const unsigned short* row = c.srow + ((size_t)l * c.N + (act ? mol : 0)) * c.rstride;
            auto ent = [&](int s) -> uint32_t { const uint32_t e = s < n ? (uint32_t)row[s] : 0u; return (e & 63u) | ((e >> 6) << kJBits); };
            AtomSum a = atom_energy<64, BATCH4, true, kSweepQCap>(ListRsrc(), kNoColumn, kNoColumn, act ? mol : 0, n, nmax, 0, c.N, c.S, c.queue, getpos, getiv, cur,
                                                                 (mom_l && act) ? mom_l + (size_t)mol * kMomStride : nullptr, ent);
            if (act) esum += a.e;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off, 64);
        return esum;
    }
    const ListRsrc rs = list_rsrc(Lg, c.N, c.S);
    uint32_t cur[8];
    int n_cur = 0, mol = 0;
    uint32_t col = kNoColumn;
    if (lane < c.N) { col = (uint32_t)lane * 4u; n_cur = NNS[lane]; mol = ORD[lane]; }
    for (int base = 0; base < c.N; base += 64) {                 // wave-uniform: one group of 64 list columns per pass
        const bool act = col != kNoColumn;
        const int tn = base + 64 + lane;
        uint32_t col_next = kNoColumn;
        int n_next = 0, mol_next = 0;
        if (tn < c.N) { col_next = (uint32_t)tn * 4u; n_next = NNS[tn]; mol_next = ORD[tn]; }
        const int cm = __builtin_amdgcn_readfirstlane(CM[base >> 6]);
        AtomSum a = atom_energy<64, false, true, kSweepQCap>(rs, col, col_next, mol, act ? (n_cur & 0xff) : 0, cm & 0xff, cm >> 8, c.N, c.S, c.queue, getpos, getiv, cur,
                                                             (mom_l && act) ? mom_l + (size_t)mol * kMomStride : nullptr);
        if (act) esum += a.e;
        n_cur = n_next; mol = mol_next; col = col_next;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off, 64);
    return esum;
}

// -------------------------------------------------------------------------------------
// The workgroup of a walker: one wavefront per lattice.  Each wavefront evaluates ITS lattice (the fused old/new local
// energy of a translation, the rescaled box of a volume move); wavefront 0 then takes the move's decision -- order
// parameter, weights, Metropolis test, Wang-Landau update, lattice switch -- and hands {accepted, active lattice} back.
// Two workgroup barriers per move for two lattices, none for one.
// -------------------------------------------------------------------------------------
template <int NW>                                           // NW = wavefronts in the workgroup
__device__ __forceinline__ void wg_sync()
{
    if constexpr (NW == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// moves per batch of uniforms (Philox calls are lanes of one pass): 16, or 8 for the builds with volume moves -- 512 B of LDS that an
// NPT walker of the reference's examples does not have (sweep_lds); the smaller batch costs the translation-only build ~2 %
__host__ __device__ constexpr int sweep_batch(bool withvol, int spec = 1) { return spec > (withvol ? 8 : 16) ? spec : (withvol ? 8 : 16); }
constexpr unsigned kSweepScratch = (unsigned)((sizeof(WaveScratch) + 15) & ~(size_t)15);
constexpr unsigned kSweepScratchVol = (unsigned)(((kSweepQCap + 1) * 64 * sizeof(uint32_t)) > kSweepScratch ? ((kSweepQCap + 1) * 64 * sizeof(uint32_t)) : kSweepScratch);
static_assert(kSweepScratchVol == kSweepScratch, "the builds with volume moves take no more LDS per wavefront than the others");

// Dynamic LDS of a walker's workgroup (byte offsets), the same arithmetic on the host (launch size) and on the device.
// Every byte counts for the reference's own 48-molecule cells: eight walkers share a CU when a workgroup's static + dynamic
// LDS stays within 160 KiB / 8 = 20480 B (mw_sweep_translation_launch).
struct SweepLds { unsigned iv, pos, tab, uni, mv, scr, row, nn, mom, lmask, inmask, rec, total, scr_bytes; };
// (a volume move's full-box energy by all wavefronts of the workgroup: two lattices entirely in LDS, four or more moves in flight)
__host__ __device__ constexpr bool sweep_split(int L, bool ldslist, bool withvol, int spec) { return ldslist && withvol && L == 2 && spec >= 4; }
__host__ __device__ inline SweepLds sweep_lds(int L, int nw, int ivcap, int N, int nbins, bool ldspos, bool ldslist, int rstride, bool withvol,
                                              bool samplerun, int spec = 1)
{
    SweepLds o;
    unsigned p = 0;
    o.iv = p;  p += (unsigned)L * ivcap * 24u;                         // image vectors [L][ivcap][3]
    o.pos = p; p += ldspos ? (unsigned)L * N * 24u : 0u;               // positions     [L][N][3]     (small systems)
    o.tab = p; p += L == 2 ? (samplerun ? 5u : 4u) * nbins * 8u : 0u;  // weight, mu_bin, binwidth, histogram; unbiased_hist in a sample run only
    o.uni = p; p += (unsigned)sweep_batch(withvol, spec) * 8u * 8u;    // uniforms of a batch of moves [kUB][8]
    o.mv = o.uni;                                                      // a translation's molecule + displacement {x, y, z, imol}: written over its
                                                                       // spent uniforms u0..u3 (a volume move keeps its own: it reads them again)
    p = (p + 15u) & ~15u;
    o.scr_bytes = withvol ? kSweepScratchVol : kSweepScratch;          // per wavefront: WaveScratch / the full-box energy's queue
    o.scr = p; p += (unsigned)nw * o.scr_bytes;
    o.row = p; p += ldslist ? (unsigned)L * N * rstride * 2u : 0u;     // list rows, 16-bit entries (j | image << 6; N <= 64)
    o.nn = p;  p += ldslist ? (unsigned)L * N : 0u;                    // row lengths, one byte each
    p = (p + 15u) & ~15u;
    // look-ahead builds (a handful of walkers: LDS to spare) keep the moment path's data here: every molecule's moments, current and
    // a volume move's trial set [2][L][N][kMomStride], and every row's molecules as a bit mask [L][N]
    const bool momlds = ldslist && spec > 1;
    o.mom = p;   p += momlds ? 2u * L * N * (unsigned)kMomStride * 8u : 0u;
    o.lmask = p; p += momlds ? (unsigned)L * N * 8u : 0u;
    const bool split = sweep_split(L, ldslist, withvol, spec);
    o.inmask = p; p += split ? (unsigned)L * N * 4u : 0u;
    p = (p + 15u) & ~15u;
    o.rec = p;    p += split ? (unsigned)L * N * 12u * 48u : 0u;              // [L][N][kSplitQ][6] doubles
    o.total = (p + 15u) & ~15u;
    return o;
}

// SPLIT: a volume move's full-box energy (compute_model_energy, molint.F90:407-499) spread over ALL wavefronts of a look-ahead
// workgroup -- the builds of a handful of walkers, whose wavefronts beyond one per lattice have nothing else to do during a volume
// move, and whose speed is one chain's: the one-wavefront routine above walks a molecule's row and its ~8 in-range neighbours' pair
// terms (rsqrt, reciprocal, exp) one after the other, 7 of a volume move's 16 us.  Walkers entirely in LDS, lane = molecule,
// wavefront `part` of `nparts` of the lattice:
//   A  distance tests of row slots part, part + nparts, ...: the in-range slots are OR-ed into inmask[molecule];
//   B  in-range neighbour q (in list order) of every molecule by wavefront q mod nparts: {d, 1/r, e1, g} into rec[molecule][q];
//   C  the lattice's first wavefront adds the records up in list order with atom_energy's own arithmetic (MomentSums): the energy
//      and the moments are atom_energy's bit for bit, so the chain does not depend on the look-ahead of the build that runs it.
// Between the phases: the caller's workgroup barriers.  B reports a molecule with more than kSplitQ in-range neighbours (`overflow`):
// the caller then takes the one-wavefront routine.
__device__ __forceinline__ void dev_split_tests(const VolCtx& c, int l, int part, int nparts, int lane)
{
    const int mol = lane < c.N ? lane : 0;
    const bool act = lane < c.N;
    const double* Ps = c.spos + (size_t)l * c.N * 3;
    const double* IVl = c.siv + (size_t)l * c.ivcap * 3;
    const int n = act ? (int)c.snn[l * c.N + mol] : 0;
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max_i(n));
    const unsigned short* row = c.srow + ((size_t)l * c.N + mol) * c.rstride;
    const double xi = Ps[3 * mol], yi = Ps[3 * mol + 1], zi = Ps[3 * mol + 2];
    unsigned m = 0u;
    for (int s = part; s < nmax; s += nparts) {
        const uint32_t e = s < n ? (uint32_t)row[s] : 0u;
        const double* pj = Ps + 3 * (size_t)(e & 63u);
        const double* iv = IVl + 3 * (size_t)(e >> 6);
        const double dx = (pj[0] + iv[0]) - xi, dy = (pj[1] + iv[1]) - yi, dz = (pj[2] + iv[2]) - zi;     // molint.F90:447,450
        const double r2 = dist2(dx, dy, dz);
        if (s < n && r2 < kRcSq) m |= 1u << s;                                                          // :454
    }
    if (m != 0u) atomicOr(&c.inmask[l * c.N + mol], m);
}

__device__ __forceinline__ bool dev_split_records(const VolCtx& c, int l, int part, int nparts, int lane)
{
    const int mol = lane < c.N ? lane : 0;
    const bool act = lane < c.N;
    const double* Ps = c.spos + (size_t)l * c.N * 3;
    const double* IVl = c.siv + (size_t)l * c.ivcap * 3;
    const unsigned short* row = c.srow + ((size_t)l * c.N + mol) * c.rstride;
    const double xi = Ps[3 * mol], yi = Ps[3 * mol + 1], zi = Ps[3 * mol + 2];
    const unsigned mask = act ? c.inmask[l * c.N + mol] : 0u;
    const int cnt = __popc(mask);
    double* R = c.rec + ((size_t)l * c.N + mol) * kSplitQ * 6;
    for (int q = part; q < kSplitQ; q += nparts) {            // (uniform bounds; a lane with fewer in-range neighbours sits the step out)
        if (q < cnt) {
            unsigned mm = mask;
            for (int t = 0; t < q; ++t) mm &= mm - 1u;        // the q-th in-range slot
            const int s = __ffs((int)mm) - 1;
            const uint32_t e = (uint32_t)row[s];
            const double* pj = Ps + 3 * (size_t)(e & 63u);
            const double* iv = IVl + 3 * (size_t)(e >> 6);
            const double dx = (pj[0] + iv[0]) - xi, dy = (pj[1] + iv[1]) - yi, dz = (pj[2] + iv[2]) - zi;
            const double r2 = dist2(dx, dy, dz);
            double rinv, e1, g;
            pair_terms(r2, rinv, e1, g);                                                  // :456-462
            double2* r2p = reinterpret_cast<double2*>(R + 6 * q);
            r2p[0] = make_double2(dx, dy); r2p[1] = make_double2(dz, rinv); r2p[2] = make_double2(e1, g);
        }
    }
    return __ballot(cnt > kSplitQ) != 0ull;
}

// (value in every lane, like dev_wave_model_energy)
__device__ __forceinline__ double dev_split_sum(const VolCtx& c, int l, int lane, double* __restrict__ mom_l, double* lane_e = nullptr)
{
    const int mol = lane < c.N ? lane : 0;
    const bool act = lane < c.N;
    const int cnt = act ? __popc(c.inmask[l * c.N + mol]) : 0;
    const double* R = c.rec + ((size_t)l * c.N + mol) * kSplitQ * 6;
    MomentSums ms;
    for (int q = 0; q < cnt; ++q) {
        const double2* r2p = reinterpret_cast<const double2*>(R + 6 * q);
        const double2 a = r2p[0], b = r2p[1], d = r2p[2];
        ms.add(a.x, a.y, b.x, b.y, d.x, d.y);
    }
    double esum = 0.0;
    const double e = ms.finish(cnt, (mom_l && act) ? mom_l + (size_t)mol * kMomStride : nullptr);
    if (lane_e) *lane_e = e;
    if (act) esum += e;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off, 64);
    return esum;
}

// Volume move of one walker (mc_volume, mc_moves.F90:1216-1534; ref_ljr, which only chain synchronisation reads, is not
// carried).  Rare (probability ~1/N per move).  One symmetric hmatrix element of both lattices changes; every wavefront
// rescales ITS lattice through fractional coordinates (lanes over molecules), rebuilds its image vectors in the
// reference's order and arithmetic and recomputes its full-box energy WITH THE EXISTING LISTS (atom_energy over the
// slot-major list); wavefront 0 decides; on rejection every wavefront puts its lattice back the way the reference does
// (positions mapped back through the NEW reciprocal matrix, :1413-1506).
// Returns (every wavefront): 1 accepted, 0 rejected, -1 rejected because a cell needed more image vectors than ivcap.
// With look-ahead (NW > NLAT wavefronts) the wavefronts beyond the first NLAT have no lattice of their own here: they keep
// the workgroup's barriers company.
template <int NLAT, int NW, bool LDSPOS, bool SPLIT, typename DecideFn>
__device__ __forceinline__
int volume_move_wg(const VolCtx& c, const double* __restrict__ U, double dv_max, int wv, int lane,
                   double* __restrict__ sx, int* __restrict__ sdec, DecideFn decide)
{
    constexpr int L = NLAT;
    const bool active = wv < NLAT;
    const int l = active ? wv : 0;                                                 // this wavefront's lattice
    [[maybe_unused]] const unsigned long long tv0 = MW_SW_NOW();
    // the old cell of this lattice, kept in LDS (c.sbk: [lattice][hmatrix 9 | recip 9 | new recip 9]): eighteen wave-uniform
    // doubles are thirty-six vector registers, held across the full-box energy evaluation
    double* bk_h = c.sbk + 27 * l;
    double* bk_r = bk_h + 9;
    double* bk_n = bk_h + 18;
    if (active && lane < 9) { bk_h[lane] = c.shmat[l * 9 + lane]; bk_r[lane] = c.srecip[l * 9 + lane]; }
    const double old_vol_l = c.svol[l];
    const int idim = (int)(U[0] * 3.0) + 1, jdim = (int)(U[1] * 3.0) + 1;                       // :1269-1272
    const double dh = (2.0 * U[2] - 1.0) * dv_max;                                              // :1276
    wg_sync<NW>();                                     // (everybody has read the old cells)
    if (active && lane == 0) {                                                                  // :1281-1282
        MW_HM(c.shmat + 9 * l, idim, jdim) = MW_HM(c.shmat + 9 * l, idim, jdim) + dh;
        if (idim != jdim) MW_HM(c.shmat + 9 * l, jdim, idim) = MW_HM(c.shmat + 9 * l, jdim, idim) + dh;
    }
    wg_sync<NW>();
    // The reference takes the lattices in turn and stops at the first whose new cell needs more image vectors than there
    // is room for (:1285-1358; here: the move counts as rejected and is flagged): lattice 2 is then never touched.
    bool bad0 = false;
    if (L == 2 && l == 1) {
        const double* h = c.shmat;
        const double rc = kSmallA * kSigma;
        const int im = (int)floor(rc / sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2])) + 1;
        const int jm = (int)floor(rc / sqrt(h[3] * h[3] + h[4] * h[4] + h[5] * h[5])) + 1;
        const int km = (int)floor(rc / sqrt(h[6] * h[6] + h[7] * h[7] + h[8] * h[8])) + 1;
        bad0 = (2 * im + 1) * (2 * jm + 1) * (2 * km + 1) > c.ivcap;
    }
    double new_e = 0.0;
    int bad = 0;
    bool rescaled = false;
    [[maybe_unused]] const unsigned long long tv1 = MW_SW_NOW();
    MW_SW_ACC(34, tv1 - tv0);
    if (active && !bad0) {
        dev_rescale<LDSPOS>(c, l, bk_r, c.shmat + 9 * l, lane);
        rescaled = true;
        wave_sync();
        MW_SW_ACC(35, MW_SW_NOW() - tv1);
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
        wave_sync();
        [[maybe_unused]] const unsigned long long tv2 = MW_SW_NOW();
        if (niv < 0) bad = 1;
        else if constexpr (!SPLIT) new_e = dev_wave_model_energy<LDSPOS, (NW > NLAT)>(c, l, lane, c.mom_trial ? c.mom_trial + (size_t)l * c.N * kMomStride : nullptr);
        MW_SW_ACC(36, MW_SW_NOW() - tv2); MW_SW_ACC(37, tv2 - tv1);
    }
    if constexpr (SPLIT) {
        // the full-box energy by every wavefront of the workgroup (see dev_split_tests): sdec[2 + lattice] != 0 -- a cell that is
        // not to be evaluated (bad, or lattice 2 after a bad lattice 1); sdec[0] -- a molecule with more in-range neighbours than
        // the records hold
        [[maybe_unused]] const unsigned long long tv2 = MW_SW_NOW();
        constexpr int P = NW / NLAT;
        const int lw = wv % NLAT, part = wv / NLAT;
        if (active && lane == 0) sdec[2 + l] = bad | (bad0 ? 2 : 0);
        if (part == 1 && lane < c.N) c.inmask[lw * c.N + lane] = 0u;                 // (a wavefront that is idle until here)
        if (wv == NLAT && lane == 0) sdec[0] = 0;
        wg_sync<NW>();                                  // the trial cell -- positions, image vectors -- is there for everybody
        const bool run = sdec[2 + lw] == 0;
        if (run) dev_split_tests(c, lw, part, P, lane);
        wg_sync<NW>();
        if (run && dev_split_records(c, lw, part, P, lane) && lane == 0) sdec[0] = 1;
        wg_sync<NW>();
        if (active && !bad0 && !bad) {
            double* mom_l = c.mom_trial ? c.mom_trial + (size_t)l * c.N * kMomStride : nullptr;
            new_e = sdec[0] != 0 ? dev_wave_model_energy<LDSPOS, true>(c, l, lane, mom_l) : dev_split_sum(c, l, lane, mom_l);
#ifdef MW_SPLIT_CHECK     // diagnostic build (tools/variants.py splitcheck): the split sum against the one-wavefront routine, molecule by molecule
                          // -- g_sweep_stamps[44] lattice energies checked, [45] of them with a molecule that differs, [43], [46], [47] the last such
            if (sdec[0] == 0) {
                double es = 0.0;
                (void)dev_split_sum(c, l, lane, nullptr, &es);
                const int mol = lane < c.N ? lane : 0;
                const int n = lane < c.N ? (int)c.snn[l * c.N + mol] : 0;
                const int nmax = __builtin_amdgcn_readfirstlane(wave_max_i(n));
                const unsigned short* row = c.srow + ((size_t)l * c.N + mol) * c.rstride;
                const double* Ps = c.spos + (size_t)l * c.N * 3;
                const double* IVl = c.siv + (size_t)l * c.ivcap * 3;
                auto getiv = [&](int k, double& x, double& y, double& z) { x = IVl[3 * k]; y = IVl[3 * k + 1]; z = IVl[3 * k + 2]; };
                auto getpos = [&](int j, double& x, double& y, double& z) { const double* p = Ps + 3 * (size_t)j; x = p[0]; y = p[1]; z = p[2]; };
                auto ent = [&](int s) -> uint32_t { const uint32_t e = s < n ? (uint32_t)row[s] : 0u; return (e & 63u) | ((e >> 6) << kJBits); };
                uint32_t cur[8];
                AtomSum a = atom_energy<64, true, true, kSweepQCap>(ListRsrc(), kNoColumn, kNoColumn, mol, n, nmax, 0, c.N, c.S, c.queue, getpos, getiv, cur, nullptr, ent);
                const bool bad_l = lane < c.N && a.e != es;
                const unsigned long long bm = __ballot(bad_l);
                if (blockIdx.x == 0 && lane == 0) atomicAdd(&g_sweep_stamps[44], 1ull);
                if (bm != 0ull && blockIdx.x == 0 && lane == __ffsll((long long)bm) - 1) {
                    atomicAdd(&g_sweep_stamps[45], 1ull);
                    g_sweep_stamps[43] = (unsigned long long)(a.cnt | (__popc(c.inmask[l * c.N + mol]) << 8) | (lane << 16) | (n << 24));
                    g_sweep_stamps[46] = (unsigned long long)__double_as_longlong(a.e); g_sweep_stamps[47] = (unsigned long long)__double_as_longlong(es);
                }
            }
#endif
        }
        MW_SW_ACC(36, MW_SW_NOW() - tv2);
    }
    [[maybe_unused]] const unsigned long long tv3 = MW_SW_NOW();
    if (active && lane == 0) { sx[l] = new_e; sdec[2 + l] = bad; }
    wg_sync<NW>();
    int ok = 0, anybad = 0;
    if (wv == 0) {
        anybad = sdec[2] | (L == 2 ? sdec[3] : 0);
        ok = decide(sx[0], L == 2 ? sx[1] : 0.0, anybad);          // (updates the walker's state; energies in every lane)
        if (lane == 0) { sdec[0] = ok; sdec[1] = anybad; }
    }
    wg_sync<NW>();
    [[maybe_unused]] const unsigned long long tv4 = MW_SW_NOW();
    MW_SW_ACC(38, tv4 - tv3);
    ok = sdec[0]; anybad = sdec[1];
    if (active && !ok) {                                                                         // :1426-1530
        if (lane < 9) bk_n[lane] = c.srecip[l * 9 + lane];
        wave_sync();
        if (lane < 9) { c.shmat[l * 9 + lane] = bk_h[lane]; c.srecip[l * 9 + lane] = bk_r[lane]; }
        if (lane == 0) c.svol[l] = old_vol_l;
        wave_sync();
        if (rescaled) {
            dev_rescale<LDSPOS>(c, l, bk_n, c.shmat + 9 * l, lane);                                      // back through the NEW recip
            const int niv = dev_compute_ivects(c.shmat + 9 * l, c.siv + (size_t)l * c.ivcap * 3,
                                               c.ivect_g + (size_t)l * c.ivcap * 3, c.ivcap, lane);   // :1510-1512
            if (lane == 0 && niv > 0) { c.sniv[l] = niv; c.nivect_g[l] = niv; }
        }
    }
    if (active && lane == 0) {                         // global mirrors of the cell
#pragma unroll
        for (int t = 0; t < 9; ++t) c.hmat_g[l * 9 + t] = c.shmat[l * 9 + t];
        c.vol_g[l] = c.svol[l];
    }
    wg_sync<NW>();
    MW_SW_ACC(39, MW_SW_NOW() - tv4); MW_SW_ACC(40, 1ull); MW_SW_ACC(41, MW_SW_NOW() - tv0);
    return anybad ? -1 : ok;
}

// =====================================================================================
// k_sweep: the device-resident Monte Carlo driver.  grid = walkers of the launch, block = 64 x NLAT.
//
// Per-walker tables (two lattices only): weight / histogram / unbiased_hist [walker][nbins]; every walker reads its OWN
// weights in eta_weight, so Wang-Landau updates stay local until the host synchronises them (comms_allreduce_eta/hist/
// uhist semantics, WalkerComms).  The three tables live in LDS for the launch and go back at its end.
//
// What bounds the kernel is vector-instruction issue of dependent chains (profiles/r03a_sweep_pmc_counters.txt), so the
// design is about the LENGTH of a move's chain and the number of chains in flight per SIMD:
//  * one wavefront per LATTICE, not per walker: the two local-energy evaluations of a move run side by side;
//  * the move's scalar arithmetic is lane-parallel (lane_mu_to_bin / lane_eta above, one exp stream per move) and done by
//    wavefront 0 only; bins are carried from move to move; the Wang-Landau minimum is tracked, not re-scanned;
//  * the uniforms, molecule and displacement of kUB moves come from one Philox pass of the whole workgroup;
//  * the walker's parameters and its state between moves (energies, order parameter, counters) live in LDS (WalkerCtl),
//    not in registers: a wave-uniform double costs two VECTOR registers, and thirty of them held across the energy
//    evaluation were the difference between two and four wavefronts per SIMD.
// =====================================================================================
struct WalkerCtl {
    // the launch's parameters as this walker sees them (its own window, step sizes, increment)
    double beta, pressure, dref, av_binwidth, log_unbiased_norm, transP, ref1, ref2, wl_alpha, orig_wl_factor, mu_min, mu_max;
    double max_trans, dv_max;
    MuGridDev mg;                            // (mg.in_window changes at the top of a 'dd' cycle)
    int record, samplerun, always_switch, npt, swetnam, dd, minu, eq_cycles, nbins;
    int tab_small;                           // every |weight| < 2^20 (the lattice switch's shortcut; kept up with every update)
    // the walker's state between moves
    double men0, men1, ls_mu, gauge, wlf, sumh, cur_min, lgv12, lgv21;
    unsigned long long acc, nsw, nvol_try, nvol_acc;
    int ls, k_cur, k_valid, flag, cyc, within;
};

// The MINU branch of both move types: the lattice the move would end in; diffkT rewritten with the switch's terms if it differs.
// E = trial energies, V = trial volumes, Eb / Vb = energy and volume of the CURRENT lattice before the move.
__device__ __forceinline__ int dev_minu_branch(const WalkerCtl& sp, int ls, double E1, double E2, double V1, double V2,
                                               double Eb, double Vb, bool vol_terms, int N, double new_eta, double old_eta,
                                               double& diffkT)
{
    const double h1 = E1 + sp.pressure * V1 - sp.ref1, h2 = E2 + sp.pressure * V2 - sp.ref2;   // minloc, :1122-1126
    const int lsn = h2 < h1 ? 2 : 1;
    if (lsn != ls) {
        const double En = lsn == 1 ? E1 : E2, Vn = lsn == 1 ? V1 : V2;
        double d;
        if (vol_terms) d = sp.beta * En - sp.beta * Eb + sp.beta * sp.pressure * (Vn - Vb) - (double)N * fast_log_pos(Vn / Vb) + new_eta - old_eta;   // :1131-1133,1396-1397
        else           d = sp.beta * En - sp.beta * Eb + new_eta - old_eta;                                                                  // :1135
        if (sp.ref1 != 0.0 || sp.ref2 != 0.0)                                                                                               // leshift, :1134,1136,1398
            d = d - sp.beta * (lsn == 1 ? sp.ref1 : sp.ref2) + sp.beta * (ls == 1 ? sp.ref1 : sp.ref2);
        diffkT = d;
    }
    return lsn;
}

// SPEC > 1: LOOK-AHEAD for walkers that cannot fill the chip by their number (a few thousand 4096-molecule boxes leave two
// wavefronts per SIMD; a few dozen leave almost all of it idle).  The workgroup holds SPEC wavefronts per lattice and
// evaluates SPEC consecutive translations of the chain AT ONCE, all from the configuration the round starts with: the
// random stream is counter-based, so move m + s knows its molecule and displacement without waiting for move m.  The
// decisions are then taken in order by wavefront 0.  An evaluation is exactly the one the sequential chain would have made
// unless an EARLIER move of the round was accepted and moved a molecule whose position this one read (move_energy_wave
// reports which, bit per earlier slot), or changed the active lattice (the partner lattice's displacement depends on it):
// the round then ends before that move, and the next round starts with it.  Nothing is approximated -- the chain is the
// sequential one, move for move and bit for bit -- and a round costs one barrier more than a move did.
template <int NLAT, int SPEC, bool LDSPOS, bool LDSLIST, bool WITHVOL>
// Four wavefronts per SIMD are the design point (128 VGPRs; the translation-only build needs 127-136 left to itself, and
// which side of 128 it lands on depends on what else is compiled with it).  The builds that carry mc_volume keep to it too
// -- the reference's own examples are NPT -- since the volume move's addresses are worked out inside its branch (below) and
// its old volumes wait in LDS; only two lattices + look-ahead + volume moves (few walkers by construction) get three.
#ifdef MW_SWEEP_WAVES_CAP     // diagnostic builds only (tools/variants.py): e.g. 5 -> 96 vector registers, a build that SPILLS, to show that one is still correct
__global__ __launch_bounds__(64 * NLAT * SPEC) __attribute__((amdgpu_waves_per_eu(MW_SWEEP_WAVES_CAP, MW_SWEEP_WAVES_CAP)))
#else
__global__ __launch_bounds__(64 * NLAT * SPEC) __attribute__((amdgpu_waves_per_eu((NLAT == 2 && SPEC > 1) ? 3 : 4, (NLAT == 2 && SPEC > 1) ? 3 : 4)))
#endif
void k_sweep(double* pos, double* hmat, double* ivect,
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
             int* __restrict__ winflag, const double* __restrict__ wstep,
             double* wmom)     // LDSLIST: [walker of the launch][2][L][N][kMomStride], the moment path's scratch (nullptr: the path is off)
{
    static_assert(!LDSLIST || LDSPOS, "rows in LDS: positions too");
    constexpr int L = NLAT, NW = NLAT * SPEC, NTHR = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double shmat[2][9], srecip[2][9], svol[2];      // the walker's cells: volume moves change them in place
    __shared__ int sniv[2];
    __shared__ double sbk[2][27];     // a volume move's old hmatrix / recip and new recip per lattice
    __shared__ double svold[2];       // ... and the old volumes (read by the decision after the full-box energies)
    __shared__ double sx[2 * NW < 4 ? 4 : 2 * NW];   // what every wavefront hands to the deciding one: its {e_old, e_new} (volume moves: full-box energies)
    __shared__ unsigned scm[NW];      // ... and which earlier moves of the round its evaluation depends on
    __shared__ int sdec[4 + SPEC];    // the decisions: moves decided this round, active lattice, (volume moves: accepted, bad, bad per lattice), accepted per slot
    __shared__ WalkerCtl ctl;
    WalkerCtl& C = ctl;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
    const int slot = wv / NLAT, lat = wv % NLAT;          // this wavefront's move of the round, its lattice
    const int wlk = walker0 + blockIdx.x;
    const int box0 = wlk * L;
    const int nbins = sp.nbins;
    const SweepLds lay = sweep_lds(L, NW, ivcap, N, nbins, LDSPOS, LDSLIST, rstride, WITHVOL, sp.samplerun != 0, SPEC);
    double* siv = reinterpret_cast<double*>(smem_raw + lay.iv);
    double* spos = reinterpret_cast<double*>(smem_raw + lay.pos);
    double* sweight = reinterpret_cast<double*>(smem_raw + lay.tab);
    double* smub = sweight + nbins;
    double* sbw = smub + nbins;
    double* shist = sbw + nbins;
    double* suhist = shist + nbins;                       // (there in a sample run only)
    constexpr int MVS = 8;                                // doubles between two moves' {x, y, z, imol} (sweep_lds)
    double* suni = reinterpret_cast<double*>(smem_raw + lay.uni);
    double* smv = reinterpret_cast<double*>(smem_raw + lay.mv);
    WaveScratch* ws = reinterpret_cast<WaveScratch*>(smem_raw + lay.scr + (unsigned)wv * lay.scr_bytes);
    unsigned short* srow = reinterpret_cast<unsigned short*>(smem_raw + lay.row);
    unsigned char* snn = smem_raw + lay.nn;
    constexpr bool MOMLDS = LDSLIST && SPEC > 1;          // (see sweep_lds)
    double* smom = reinterpret_cast<double*>(smem_raw + lay.mom);
    unsigned long long* slmask = reinterpret_cast<unsigned long long*>(smem_raw + lay.lmask);
    constexpr bool SPLIT = sweep_split(NLAT, LDSLIST, WITHVOL, SPEC);
    static_assert(kSplitQ == 12, "sweep_lds sizes the records for twelve per molecule");
    const double invPi = 1.0 / 3.141592653589793238462643383279502884197;

    // ---- staging: image vectors, (small systems) positions, list rows and row lengths, the walker's tables -------------
    for (int l = 0; l < L; ++l) {
        const int niv = nivect[box0 + l];
        for (int t = tid; t < niv * 3; t += NTHR) siv[(size_t)l * ivcap * 3 + t] = ivect[(size_t)(box0 + l) * ivcap * 3 + t];
        if (LDSPOS) {
            const double* Pg = pos + (size_t)(box0 + l) * N * 3;
            for (int t = tid; t < 3 * N; t += NTHR) spos[(size_t)l * N * 3 + t] = Pg[t];
        }
        if (LDSLIST) {    // N <= 64: an entry (j, image) fits 16 bits.  (Slots past a row's end are never live.)
            const uint32_t* LMg = listm + (size_t)(box0 + l) * N * kRow;
            for (int t = tid; t < N * rstride; t += NTHR) {
                const uint32_t e = LMg[(size_t)(t / rstride) * kRow + (t % rstride)];
                srow[(size_t)l * N * rstride + t] = (unsigned short)((e & 63u) | ((e >> kJBits) << 6));
            }
            for (int t = tid; t < N; t += NTHR) snn[l * N + t] = (unsigned char)nn[(size_t)(box0 + l) * N + t];
        }
    }
    if (L == 2) {
        for (int t = tid; t < nbins; t += NTHR) {
            sweight[t] = wweight[(size_t)wlk * nbins + t];
            smub[t] = mu_bin_g[t];
            sbw[t] = binwidth_g[t];
            shist[t] = whist[(size_t)wlk * nbins + t];
            if (sp.samplerun) suhist[t] = wuhist[(size_t)wlk * nbins + t];
        }
    }
    if (tid < L) {
        const int l = tid;
        double rcp[9];
        dev_recipmatrix(hmat + (size_t)(box0 + l) * 9, rcp);
#pragma unroll
        for (int t = 0; t < 9; ++t) { srecip[l][t] = rcp[t]; shmat[l][t] = hmat[(size_t)(box0 + l) * 9 + t]; }
        svol[l] = volume[box0 + l];
        sniv[l] = nivect[box0 + l];
    }
    if (tid == 0) {
        // per-walker pieces of the parameter block: its window of the overlap parameter ('dd': mc_moves.F90:659-709; else the
        // whole range), whether it has reached that window, its step sizes, its Wang-Landau increment and Swetnam's visit total
        C.beta = sp.beta; C.pressure = sp.pressure; C.dref = sp.dref; C.av_binwidth = sp.av_binwidth;
        C.log_unbiased_norm = sp.log_unbiased_norm; C.transP = sp.transP; C.ref1 = sp.ref1; C.ref2 = sp.ref2;
        C.wl_alpha = sp.wl_alpha; C.orig_wl_factor = sp.orig_wl_factor; C.mu_min = sp.mu_min; C.mu_max = sp.mu_max;
        C.max_trans = sp.max_trans; C.dv_max = sp.dv_max;
        if (wstep) { C.max_trans = wstep[2 * (size_t)wlk]; C.dv_max = wstep[2 * (size_t)wlk + 1]; }   // equilibration tuning, :1729-1732
        C.mg.c_pos = (1.0 - sp.r_pos) / sp.a_pos; C.mg.c_neg = (1.0 - sp.r_neg) / sp.a_neg;
        C.mg.ilr_pos = 1.0 / log(sp.r_pos); C.mg.ilr_neg = 1.0 / log(sp.r_neg);
        C.mg.mu_lo = sp.mu_lo; C.mg.mu_hi = sp.mu_hi; C.mg.nbins = nbins; C.mg.start_bin = sp.start_bin; C.mg.end_bin = sp.end_bin;
        C.mg.eta_interp = sp.eta_interp;
        if (wwin) {
            C.mg.start_bin = (int)wwin[4 * (size_t)wlk]; C.mg.end_bin = (int)wwin[4 * (size_t)wlk + 1];
            C.mg.mu_lo = wwin[4 * (size_t)wlk + 2]; C.mg.mu_hi = wwin[4 * (size_t)wlk + 3];
        }
        C.mg.in_window = sp.dd ? winflag[wlk] : 1;                            // mc_moves.F90:112,872
        C.record = sp.record; C.samplerun = sp.samplerun; C.always_switch = sp.always_switch; C.npt = sp.npt;
        C.swetnam = sp.swetnam; C.dd = sp.dd; C.minu = sp.minu; C.eq_cycles = sp.eq_cycles; C.nbins = nbins;
        C.men0 = energy[box0]; C.men1 = L == 2 ? energy[box0 + 1] : 0.0;
        C.ls_mu = wmu[wlk]; C.ls = wls[wlk];                                  // active lattice, 1-based
        C.gauge = 0.0;                                                        // total of the minima subtracted from this walker's weights (:1682-1685)
        C.wlf = L == 2 ? wfac[wlk] : 0.0;                                     // wl_factor of this walker (:1615,1677)
        C.sumh = L == 2 ? wsum[wlk] : 0.0;                                    // sumhist (:94,1638)
        C.acc = 0; C.nsw = 0; C.nvol_try = 0; C.nvol_acc = 0; C.flag = 0;
        C.k_cur = 0; C.k_valid = 0;                                           // bin of ls_mu, carried from move to move
        // mc_cycle_num of the next move and its place inside the cycle ('dd' only: the equilibration rules, :181-210)
        C.cyc = sp.dd ? (int)(move0 / (unsigned long long)N) + 1 : 0;
        C.within = sp.dd ? (int)(move0 % (unsigned long long)N) : 0;
        sdec[2] = sdec[3] = (LDSLIST && wmom) ? 1 : 0;                        // (moment path of walkers in LDS: lattice l's moments are to be made, see the move loop)
    }
    __syncthreads();
    if constexpr (MOMLDS) {
        for (int t = tid; t < L * N; t += NTHR) {
            unsigned long long m = 0ull;
            const int n = snn[t];
            for (int q = 0; q < n; ++q) m |= 1ull << (srow[(size_t)t * rstride + q] & 63u);
            slmask[t] = m;
        }
    }
    if (wv == 0) {
        // log(V1/V2), log(V2/V1): change with volume moves only; the minimum of the weights over the walker's window (0
        // after the first update)
        double l12 = 0.0, l21 = 0.0, cmin = 0.0;
        if (L == 2) {
            l12 = fast_log_pos(svol[0] / svol[1]); l21 = fast_log_pos(svol[1] / svol[0]);
            if (C.record && !C.samplerun) {
                double mn = 1.7976931348623157e308;
                for (int b = C.mg.start_bin - 1 + lane; b < C.mg.end_bin; b += 64) { const double w = sweight[b]; mn = w < mn ? w : mn; }
                cmin = readlane_f64(dpp_wave_min(mn), 63);
            }
        }
        bool sm = true;
        if (L == 2) for (int b = lane; b < nbins; b += 64) sm = sm && fabs(sweight[b]) < 1048576.0;
        const bool allsm = __ballot(!sm) == 0ull;
        if (lane == 0) { C.lgv12 = l12; C.lgv21 = l21; C.cur_min = cmin; C.tab_small = allsm ? 1 : 0; }
    }
    __syncthreads();


    // mc_lattice_switch's exponent for a walker in lattice lsx with energies E0, E1 (:1557-1572) in two parts: the energy /
    // volume terms that stand BEFORE "+ new_eta - old_eta" in the reference's expression, and the leshift terms added after it
    auto switch_dk_terms = [&](double E0, double E1, int lsx, double& lesh) {
        const double Els = lsx == 1 ? E0 : E1, Elsn = lsx == 1 ? E1 : E0;
        const double V1 = svol[0], V2 = svol[1];
        const double Vls = lsx == 1 ? V1 : V2, Vlsn = lsx == 1 ? V2 : V1;
        double dk;
        if (C.npt) dk = C.beta * Elsn - C.beta * Els + C.beta * C.pressure * (Vlsn - Vls) - (double)N * (lsx == 1 ? C.lgv21 : C.lgv12);
        else       dk = C.beta * Elsn - C.beta * Els;
        lesh = lsx == 1 ? C.beta * C.dref : -(C.beta * C.dref);               // leshift: - beta ref(lsn) + beta ref(ls), :1567,1572
        return dk;
    };
    // ... and the whole of it less new_eta - old_eta (= eta_weight(ls_mu) - eta_weight(ls_mu): see post_move)
    auto switch_dk = [&](double E0, double E1, int lsx) {
        double lesh;
        const double dk = switch_dk_terms(E0, E1, lsx, lesh);
        return dk + lesh;
    };
    // What follows EITHER move type (wavefront 0): mc_update_wl_bins (:1597-1689) on the walker's tables, then one
    // mc_lattice_switch attempt (:1536-1594).  eta_fin = eta_weight(ls_mu) with the weights as the move found them,
    // cmp_sw = exp(-dk) of the switch without its eta terms, ufac = exp(eta_fin - log_unbiased_norm); C.k_cur = bin of ls_mu.
    auto post_move = [&](double eta_fin, double cmp_sw, double ufac, bool do_switch, double u6) -> int {
        int sw = 0;
        const int kc = C.k_cur;
        [[maybe_unused]] const unsigned long long sw_p0 = MW_SW_NOW();
        if (C.record) {                                                           // mc_update_wl_bins, :1597-1689
            const int k = kc;
            if (k >= 1 && k <= nbins) {
                const double bwk = sbw[k - 1];
                const double visit = C.av_binwidth / bwk;
                if (C.samplerun) {
                    if (lane == 0) {
                        shist[k - 1] = shist[k - 1] + visit;                                  // :1621
                        suhist[k - 1] = suhist[k - 1] + visit * ufac;                         // :1627-1629
                    }
                } else {
                    double wlf = C.wlf;
                    if (C.swetnam) {                                              // :1636-1653
                        const double sumh = C.sumh + 1.0;
                        double a2 = 0.0;
                        const double span = C.mu_max - C.mu_min - 1.0;
                        int lane_s = lane;                                        // (its table addresses are not worth a register held
                        asm volatile("" : "+v"(lane_s));                          //  through every move of every build)
                        for (int b = lane_s; b < nbins; b += 64) {
                            const double hb = shist[b] + (b == k - 1 ? visit : 0.0);          // this move's visit is already counted (:1621)
                            const double dev = hb * sbw[b] / sumh - sbw[b] / span;
                            a2 += dev * dev;
                        }
                        a2 = readlane_f64(dpp_wave_sum(a2), 63);
                        double f = sqrt(a2 / (double)nbins);
                        f = log(f) * C.wl_alpha * (double)nbins;
                        wlf = f < C.orig_wl_factor ? f : C.orig_wl_factor;
                        if (lane == 0) { C.sumh = sumh; C.wlf = wlf; }
                    }
                    // weight(k) += av_binwidth*wl_factor/binwidth(k) -- whichever bin k is (:1680); then the minimum over the
                    // walker's window is subtracted inside the window (:1682-1685; with 'dd' windows k may lie outside).
                    // The minimum is TRACKED: it is 0 after the first update, and it only moves when the visited bin was
                    // (one of) the lowest -- then, and only then, the window is scanned.
                    const double inc = C.av_binwidth * wlf / bwk;
                    const double wk = sweight[k - 1];
                    const double cur_min = C.cur_min;
                    const int sb = C.mg.start_bin, eb = C.mg.end_bin;
                    const bool k_in = k >= sb && k <= eb;
                    double mn;
                    bool scan = false;
                    if (!k_in) mn = cur_min;
                    else if (wk > cur_min) { const double wn = wk + inc; mn = wn < cur_min ? wn : cur_min; }
                    else { scan = true; mn = 1.7976931348623157e308; }
                    if (scan) {
                        for (int b = sb - 1 + lane; b < eb; b += 64) {
                            double w = sweight[b];
                            if (b == k - 1) w = w + inc;
                            mn = w < mn ? w : mn;
                        }
                        mn = readlane_f64(dpp_wave_min(mn), 63);
                    }
                    wave_sync();
                    if (mn != 0.0) {
                        for (int b = sb - 1 + lane; b < eb; b += 64) {
                            double w = sweight[b];
                            if (b == k - 1) w = w + inc;
                            sweight[b] = w - mn;
                        }
                        if (lane == 0 && !k_in) sweight[k - 1] = sweight[k - 1] + inc;
                        wave_sync();
                        bool sm = true;                                           // (the window was re-gauged: is every weight still below 2^20?)
                        for (int b = lane; b < nbins; b += 64) sm = sm && fabs(sweight[b]) < 1048576.0;
                        const bool allsm = __ballot(!sm) == 0ull;
                        if (lane == 0) C.tab_small = allsm ? 1 : 0;
                    } else if (lane == 0) {
                        const double w_upd = k_in ? (wk + inc) - mn : wk + inc;
                        sweight[k - 1] = w_upd;
                        if (!(fabs(w_upd) < 1048576.0)) C.tab_small = 0;
                    }
                    if (lane == 0) {
                        C.cur_min = 0.0;                                          // the lowest bin of the window is now exactly mn - mn
                        C.gauge = C.gauge + mn;
                        shist[k - 1] = shist[k - 1] + visit;
                    }
                }
                wave_sync();
            }
        }
        [[maybe_unused]] const unsigned long long sw_p1 = MW_SW_NOW();
        MW_SW_ACC(13, sw_p1 - sw_p0);
        if (do_switch) {
            // new_eta - old_eta of the switch (:1557-1558) = eta_weight(ls_mu) - eta_weight(ls_mu) with the weights as they are NOW,
            // added to the energy terms ONE AFTER THE OTHER (:1561-1563): (x + eta) - eta.  For a modest eta that is x to a rounding
            // residue below 1.2e-10 (|eta| < 2^20; less than two exp implementations differ by in exp(-x) once |x| > 1e-6) and the
            // exponential computed ahead (cmp_sw) stands.  A large eta absorbs x -- above all the hard wall of a walker OUTSIDE its
            // order-parameter range, eta = huge(1.0_dp): the difference is then exactly 0 and the reference always switches.  There
            // the expression is evaluated as the reference writes it.
            double cmp = cmp_sw, ew = 0.0;
            const bool wall = C.mg.in_window && (C.ls_mu < C.mg.mu_lo || C.ls_mu > C.mg.mu_hi);
            bool plain;
            if (C.samplerun || !C.record) { ew = eta_fin; plain = !wall && fabs(ew) < 1048576.0; }
            else {
                const int k = kc < 2 ? 2 : (kc > nbins - 1 ? nbins - 1 : kc);
                const double wa = fabs(sweight[k - 2]), wb = fabs(sweight[k - 1]), wc = fabs(sweight[k]);
                plain = !wall && wa < 1048576.0 && wb < 1048576.0 && wc < 1048576.0;   // (interpolation stays between its nodes' weights)
                if (!plain) ew = lane_eta(C.mg, sweight, smub, sbw, C.ls_mu, kc);
            }
            if (!plain) {
                double lesh;
                double d = switch_dk_terms(C.men0, C.men1, C.ls, lesh);
                d = d + ew;
                d = d - ew;
                d = d + lesh;
                cmp = exp_any(-d);                                                // (NaN for weights that are not finite: no switch)
            }
            cmp = cmp > 1.0 ? 1.0 : cmp;
            if (u6 < cmp) {
                const double V1 = svol[0], V2 = svol[1];
                double mu = (C.men0 + C.pressure * V1) - (C.men1 + C.pressure * V2);            // :1581-1583
                mu = mu - C.dref;                                                               // :1584 (leshift)
                mu = mu * C.beta - (double)N * C.lgv12;
                sw = 1;
                if (lane == 0) { C.ls_mu = mu; C.ls = 3 - C.ls; C.nsw = C.nsw + 1; C.k_valid = 0; }
                wave_sync();
            }
        }
        MW_SW_ACC(14, MW_SW_NOW() - sw_p1);
        return sw;
    };

    int ls = C.ls;                               // the active lattice, followed by every wavefront
    [[maybe_unused]] const unsigned long long sw_k0 = MW_SW_NOW();
#ifdef MW_SWEEP_STAMPS
    const unsigned long long sw_w0 = wall_clock64();
#endif

    // ---- the decisions on a round of translations (wavefront 0), mc_moves.F90:1090-1209 + mc_update_wl_bins + mc_lattice_switch ----
    // One chain's speed is the length of this routine, and what a LONE wavefront pays for is not arithmetic (tools/lat_probe.hip on
    // gfx950: a dependent v_fma_f64 6 cycles) but every trip through the scalar unit -- a value read from a lane and used again 25-30
    // cycles, a compare that feeds a branch 60, a taken branch 20-25, a lane-0-only region 22 -- and every LDS round trip, 60-80.
    // With the walker's state in LDS and every decision computed from scratch in scalar style a decision took 2.6 us where its
    // evaluation took 1.2 (profiles/r04a_sweep_stamps.json).  So the round is decided in lanes, in three pieces:
    //  * PRE-PHASE, once per round: everything that follows from the evaluated energies and the state at the round's start ALONE,
    //    under the assumption that holds for as long as the round goes on anyway on small boxes -- the earlier moves of the round are
    //    rejected, no lattice switch.  With a the round's first undecided move: lane 2a holds the current order parameter, lane 2s+1
    //    the value move s takes it to if accepted, lane 2s+2 the value its rejection restores (the chain mu -> (mu + d) - d, a handful
    //    of additions every lane runs for itself); ONE instruction stream gives all their bins (the engine's own log), the
    //    interpolation of eta_weight reduced to {two table indices, one slope factor}, the reciprocal bin width of the Wang-Landau
    //    visit, and the exponential of the lattice switch that would follow the move.
    //  * the RUN (`run_of_moves`), once per round in the normal case: five moves in six are rejected and change nothing but the
    //    tables -- the visited bin's weight and histogram entry -- so the table move s will find, IF the moves before it are such
    //    rejections, follows from the pre-phase: every lane adds the earlier moves' increments to the weights it touches in the
    //    chain's order (the very additions the moves would make), and every move's eta_weight pair, acceptance exponential and
    //    switch test come out of one straight-line pass -- no loop, no value carried through a scalar.  The run up to the first
    //    move that is accepted, switches lattice, needs the general routine (a bin at the window's minimum, a re-gauge, a wall, a
    //    large weight) or whose evaluation no longer stands is committed at once, an accepted move of the routine kind with it.
    //  * the SERIAL step, for the move the run stopped at when it is not of the routine kind, and for every move of the run options
    //    the run does not cover (sample runs, 'dd' windows, MINU, Swetnam's increment): the reference's sequence as written.
    // After an ACCEPTED move that leaves the round standing (large boxes) the pre-phase is run again for the moves that remain.
    // eta_weight's interpolation is evaluated as  w(base) + [(mu - mu_bin(base)) 2 / (binwidth(hi) + binwidth(lo))] (w(hi) - w(lo)):
    // the reference's  (mu - mu_bin) * (2 (w(hi) - w(lo)) / (bw + bw))  with the division taken out of the serial step (1e-16 relative).
    // Floating-point contraction is OFF in here: a product that the run adds to a weight in one place and the serial step in another
    // must round the same way wherever it is inlined (look-ahead = the sequential chain, bit for bit).
    auto wave_fence = [&]() {                    // orders this wavefront's own LDS traffic for the compiler; LDS serves a wavefront in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    __shared__ int bx_k[SPEC + 1];                // the run's exchange: bin, weight increment, histogram increment of every move's rejection
    __shared__ double bx_inc[SPEC + 1], bx_vis[SPEC + 1], bx_dump[2];
#ifdef MW_RUN_SPEC1        // A/B only: the straight-line pass of the look-ahead builds also for one move at a time
    constexpr bool kRunPass = true;
#else
    constexpr bool kRunPass = SPEC > 1;
#endif
    auto decide_round = [&](int ntr, int ub, const double* U0, int mvbase) {
#pragma clang fp contract(off)
        constexpr double kHuge = 1.7976931348623157e308;                         // huge(1.0_dp)
        // (bit 31 of an evaluation's dependence word -- the moment path declined it -- ends the round BEFORE that move; the next round
        //  finds it in slot 0 and decides nothing else)
        unsigned accmask = 0x80000000u;
        if constexpr (SPEC > 1) { if (((scm[0] | (NLAT == 2 ? scm[1] : 0u)) >> 31) != 0u) ntr = 1; }
        int nvalid = 0;
        // the walker's state, in registers for the length of the round
        double bk0 = C.men0, bk1 = L == 2 ? C.men1 : 0.0;                         // :1013
        int ls_c = C.ls;
        unsigned long long acc_d = 0ull;
        if constexpr (L == 1) {
            // one lattice: a move's acceptance depends on its own energies only -- every exponential of the round in one pass
            const int sl = lane < SPEC ? lane : 0;
            const double eo = sx[2 * sl], en = sx[2 * sl + 1];
            const double beta = C.beta;
            const double pex = exp_any(-(beta * (en - eo)));                      // :1106,1145
            for (int s = 0; s < ntr; ++s) {
                if (s > 0 && (scm[s] & accmask) != 0u) break;                     // that evaluation no longer stands: the next round starts with it
                const double* U = U0 + 8 * s;
                double pacc = readlane_f64(pex, s);
                pacc = pacc > 1.0 ? 1.0 : pacc;
                const bool ok = U[5] < pacc;                                      // (false for NaN)
                const double eo_s = readlane_f64(eo, s), en_s = readlane_f64(en, s);
                if (ok) { accmask |= 1u << s; ++acc_d; bk0 = (bk0 - eo_s) + en_s; }     // :1016,1150-1170
                if (mvlog && lane == 0) {
                    const int im = (int)__double_as_longlong(smv[(ub + s) * MVS + 3]);
                    double* q = mvlog + ((size_t)blockIdx.x * nmoves + (size_t)(mvbase + s)) * 8;
                    q[0] = (double)im; q[1] = ok ? 1.0 : 0.0; q[2] = eo_s; q[3] = en_s; q[4] = 0.0; q[5] = 0.0; q[6] = C.ls_mu; q[7] = beta * (en_s - eo_s);
                }
                ++nvalid;
            }
            if (lane == 0) { C.men0 = bk0; C.acc = C.acc + acc_d; }
        } else {
            double mu_c = C.ls_mu, cur_min = C.cur_min, wlf = C.wlf, gauge = C.gauge;   // (gauge: the minima subtracted so far, :1682-1685, summed move by move)
            int k_c = C.k_cur, k_val = C.k_valid, in_win = C.mg.in_window, cyc = C.cyc, within = C.within, flag_d = 0;
            unsigned long long nsw_d = 0ull;
            const double beta = C.beta, lun = C.log_unbiased_norm, avbw = C.av_binwidth;
            const int record = C.record, samplerun = C.samplerun, always_switch = C.always_switch, dd = C.dd, minu = C.minu, swet = C.swetnam;
            const int nb = nbins, sb = C.mg.start_bin, eb = C.mg.end_bin, interp = C.mg.eta_interp;
            const double mu_lo = C.mg.mu_lo, mu_hi = C.mg.mu_hi;
            // per lane: what the pre-phase leaves for the run and the serial steps (lane 2a: the current value; 2s+1 / 2s+2: move s accepted / rejected)
            double dEb = 0.0, mn0l = 0.0, mn1l = 0.0, mul = 0.0, c_m = 0.0, c_rb = 1.0, exs = 0.0;
            int kl = 0, kc = 1, c_b = 1, c_hi = 2;
            bool c_out = false;
            auto all_weights_small = [&]() {
                bool sm = true;
                for (int b = lane; b < nb; b += 64) sm = sm && fabs(sweight[b]) < 1048576.0;
                return __ballot(!sm) == 0ull;
            };
            bool tab_small = C.tab_small != 0;     // |weight| < 2^20 in every bin (the lattice switch's shortcut below): kept up with every update
            int cur_lane = 0;
            const int slot_l = lane >= 1 && lane <= 2 * SPEC ? (lane - 1) >> 1 : 0;      // this lane's move of the round
            const bool newl = (lane & 1) != 0;                                            // ... its "accepted" (odd lane) or "rejected" candidate
            auto prephase = [&](int a) {
                [[maybe_unused]] const unsigned long long t0 = MW_SW_NOW();
                // the chain of order parameters, every lane for itself (uniform reads of the round's energies: no value travels between lanes)
                double mu_run = mu_c;
                mul = mu_c;
#pragma unroll
                for (int j = 0; j < SPEC; ++j) {
                    const double e0o = sx[2 * (j * NLAT)], e0n = sx[2 * (j * NLAT) + 1], e1o = sx[2 * (j * NLAT + 1)], e1n = sx[2 * (j * NLAT + 1) + 1];
                    const double dj = ((e0n - e0o) - (e1n - e1o)) * beta;         // :1114 ... and what :1192 takes off again
                    const double mu_n = mu_run + dj, mu_r = mu_n - dj;
                    const bool act = j >= a && j < ntr;
                    mul = (act && lane == 2 * j + 1) ? mu_n : ((act && lane == 2 * j + 2) ? mu_r : mul);
                    mu_run = act ? mu_r : mu_run;
                }
                const double eo0 = sx[2 * (slot_l * NLAT)], en0 = sx[2 * (slot_l * NLAT) + 1];
                const double eo1 = sx[2 * (slot_l * NLAT + 1)], en1 = sx[2 * (slot_l * NLAT + 1) + 1];
                const double dE0 = en0 - eo0, dE1 = en1 - eo1;                    // :1090
                dEb = (ls_c == 1 ? dE0 : dE1) * beta;
                mn0l = (bk0 - eo0) + en0; mn1l = (bk1 - eo1) + en1;               // :1016,1087
#ifdef MW_ABL_D_NOBIN
                int kq = nb / 2 + 1 + (lane & 3);
#else
                int kq = lane_mu_to_bin(C.mg, mul);                               // :1112-1116
#endif
                kq = (k_val && lane == 2 * a) ? k_c : kq;                         // (the bin of the current value is carried)
                kl = kq;
                kc = kq < 1 ? 1 : (kq > nb ? nb : kq);                            // eta_weight (:893-964), see lane_eta
                const bool up = (kc == sb) || (kc != eb && mul > smub[kc - 1]);
                int hi = up ? kc + 1 : kc;
                hi = hi > nb ? nb : (hi < 2 ? 2 : hi);
                const int base = (up || kc == eb) ? kc : (kc > 1 ? kc - 1 : 1);
                c_hi = hi;
                c_b = interp ? base : kc;
                c_m = (mul - smub[base - 1]) * (2.0 * fast_rcp(sbw[hi - 1] + sbw[hi - 2]));
                c_out = mul < mu_lo || mul > mu_hi;
                c_rb = fast_rcp(sbw[kc - 1]);                                     // (reciprocal + one correction: the divisions of :1618,1680 to an ulp)
#ifdef MW_ABL_D_NOSWEXP
                if (false) {
#else
                if (always_switch) {                                              // mc_lattice_switch's exponential after an accepted / a rejected move
#endif
                    const double dk = newl ? switch_dk(mn0l, mn1l, ls_c) : switch_dk(bk0, bk1, ls_c);
                    // (one move at a time: every move takes the serial step, whose own exponential stream has lanes to spare -- the
                    //  argument waits there; with look-ahead the serial step is the exception and the exponential is taken here)
                    exs = !kRunPass ? -dk : exp_any(-dk);
                }
                cur_lane = 2 * a;
                MW_SW_ACC(10, MW_SW_NOW() - t0);
            };
            // eta_weight of a candidate from the three weights it touches (the serial step and the run: one expression)
            auto eta_of = [&](double w_b, double w_h, double w_l) {
                double val = interp ? __builtin_fma(c_m, w_h - w_l, w_b) : w_b;
                val = c_out ? kHuge : val;
                return in_win ? val : 0.0;                                        // (:913, G12: a 'dd' walker outside its window carries no weight)
            };
            // THE RUN (see the head of this routine).  Returns the first move it has NOT dealt with; `serial` says whether that move needs
            // the serial step (otherwise the caller's own checks end the round or the run has simply reached the round's end).
            // (Look-ahead builds only -- the handful of walkers whose speed is one chain's.  The builds that take one move at a time serve
            //  thousands of walkers and are bound by the NUMBER of vector instructions issued: there the serial step, which skips what a
            //  move does not need, issues ~100 fewer per move than the straight-line pass.  Both produce the same numbers, bit for bit.)
            const bool run_on = kRunPass && !samplerun && !swet && !dd && !minu;
            bool need_pre = true;
            auto run_of_moves = [&](int a, bool& serial) -> int {
                [[maybe_unused]] const unsigned long long t0 = MW_SW_NOW();
                const bool cand = lane >= 1 && lane <= 2 * SPEC && slot_l >= a && slot_l < ntr;
                const bool kval = kl >= 1 && kl <= nb, kin = kl >= sb && kl <= eb;
                const double incl = (avbw * wlf) * c_rb, visl = avbw * c_rb;      // :1680, :1618
                // every rejection's visit, for the lanes of the later moves (the other lanes write a dump entry: no lane-masked region)
                const int xi = (cand && !newl) ? slot_l : SPEC;
                bx_k[xi] = kval ? kc : 0; bx_inc[xi] = incl; bx_vis[xi] = visl;
                double w_b = sweight[c_b - 1], w_h = sweight[c_hi - 1], w_l = sweight[c_hi - 2];
                double w_k = 0.0, h_k = 0.0;
                if (record) { w_k = sweight[kc - 1]; h_k = shist[kc - 1]; }
                const double u5 = U0[8 * slot_l + 5], u6 = U0[8 * slot_l + 6];
                const unsigned depl = slot_l == 0 ? 0u : (scm[slot_l * NLAT] | scm[slot_l * NLAT + 1]) & accmask;   // (slot 0 depends on nothing; its bit 31: see above)
                wave_fence();
                // the rejections before this candidate, in the chain's order: `e` those before its eta_weight is looked up (for a
                // "rejected" lane that includes its own move's: it is the NEXT move's current value), `o` those before its own visit
                bool dup = false;                                                 // a later rejection of the run visits this lane's bin too
                int kx[SPEC];
                if (record) {
#pragma unroll
                    for (int j = 0; j < SPEC; ++j) {
                        kx[j] = bx_k[j];
                        const double ij = bx_inc[j], vj = bx_vis[j];
                        const bool inr = cand && j >= a;
                        const bool e = inr && (newl ? j < slot_l : j <= slot_l), o = inr && j < slot_l;
                        w_b = (e && kx[j] == c_b) ? w_b + ij : w_b;
                        w_h = (e && kx[j] == c_hi) ? w_h + ij : w_h;
                        w_l = (e && kx[j] == c_hi - 1) ? w_l + ij : w_l;
                        const bool hit = o && kx[j] == kc;
                        w_k = hit ? w_k + ij : w_k;
                        h_k = hit ? h_k + vj : h_k;
                    }
                }
                // is this candidate's own visit the routine one?  the bin gains its increment, the window's minimum stays 0 (:1680-1685),
                // no weight grows large
                const double wn = w_k + incl;
                const bool simple_l = !record || !kval || ((!kin || (w_k > 0.0 && wn >= 0.0)) && fabs(wn) < 1048576.0);
                const double val = eta_of(w_b, w_h, w_l);
                // what a move's new value replaces sits one lane down: the value the previous rejection restored, or the current one
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(val), 0x138, 0xf, 0xf, false);      // wave_shr:1
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(val), 0x138, 0xf, 0xf, false);
                const double diffkT = dEb + val - __hiloint2double(hi, lo);      // :1116 (odd lanes)
                const double ex = exp_any(-diffkT);
                const double pacc = ex > 1.0 ? 1.0 : ex;
                const bool okl = u5 < pacc;                                       // :1145-1146
                bool swl = false, hard = false;                                   // the switch attempt after the move
                if (always_switch) {
                    const bool wall = in_win && c_out;
                    const bool plain = !wall && (record || fabs(val) < 1048576.0);        // (record: every weight is small, and stays so)
                    hard = !plain;
                    const double cmp = exs > 1.0 ? 1.0 : exs;
                    swl = plain && u6 < cmp;
                }
                const bool routine = simple_l && !swl && !hard;
                // the first move of the run that does not go through as a routine rejection
                const bool stopl = cand && (newl ? (depl != 0u || okl) : !routine);
                const unsigned long long stops = __ballot(stopl);
                const unsigned long long accs = __ballot(cand && newl && depl == 0u && okl && routine);
                const unsigned long long deps = __ballot(cand && newl && depl != 0u);
                const int lf = stops ? __ffsll((long long)stops) - 1 : 2 * ntr + 1;      // its lane
                const int f = (lf - 1) >> 1;
                const bool at_new = (lf & 1) != 0 && f < ntr;
                const bool acc_f = at_new && ((accs >> lf) & 1ull) != 0ull;      // ... an accepted move of the routine kind: committed here too
                serial = f < ntr && !acc_f && !(at_new && ((deps >> lf) & 1ull) != 0ull);
                // commit: the rejections a .. f-1 (a bin visited twice keeps the later value), then the accepted move's visit
                if (record) {
#pragma unroll
                    for (int j = 0; j < SPEC; ++j) dup = dup || (j > slot_l && j < f && kx[j] == kc);
                    const bool wr = cand && !newl && slot_l < f && kval && !dup;
                    double* pw = wr ? sweight + (kc - 1) : bx_dump;
                    double* ph = wr ? shist + (kc - 1) : bx_dump + 1;
                    *pw = wn; *ph = h_k + visl;
                    if (acc_f) {
                        wave_fence();
                        const bool wa = lane == lf && kval;
                        double* qw = wa ? sweight + (kc - 1) : bx_dump;
                        double* qh = wa ? shist + (kc - 1) : bx_dump + 1;
                        *qw = wn; *qh = h_k + visl;
                    }
                }
                if (mvlog) {
                    for (int j = a; j < (acc_f ? f + 1 : f); ++j) {
                        const int lj = (acc_f && j == f) ? 2 * j + 1 : 2 * j + 2;
                        const double dk_j = readlane_f64(diffkT, 2 * j + 1), mu_j = readlane_f64(mul, lj);
                        if (lane == 0) {
                            const int im = (int)__double_as_longlong(smv[(ub + j) * MVS + 3]);
                            double* q = mvlog + ((size_t)blockIdx.x * nmoves + (size_t)(mvbase + j)) * 8;
                            q[0] = (double)im; q[1] = (acc_f && j == f) ? 1.0 : 0.0;
                            q[2] = sx[2 * (j * NLAT)]; q[3] = sx[2 * (j * NLAT) + 1]; q[4] = sx[2 * (j * NLAT + 1)]; q[5] = sx[2 * (j * NLAT + 1) + 1];
                            q[6] = mu_j; q[7] = dk_j;
                        }
                    }
                }
                const int lz = acc_f ? lf : 2 * f;                                // the candidate the chain has reached
                if (acc_f || f > a) {
                    mu_c = readlane_f64(mul, lz); k_c = __builtin_amdgcn_readlane(kl, lz); k_val = 1; cur_lane = lz;
                }
                if (acc_f) {                                                      // :1150-1170
                    bk0 = readlane_f64(mn0l, lf); bk1 = readlane_f64(mn1l, lf);
                    ++acc_d; accmask |= 1u << f; need_pre = true;
                }
                wave_fence();
                MW_SW_ACC(12, MW_SW_NOW() - t0); MW_SW_ACC(9, (unsigned long long)((acc_f ? f + 1 : f) - a)); MW_SW_ACC(16, 1ull);
                return acc_f ? f + 1 : f;
            };
            int s = 0;
            while (s < ntr) {                                                     // the decisions, in the chain's order
                // (an evaluation that no longer stands ends the round: the next one starts with that move)
                if (s > 0 && (((scm[s * NLAT] | scm[s * NLAT + 1]) & accmask) != 0u || ls_c != ls)) break;
                if (need_pre) { prephase(s); need_pre = false; }
                if constexpr (kRunPass) {
                    if (run_on && (!record || (cur_min == 0.0 && tab_small))) {
                        bool serial = false;
                        const int f = run_of_moves(s, serial);
                        nvalid += f - s; s = f;
                        if (!serial) continue;                                    // (the loop's head ends the round, or goes on after an accepted move)
                    }
                }
                [[maybe_unused]] const unsigned long long t1 = MW_SW_NOW();
                const double* U = U0 + 8 * s;
                if (dd && within == 0) {                                          // top of a cycle: the equilibration check of mc_cycle (:181-210)
                    if (cyc < C.eq_cycles) in_win = (mu_c > mu_lo && mu_c < mu_hi) ? 1 : 0;
                    else if (cyc == C.eq_cycles) { if (!in_win) flag_d |= 2; }    // "Not all walkers have reached their designated window"
                    else in_win = 1;                                              // a restart
                    if (lane == 0) C.mg.in_window = in_win;
                    wave_fence();
                }
                const bool do_switch = always_switch && !(dd && cyc < C.eq_cycles);     // (:243-248: not while a 'dd' run equilibrates)
                // the weights every candidate touches, as the earlier moves of the round have left them: one round trip
                const double w_b = sweight[c_b - 1], w_h = sweight[c_hi - 1], w_l = sweight[c_hi - 2], w_k = sweight[kc - 1];
                const double val = eta_of(w_b, w_h, w_l);
                const int ln = 2 * s + 1, lr = 2 * s + 2;
                const double eta_new = readlane_f64(val, ln), eta_rev = readlane_f64(val, lr), eta_old = readlane_f64(val, cur_lane);
                double diffkT = readlane_f64(dEb, ln) + eta_new - eta_old;        // :1116
                int minu_ls = ls_c;
                double cmpA_minu = 0.0;
                if (minu) {                                                       // :1119-1140
                    const double mn0 = readlane_f64(mn0l, ln), mn1 = readlane_f64(mn1l, ln);
                    minu_ls = dev_minu_branch(C, ls_c, mn0, mn1, svol[0], svol[1], ls_c == 1 ? bk0 : bk1, ls_c == 1 ? svol[0] : svol[1],
                                              C.npt != 0, N, eta_new, eta_old, diffkT);
                    if (minu_ls != ls_c && always_switch) cmpA_minu = exp_any(-switch_dk(mn0, mn1, minu_ls));
                }
                // lane 0: the acceptance; lanes 61, 62: the unbiased histogram's factor after an accepted / a rejected move (:1627-1629);
                // one move at a time: lanes 1, 2 the lattice switch's exponentials, whose arguments the pre-phase left there
                double xarg = lane == 0 ? -diffkT : (lane == 61 ? eta_new - lun : eta_rev - lun);
                if constexpr (!kRunPass) xarg = (lane == 1 || lane == 2) ? exs : xarg;
#ifdef MW_ABL_D_NOEXP
                const double ex = 0.1 + 0.01 * xarg;
#else
                const double ex = exp_any(xarg);
#endif
                if constexpr (!kRunPass) exs = ex;
                double pacc = readlane_f64(ex, 0);
                pacc = pacc > 1.0 ? 1.0 : pacc;
                const bool ok = U[5] < pacc;                                      // :1145-1146 (false for NaN)
                const int cl = ok ? ln : lr;
                double eta_fin, cmp_sw = 0.0, ufac = 0.0;
                if (ok) {                                                         // :1150-1170
                    ++acc_d; accmask |= 1u << s; need_pre = true;
                    if (do_switch) cmp_sw = minu_ls != ls_c ? cmpA_minu : readlane_f64(exs, ln);
                    ls_c = minu_ls;
                    bk0 = readlane_f64(mn0l, ln); bk1 = readlane_f64(mn1l, ln);   // :1016,1087
                    eta_fin = eta_new;
                    if (samplerun) ufac = readlane_f64(ex, 61);
                } else {                                                          // :1182-1195
                    if (do_switch) cmp_sw = readlane_f64(exs, lr);
                    eta_fin = eta_rev;
                    if (samplerun) ufac = readlane_f64(ex, 62);
                }
                mu_c = readlane_f64(mul, cl);
                k_c = __builtin_amdgcn_readlane(kl, cl); k_val = 1;
                cur_lane = cl;
                [[maybe_unused]] const unsigned long long t2 = MW_SW_NOW();
                MW_SW_ACC(11, t2 - t1);
                // ---- mc_update_wl_bins (:1597-1689) on the walker's tables --------------------------------------------------
#ifdef MW_ABL_D_NOWL
                if (false) {
#else
                if (record) {
#endif
                    const int k = k_c;
                    if (k >= 1 && k <= nb) {
                        const double rbk = readlane_f64(c_rb, cl);
                        const double visit = avbw * rbk;                                          // :1618
                        if (samplerun) {
                            if (lane == 0) {
                                shist[k - 1] = shist[k - 1] + visit;                              // :1621
                                suhist[k - 1] = suhist[k - 1] + visit * ufac;                     // :1627-1629
                            }
                        } else {
                            double inc = (avbw * wlf) * rbk;                                      // :1680
                            if (swet) {                                           // :1636-1653
                                const double sumh = C.sumh + 1.0;
                                double a2 = 0.0;
                                const double span = C.mu_max - C.mu_min - 1.0;
                                int lane_s = lane;
                                asm volatile("" : "+v"(lane_s));
                                for (int b = lane_s; b < nb; b += 64) {
                                    const double hb = shist[b] + (b == k - 1 ? visit : 0.0);      // this move's visit is already counted (:1621)
                                    const double dev = hb * sbw[b] / sumh - sbw[b] / span;
                                    a2 += dev * dev;
                                }
                                a2 = readlane_f64(dpp_wave_sum(a2), 63);
                                double f = sqrt(a2 / (double)nb);
                                f = log(f) * C.wl_alpha * (double)nb;
                                wlf = f < C.orig_wl_factor ? f : C.orig_wl_factor;
                                if (lane == 0) C.sumh = sumh;
                                inc = (avbw * wlf) * rbk;
                            }
                            // weight(k) += av_binwidth*wl_factor/binwidth(k) -- whichever bin k is (:1680); then the minimum over the
                            // walker's window is subtracted inside the window (:1682-1685; with 'dd' windows k may lie outside).
                            // The minimum is TRACKED: it is 0 after the first update, and it only moves when the visited bin was
                            // (one of) the lowest -- then, and only then, the window is scanned.
                            const double wk = readlane_f64(w_k, cl);
                            const bool k_in = k >= sb && k <= eb;
                            double mn;
                            bool scan = false;
                            if (!k_in) mn = cur_min;
                            else if (wk > cur_min) { const double wn = wk + inc; mn = wn < cur_min ? wn : cur_min; }
                            else { scan = true; mn = kHuge; }
                            if (scan) {
                                for (int b = sb - 1 + lane; b < eb; b += 64) {
                                    double w = sweight[b];
                                    if (b == k - 1) w = w + inc;
                                    mn = w < mn ? w : mn;
                                }
                                mn = readlane_f64(dpp_wave_min(mn), 63);
                            }
                            if (mn != 0.0) {
                                wave_fence();
                                for (int b = sb - 1 + lane; b < eb; b += 64) {
                                    double w = sweight[b];
                                    if (b == k - 1) w = w + inc;
                                    sweight[b] = w - mn;
                                }
                                if (lane == 0 && !k_in) sweight[k - 1] = sweight[k - 1] + inc;
                                wave_fence();
                                tab_small = all_weights_small();                  // (the window was re-gauged: rare)
                            } else {
                                const double w_upd = k_in ? (wk + inc) - mn : wk + inc;
                                if (lane == 0) sweight[k - 1] = w_upd;
                                tab_small = tab_small && fabs(w_upd) < 1048576.0;
                            }
                            cur_min = 0.0;                                        // the lowest bin of the window is now exactly mn - mn
                            gauge = gauge + mn;
                            if (lane == 0) shist[k - 1] = shist[k - 1] + visit;
                        }
                        wave_fence();
                    }
                }
                [[maybe_unused]] const unsigned long long t3 = MW_SW_NOW();
                MW_SW_ACC(13, t3 - t2);
                // ---- one mc_lattice_switch attempt (:1536-1594) -----------------------------------------------------------------
                int sw = 0;
#ifdef MW_ABL_D_NOSW
                if (false) {
#else
                if (do_switch) {
#endif
                    // new_eta - old_eta of the switch (:1557-1558) = eta_weight(ls_mu) - eta_weight(ls_mu) with the weights as they are NOW,
                    // added to the energy terms ONE AFTER THE OTHER (:1561-1563): (x + eta) - eta.  For a modest eta that is x to a rounding
                    // residue below 1.2e-10 (|eta| < 2^20) and the exponential computed ahead (cmp_sw) stands.  A large eta absorbs x --
                    // above all the hard wall of a walker OUTSIDE its order-parameter range, eta = huge(1.0_dp): the difference is then
                    // exactly 0 and the reference always switches.  There the expression is evaluated as the reference writes it.
                    double cmp = cmp_sw, ew = 0.0;
                    const bool wall = in_win && (mu_c < mu_lo || mu_c > mu_hi);
                    bool plain;
                    if (samplerun || !record) { ew = eta_fin; plain = !wall && fabs(ew) < 1048576.0; }
                    else if (tab_small) plain = !wall;                            // (every weight is small: so are the three the interpolation can touch)
                    else {                                                        // a large weight somewhere: the three around the bin, as they are now
                        const int k = k_c < 2 ? 2 : (k_c > nb - 1 ? nb - 1 : k_c);
                        plain = !wall && fabs(sweight[k - 2]) < 1048576.0 && fabs(sweight[k - 1]) < 1048576.0 && fabs(sweight[k]) < 1048576.0;   // (interpolation stays between its nodes' weights)
                    }
                    if (!plain) {
                        if (!(samplerun || !record)) ew = lane_eta(C.mg, sweight, smub, sbw, mu_c, k_c);
                        double lesh;
                        double d = switch_dk_terms(bk0, bk1, ls_c, lesh);
                        d = d + ew;
                        d = d - ew;
                        d = d + lesh;
                        cmp = exp_any(-d);                                        // (NaN for weights that are not finite: no switch)
                    }
                    cmp = cmp > 1.0 ? 1.0 : cmp;
                    if (U[6] < cmp) {
                        const double V1 = svol[0], V2 = svol[1];
                        double mu = (bk0 + C.pressure * V1) - (bk1 + C.pressure * V2);            // :1581-1583
                        mu = mu - C.dref;                                                         // :1584 (leshift)
                        mu = mu * beta - (double)N * C.lgv12;
                        sw = 1;
                        mu_c = mu; ls_c = 3 - ls_c; ++nsw_d; k_val = 0;
                    }
                }
                if (lane == 0) {
                    if (mvlog) {
                        const int im = (int)__double_as_longlong(smv[(ub + s) * MVS + 3]);
                        double* q = mvlog + ((size_t)blockIdx.x * nmoves + (size_t)(mvbase + s)) * 8;
                        q[0] = (double)im; q[1] = (ok ? 1.0 : 0.0) + 2.0 * sw;
                        q[2] = sx[2 * (s * NLAT)]; q[3] = sx[2 * (s * NLAT) + 1]; q[4] = sx[2 * (s * NLAT + 1)]; q[5] = sx[2 * (s * NLAT + 1) + 1];
                        q[6] = mu_c; q[7] = diffkT;
                    }
                }
                if (dd) { const int w = within + 1; if (w == N) { within = 0; cyc = cyc + 1; } else within = w; }
                MW_SW_ACC(14, MW_SW_NOW() - t3);
                ++nvalid; ++s;
            }
            if (lane == 0) {
                C.men0 = bk0; C.men1 = bk1; C.ls_mu = mu_c; C.cur_min = cur_min; C.wlf = wlf; C.gauge = gauge;
                C.k_cur = k_c; C.k_valid = k_val; C.acc = C.acc + acc_d; C.nsw = C.nsw + nsw_d; C.tab_small = tab_small ? 1 : 0;
                if (flag_d) C.flag = C.flag | flag_d;
                if (dd) { C.cyc = cyc; C.within = within; }
            }
        }
        if (lane == 0) {
            C.ls = ls_c;
            sdec[0] = nvalid; sdec[1] = ls_c;
#pragma unroll
            for (int s = 0; s < SPEC; ++s) sdec[4 + s] = (int)((accmask >> s) & 1u);
        }
        wave_fence();
    };

    // THE MOMENT PATH of a translation (walkers entirely in LDS; move_energy_mom_wave): the i--j--k sums come from every molecule's
    // moments, kept in `wmom` (global memory, L2-resident: the walker's LDS is counted in bytes) -- made by the lattice's first
    // wavefront when sdec[2 + lattice] says so (the launch's start; an accepted move whose evaluation took another routine),
    // brought up to date by every accepted move (moments_commit), and for a volume move written for the TRIAL cell into the
    // other half of the buffer, which an accepted move makes the current one (msel, the same in every wavefront).
    // Walkers in global memory (N > 64), translations only: the moments are the engine's own array (`wmom` = its part for this
    // launch's boxes, [box][N][kMomStride]: made by the full-box kernel before the launch where they are not current, kept current here
    // and by nobody else -- mw_sweep_translation_launch); no trial set, no start-up pass.
    constexpr bool MOMBIG = !LDSLIST && !WITHVOL;
    const bool usemom = (LDSLIST || MOMBIG) && wmom != nullptr;
    int msel = 0;
    auto mom_of = [&](int sel, int l) -> double* {
        if constexpr (MOMLDS) return smom + (size_t)((sel * L + l) * N) * kMomStride;
        else if constexpr (LDSLIST) return wmom + ((((size_t)blockIdx.x * 2 + sel) * L + l) * N) * kMomStride;
        else return wmom + (((size_t)blockIdx.x * L + l) * N) * kMomStride;
    };

    constexpr int kUB = sweep_batch(WITHVOL, SPEC);
    int mv = 0, ubase = -kUB;                    // next move of the chain (counted within the launch); first move of the uniforms' window
    while (mv < nmoves) {
        if constexpr (LDSLIST || MOMBIG) {
            if (usemom && (SPEC > 1 ? (sdec[2] | (L == 2 ? sdec[3] : 0)) : sdec[2 + lat]) != 0) {       // (the same in every wavefront that waits below)
                if (slot == 0 && sdec[2 + lat] != 0) {
                    int lv = lane, bv = box0;
                    asm volatile("" : "+v"(lv), "+s"(bv));
                    VolCtx vm;
                    vm.pos_g = pos + (size_t)bv * N * 3; vm.spos = spos; vm.siv = siv; vm.sniv = sniv;
                    vm.queue = reinterpret_cast<uint32_t*>(ws) + lv; vm.N = N; vm.S = S; vm.ivcap = ivcap; vm.L = L;
                    vm.srow = LDSLIST ? srow : nullptr; vm.snn = LDSLIST ? snn : nullptr; vm.rstride = rstride;
                    vm.list_g = list + (size_t)bv * S * N; vm.order_g = order + (size_t)bv * N; vm.nns_g = nns + (size_t)bv * N;
                    vm.cmax_g = cmax + (size_t)bv * ((N + 63) >> 6);
                    vm.mom_trial = nullptr; vm.inmask = nullptr; vm.rec = nullptr;
                    (void)dev_wave_model_energy<LDSPOS>(vm, lat, lv, mom_of(msel, lat));       // (4096 molecules: half a millisecond -- for a move the moment path declined AND the chain accepted)
                }
                if (SPEC > 1) wg_sync<NW>(); else wave_sync();
                if (SPEC > 1) { if (tid == 0) { sdec[2] = 0; sdec[3] = 0; } }
                else if (lane == 0) sdec[2 + lat] = 0;
                if (SPEC > 1) wg_sync<NW>(); else wave_sync();
            }
        }
        [[maybe_unused]] const unsigned long long sw_u0 = MW_SW_NOW();
        if (mv + SPEC > ubase + kUB) {
            // the next kUB moves' random numbers: Philox call c of move m is thread 4 m + c (the same stream as the
            // oracle's mwo_move_uniforms: counter (move lo, move hi, walker, call), key = seed), then per move its
            // molecule (mc_moves.F90:1001-1002) and its displacement in the active lattice (:1021-1039)
            wg_sync<NW>();                                         // (the previous window has been consumed)
            ubase = mv;
            for (int c = tid; c < 4 * kUB; c += NTHR) {
                const unsigned long long m = move0 + (unsigned long long)(mv + (c >> 2));
                uint32_t ctr[4] = {(uint32_t)m, (uint32_t)(m >> 32), (uint32_t)wlk, (uint32_t)(c & 3)};
                philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
                suni[(c >> 2) * 8 + 2 * (c & 3)] = u53(ctr[0], ctr[1]);
                suni[(c >> 2) * 8 + 2 * (c & 3) + 1] = u53(ctr[2], ctr[3]);
            }
            wg_sync<NW>();
            if (tid < kUB && !(WITHVOL && !(suni[tid * 8 + 7] < C.transP))) {     // (translations: the move-type test of the rounds below)
                const double* u = suni + tid * 8;
                const double max_trans = C.max_trans;
                int im = (int)(u[0] * (double)N) + 1;                                     // :1001-1002
                im = im > N ? N : im;
                double x = 2.0 * u[1] - 1.0, y = 2.0 * u[2] - 1.0, z = 2.0 * u[3] - 1.0;  // :1021-1027
                const double norm = 1.0 / sqrt(x * x + y * y + z * z);                    // :1029
                x *= norm; y *= norm; z *= norm;
                const double r = u[4] * 2.0 - 1.0;                                        // :1035
                smv[tid * MVS] = x * max_trans * r; smv[tid * MVS + 1] = y * max_trans * r; smv[tid * MVS + 2] = z * max_trans * r;
                smv[tid * MVS + 3] = __longlong_as_double((long long)im);
            }
            wg_sync<NW>();
        }
        [[maybe_unused]] const unsigned long long sw_r0 = MW_SW_NOW();
        MW_SW_ACC(6, sw_r0 - sw_u0);
        const int ub = mv - ubase;                // the round's first move inside the window
        const double* U0 = suni + ub * 8;         // u0..u7 per move: molecule, direction x3, length, acceptance, lattice switch (:1576), move type (:226)
        // the round: the run of translations that starts here, at most one per slot (a volume move is a round of its own)
        int ntr = 0;
#pragma unroll
        for (int s = 0; s < SPEC; ++s)
            if (ntr == s && mv + s < nmoves && !(WITHVOL && !(U0[8 * s + 7] < C.transP))) ++ntr;

        if (ntr == 0) {                                                           // mc_moves.F90:232-235: a volume move
            [[maybe_unused]] const unsigned long long tvm = MW_SW_NOW();
            if constexpr (WITHVOL) {
                // Everything the volume move addresses is worked out HERE, from a lane number and a box number the compiler
                // cannot see through: hoisted out of the move loop, these loop-invariant addresses are what pushed the build
                // past 128 vector registers (three wavefronts per SIMD instead of four, for a branch taken once in N moves).
                int lv = lane, bv = box0;
                asm volatile("" : "+v"(lv), "+s"(bv));
                VolCtx vc;
                vc.pos_g = pos + (size_t)bv * N * 3; vc.spos = spos;
                vc.shmat = &shmat[0][0]; vc.srecip = &srecip[0][0]; vc.svol = svol; vc.sbk = &sbk[0][0]; vc.siv = siv; vc.sniv = sniv;
                vc.hmat_g = hmat + (size_t)bv * 9; vc.vol_g = volume + bv; vc.ivect_g = ivect + (size_t)bv * ivcap * 3;
                vc.nivect_g = nivect + bv; vc.list_g = list + (size_t)bv * S * N;
                vc.order_g = order + (size_t)bv * N; vc.nns_g = nns + (size_t)bv * N; vc.cmax_g = cmax + (size_t)bv * ((N + 63) >> 6);
                vc.queue = reinterpret_cast<uint32_t*>(ws) + lv; vc.N = N; vc.S = S; vc.ivcap = ivcap; vc.L = L;
                vc.srow = LDSLIST ? srow : nullptr; vc.snn = LDSLIST ? snn : nullptr; vc.rstride = rstride;
                vc.mom_trial = usemom ? mom_of(msel ^ 1, 0) : nullptr;
                vc.inmask = SPLIT ? reinterpret_cast<unsigned*>(smem_raw + lay.inmask) : nullptr;
                vc.rec = SPLIT ? reinterpret_cast<double*>(smem_raw + lay.rec) : nullptr;
                const double* U = U0;
                double diffkT = 0.0;
                bool do_switch = false;
                if (wv == 0) {
                    if (C.dd && C.within == 0) {                   // top of a cycle (:181-210), as in decide_trans
                        const int cyc = C.cyc;
                        if (lane == 0) {
                            if (cyc < C.eq_cycles) C.mg.in_window = (C.ls_mu > C.mg.mu_lo && C.ls_mu < C.mg.mu_hi) ? 1 : 0;
                            else if (cyc == C.eq_cycles) { if (!C.mg.in_window) C.flag = C.flag | 2; }
                            else C.mg.in_window = 1;
                        }
                        wave_sync();
                    }
                    do_switch = L == 2 && C.always_switch && !(C.dd && C.cyc < C.eq_cycles);
                }
                if (tid < L) svold[tid] = svol[tid];               // (volume_move_wg's first barrier orders this against the decision)
                auto decide = [&](double e0n, double e1n, int anybad) -> int {
                    const double Vo0 = svold[0], Vo1 = L == 2 ? svold[1] : 0.0;
                    // wavefront 0: mc_volume's acceptance (:1361-1410) and, on rejection, the restored order parameter (:1514-1530)
                    const double bk0 = C.men0, bk1 = C.men1;
                    const int ls0 = C.ls;
                    double ls_mu = C.ls_mu;
                    int okv = 0, lsn = ls0;
                    if (!anybad) {
                        const double Vn0 = svol[0], Vn1 = L == 2 ? svol[1] : 0.0;
                        const double dE = (ls0 == 1 ? e0n - bk0 : e1n - bk1);                                // :1361
                        const double Vls = ls0 == 1 ? Vn0 : Vn1, Vold = ls0 == 1 ? Vo0 : Vo1;
                        double old_eta = 0.0, new_eta = 0.0;
                        if (L == 2) {                                                                        // :1363-1371
                            double mu = (e0n + C.pressure * Vn0) - (e1n + C.pressure * Vn1);
                            mu = mu - C.dref;                                                                // :1371 (leshift)
                            mu = mu * C.beta - (double)N * fast_log_pos(Vn0 / Vn1);
                            const double mul = lane == 0 ? ls_mu : mu;
                            const double el = lane_eta(C.mg, sweight, smub, sbw, mul, lane_mu_to_bin(C.mg, mul));
                            old_eta = readlane_f64(el, 0); new_eta = readlane_f64(el, 1);
                            ls_mu = mu;
                        }
                        diffkT = C.beta * dE + new_eta - old_eta + C.beta * C.pressure * (Vls - Vold)
                                 - (double)N * fast_log_pos(Vls / Vold);                                     // :1381-1382
                        int minu_ls = ls0;
                        if (C.minu && L == 2)                                                                // :1385-1401
                            minu_ls = dev_minu_branch(C, ls0, e0n, e1n, Vn0, Vn1, ls0 == 1 ? bk0 : bk1, Vold, true, N,
                                                      new_eta, old_eta, diffkT);
                        double cmp = exp_any(-diffkT);
                        cmp = cmp > 1.0 ? 1.0 : cmp;
                        okv = U[3] < cmp ? 1 : 0;                                                            // :1410
                        if (okv) lsn = minu_ls;                                                              // :1426-1429
                    }
                    double m0 = e0n, m1 = e1n;
                    if (!okv) {
                        m0 = bk0; m1 = bk1;                                                                  // :1514
                        if (L == 2) {                                                                        // :1516-1520 (the OLD cells)
                            double mu = (m0 + C.pressure * Vo0) - (m1 + C.pressure * Vo1);
                            mu = mu - C.dref;                                                                // :1526 (leshift)
                            mu = mu * C.beta - (double)N * fast_log_pos(Vo0 / Vo1);
                            ls_mu = mu;
                        }
                    }
                    if (lane == 0) { C.men0 = m0; C.men1 = m1; C.ls_mu = ls_mu; C.ls = lsn; }
                    wave_sync();
                    return okv;
                };
                const int rv = volume_move_wg<NLAT, NW, LDSPOS, SPLIT>(vc, U, C.dv_max, wv, lv, sx, sdec, decide);
                if (usemom && rv == 1) msel ^= 1;                  // (the trial cell's moments are the walker's now)
                if (wv == 0) {
                    int sw = 0;
                    double l12 = 0.0, l21 = 0.0;
                    if (L == 2) { l12 = fast_log_pos(svol[0] / svol[1]); l21 = fast_log_pos(svol[1] / svol[0]); }
                    if (lane == 0) {
                        C.nvol_try = C.nvol_try + 1;
                        if (rv == 1) C.nvol_acc = C.nvol_acc + 1;
                        if (rv < 0) C.flag = C.flag | 1;
                        C.k_valid = 0; C.lgv12 = l12; C.lgv21 = l21;
                    }
                    wave_sync();
                    if (L == 2 && (C.record || do_switch)) {           // the same update and switch attempt as after a translation
                        const double mu = C.ls_mu;
                        const int kl = __builtin_amdgcn_readlane(lane_mu_to_bin(C.mg, mu), 0);
                        if (lane == 0) { C.k_cur = kl; C.k_valid = 1; }
                        wave_sync();
                        const double eta_fin = readlane_f64(lane_eta(C.mg, sweight, smub, sbw, mu, kl), 0);
                        const double dk = do_switch ? switch_dk(C.men0, C.men1, C.ls) : 0.0;
                        const double ex = exp_any(lane == 0 ? -dk : eta_fin - C.log_unbiased_norm);
                        sw = post_move(eta_fin, readlane_f64(ex, 0), readlane_f64(ex, 1), do_switch, U[6]);
                    }
                    if (lane == 0) {
                        if (L == 2) sdec[1] = C.ls;
                        if (mvlog) {
                            double* q = mvlog + ((size_t)blockIdx.x * nmoves + mv) * 8;
                            q[0] = 0.0; q[1] = (rv == 1 ? 1.0 : 0.0) + 2.0 * sw + 4.0; q[2] = C.men0; q[3] = svol[0];
                            q[4] = L == 2 ? C.men1 : 0.0; q[5] = L == 2 ? svol[1] : 0.0; q[6] = C.ls_mu; q[7] = diffkT;
                        }
                        if (C.dd) { const int w = C.within + 1; if (w == N) { C.within = 0; C.cyc = C.cyc + 1; } else C.within = w; }
                    }
                }
                if (L == 2) { wg_sync<NW>(); ls = sdec[1]; }       // (MINU or the switch may have changed the active lattice)
                else wave_sync();
            }
            MW_SW_ACC(42, MW_SW_NOW() - tvm);
            mv += 1;
            continue;
        }

        // ---- translations: slot s of the round is move mv + s, this wavefront's lattice of it ---------------------------
        const bool mine = slot < ntr;
        const int l = lat;
        const double* MV = smv + (ub + (mine ? slot : 0)) * MVS;
        const double x = MV[0], y = MV[1], z = MV[2];
        const int imol = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(MV[3]));
        const int i = imol - 1;
        double* P = pos + (size_t)(box0 + l) * N * 3;                                 // :1007-1018, 1076-1092
        double pnx = 0.0, pny = 0.0, pnz = 0.0;
        int ekind = 1, ecnt = 0;                       // the routine that evaluated this move (0: the moment path) and its in-range records
        if (mine) {
            double tx = x, ty = y, tz = z;                                            // the move in the active lattice
            if (L == 2 && l != ls - 1) {                                              // mapped into the partner lattice, :1042-1067
                const double* rc = srecip[ls - 1];
                double sxf = MW_HM(rc,1,1) * x + MW_HM(rc,2,1) * y + MW_HM(rc,3,1) * z;    // :1042-1050
                double syf = MW_HM(rc,1,2) * x + MW_HM(rc,2,2) * y + MW_HM(rc,3,2) * z;
                double szf = MW_HM(rc,1,3) * x + MW_HM(rc,2,3) * y + MW_HM(rc,3,3) * z;
                sxf = sxf * 0.5 * invPi; syf = syf * 0.5 * invPi; szf = szf * 0.5 * invPi;  // :1052-1054
                const double* hn = shmat[l];
                tx = MW_HM(hn,1,1) * sxf + MW_HM(hn,1,2) * syf + MW_HM(hn,1,3) * szf;
                ty = MW_HM(hn,2,1) * sxf + MW_HM(hn,2,2) * syf + MW_HM(hn,2,3) * szf;
                tz = MW_HM(hn,3,1) * sxf + MW_HM(hn,3,2) * syf + MW_HM(hn,3,3) * szf;
            }
            const uint32_t* LM = listm + (size_t)(box0 + l) * N * kRow;
            const int* NN = nn + (size_t)(box0 + l) * N;
            const double* IVl = siv + (size_t)l * ivcap * 3;
            auto getiv = [&](int k, double& a, double& b, double& c) { a = IVl[3 * k]; b = IVl[3 * k + 1]; c = IVl[3 * k + 2]; };
            const double* Pl = LDSPOS ? (spos + (size_t)l * N * 3) : P;
            auto getpos = [&](int j, double& a, double& b, double& c) { const double* p = Pl + 3 * (size_t)j; a = p[0]; b = p[1]; c = p[2]; };
            double xo, yo, zo;
            getpos(i, xo, yo, zo);
            pnx = xo + tx; pny = yo + ty; pnz = zo + tz;                              // :1079
            const unsigned short* SR = srow + (size_t)l * N * rstride;
            const unsigned char* SN = snn + l * N;
            auto row = [&](int jx, int sl) -> uint32_t {
                if constexpr (LDSLIST) { const uint32_t e = SR[jx * rstride + sl]; return (e & 63u) | ((e >> 6) << kJBits); }
                else return LM[(size_t)jx * kRow + sl];
            };
            auto nnof = [&](int jx) { return LDSLIST ? (int)SN[jx] : NN[jx]; };
            // the molecules the EARLIER moves of the round are trying to move: this evaluation is only good if none of those it
            // reads gets moved (slot o < slot; -1: no such move)
            int oth[SPEC > 1 ? SPEC - 1 : 1];
            unsigned cm = 0u;
            if constexpr (SPEC > 1) {
#pragma unroll
                for (int o = 0; o < SPEC - 1; ++o) {
                    const int io = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(smv[(ub + o) * MVS + 3])) - 1;
                    oth[o] = o < slot ? io : -1;
                    if (o < slot && io == i) cm |= 1u << o;
                }
            }
            MoveRes res;
            unsigned cme = 0u;
#ifdef MW_ABL_NOEVAL         // diagnostic build: no evaluation (energies 0) -- the decisions' share of a round
            const bool fast = true; res.eo = 0.0; res.en = 1e-3 * (double)(i & 7); res.io = res.in_ = res.so = res.sn = 0u;
#else
            bool fast = false;
            if constexpr (LDSLIST) {
                if (usemom) {
                    unsigned int nocounts[4];
                    fast = move_energy_mom_wave<true, SPEC - 1>(getpos, getiv, nnof, mom_of(msel, l), ws, nullptr, i, nnof(i), row(i, lane & 31), xo, yo, zo,
                                                                 pnx, pny, pnz, lane, res, nocounts, &ecnt, slmask + l * N, oth, &cme);
                    if (fast) ekind = 0;
                }
            } else if constexpr (MOMBIG) {
                if (usemom) {
                    unsigned int nocounts[4];
                    if constexpr (SPEC > 1) {
                        // Look-ahead: which earlier moves of the round this evaluation depends on, by DISTANCE.  It reads positions within
                        // the list radius of i and moments of molecules within the cutoff of i, which hold molecules within the cutoff of
                        // THOSE: everything lies within two cutoffs + the displacements of i (doubled: the partner lattice's image of a
                        // displacement, mapped through fractional coordinates, is longer or shorter by the ratio of the two cells).  The
                        // NEAREST image of the earlier molecule, by rounding the fractional separation -- exact whenever that image is
                        // closer than half the cell's narrowest width, however far the unwrapped positions have drifted apart; in a cell
                        // narrower than twice the reach every earlier move counts.  A superset of the true dependences (in a
                        // 4096-molecule box one pair of moves in twelve).
                        const double reach = 2.0 * (kSmallA * kSigma) + 4.0 * C.max_trans + 1e-6;
                        const double* rc = srecip[l];
                        const double* hn = shmat[l];
                        double wmin2i = 0.0;                                   // 1 / (narrowest width)^2 = max |column of recip|^2 / (2 pi)^2
#pragma unroll
                        for (int k = 1; k <= 3; ++k) {
                            const double n2 = MW_HM(rc,1,k) * MW_HM(rc,1,k) + MW_HM(rc,2,k) * MW_HM(rc,2,k) + MW_HM(rc,3,k) * MW_HM(rc,3,k);
                            wmin2i = n2 > wmin2i ? n2 : wmin2i;
                        }
                        wmin2i = wmin2i * (0.25 * invPi * invPi);
                        const bool wide = 4.0 * reach * reach * wmin2i < 1.0;    // narrowest width > 2 reach
#pragma unroll
                        for (int o = 0; o < SPEC - 1; ++o) {
                            if (oth[o] >= 0) {
                                double ox, oy, oz;
                                getpos(oth[o], ox, oy, oz);
                                const double dx = ox - xo, dy = oy - yo, dz = oz - zo;
                                double fx = (MW_HM(rc,1,1) * dx + MW_HM(rc,2,1) * dy + MW_HM(rc,3,1) * dz) * (0.5 * invPi);
                                double fy = (MW_HM(rc,1,2) * dx + MW_HM(rc,2,2) * dy + MW_HM(rc,3,2) * dz) * (0.5 * invPi);
                                double fz = (MW_HM(rc,1,3) * dx + MW_HM(rc,2,3) * dy + MW_HM(rc,3,3) * dz) * (0.5 * invPi);
                                fx -= __builtin_rint(fx); fy -= __builtin_rint(fy); fz -= __builtin_rint(fz);
                                const double mx = MW_HM(hn,1,1) * fx + MW_HM(hn,1,2) * fy + MW_HM(hn,1,3) * fz;
                                const double my = MW_HM(hn,2,1) * fx + MW_HM(hn,2,2) * fy + MW_HM(hn,2,3) * fz;
                                const double mz = MW_HM(hn,3,1) * fx + MW_HM(hn,3,2) * fy + MW_HM(hn,3,3) * fz;
                                if (!wide || mx * mx + my * my + mz * mz < reach * reach) cm |= 1u << o;
                            }
                        }
                    }
                    fast = move_energy_mom_wave<true, SPEC - 1, MW_BIG_WHEN>(getpos, getiv, nnof, mom_of(0, l), ws, nullptr, i, nnof(i), row(i, lane & 31), xo, yo, zo,
                                                                 pnx, pny, pnz, lane, res, nocounts, &ecnt, nullptr, oth, &cme);
                    if (fast) ekind = 0;
                }
            }
            if (usemom && !fast) MW_SW_ACC(43, 1ull);                   // (stamps build: evaluations the moment path declined)
            if (!fast) fast = move_energy_wave<true, SPEC - 1, false>(getpos, getiv, row, nnof, ws, sniv[l], i, nnof(i), row(i, lane & 31), xo, yo, zo,
                                                                     pnx, pny, pnz, lane, res, oth, &cme);
#endif
            cm |= cme;
            if (!fast) {
                Override none; none.idx = -1; none.x = none.y = none.z = 0.0;
                Override tr; tr.idx = i; tr.x = pnx; tr.y = pny; tr.z = pnz;
                res.eo = local_energy_wave(P, ivect + (size_t)(box0 + l) * ivcap * 3, LM, NN, i, none, none, lane, res.io, res.so);
                res.en = local_energy_wave(P, ivect + (size_t)(box0 + l) * ivcap * 3, LM, NN, i, tr, none, lane, res.in_, res.sn);
                cm = (1u << slot) - 1u;                            // (the plain routine keeps no account of what it read)
            }
            // (bit 31: not the moment path's evaluation -- if accepted, the lattice's moments are made afresh, so the move has to be
            //  the last of its round for the chain to be the sequential one bit for bit: decide_round)
            if (lane == 0) { sx[2 * wv] = res.eo; sx[2 * wv + 1] = res.en; scm[wv] = cm | ((usemom && ekind != 0) ? 0x80000000u : 0u); }
        }
        [[maybe_unused]] const unsigned long long sw_e = MW_SW_NOW();
        wg_sync<NW>();                                 // every evaluation of the round is in (one wavefront: nothing of the walker's state is read ahead of it)
        [[maybe_unused]] const unsigned long long sw_b = MW_SW_NOW();
        MW_SW_ACC(0, sw_e - sw_r0); MW_SW_ACC(1, sw_b - sw_e);
#ifdef MW_ABL_NODECIDE       // diagnostic build (tools/variants.py): every move rejected without a decision -- the evaluation's share of a round
        if (wv == 0 && lane == 0) { sdec[0] = ntr; sdec[1] = ls; for (int s = 0; s < SPEC; ++s) sdec[4 + s] = 0; }
#else
        if (wv == 0) decide_round(ntr, ub, U0, mv);
#endif
        [[maybe_unused]] const unsigned long long sw_d = MW_SW_NOW();
        MW_SW_ACC(2, sw_d - sw_b);
        if (NW > 1) wg_sync<NW>(); else wave_sync();
        const int nvalid = sdec[0];
        ls = sdec[1];
        const bool okm = mine && slot < nvalid && sdec[4 + slot] != 0;
        double cxo = 0.0, cyo = 0.0, czo = 0.0;                                        // (the old position, read back rather than held through the decisions)
        if constexpr (LDSLIST || MOMBIG) {
            if (usemom && okm && ekind == 0) { const double* Sp = LDSPOS ? spos + ((size_t)l * N + i) * 3 : P + 3 * (size_t)i; cxo = Sp[0]; cyo = Sp[1]; czo = Sp[2]; }
        }
        if (okm && lane == 0) {                                                       // :1150-1170: this wavefront's lattice
            P[3 * i] = pnx; P[3 * i + 1] = pny; P[3 * i + 2] = pnz;
            if (LDSPOS) { double* Sp = spos + ((size_t)l * N + i) * 3; Sp[0] = pnx; Sp[1] = pny; Sp[2] = pnz; }
        }
        if constexpr (LDSLIST || MOMBIG) {
            if (usemom && okm) {
                if (ekind == 0) moments_commit(mom_of(msel, l), ws, i, ecnt, cxo, cyo, czo, pnx, pny, pnz, lane);
                else if (lane == 0) sdec[2 + l] = 1;               // (made afresh from the committed positions at the top of the next round)
            }
        }
        mv += nvalid;
        // the next round must see the committed positions: with look-ahead, the other slots' of the same lattice too
        if (SPEC > 1) wg_sync<NW>(); else wave_sync();
        MW_SW_ACC(3, MW_SW_NOW() - sw_d); MW_SW_ACC(4, (unsigned long long)nvalid); MW_SW_ACC(5, 1ull);
    }
    MW_SW_ACC(7, MW_SW_NOW() - sw_k0);
#ifdef MW_SWEEP_STAMPS
    MW_SW_ACC(8, wall_clock64() - sw_w0);
#endif
    __syncthreads();
    if (L == 2 && C.record) {
        for (int t = tid; t < nbins; t += NTHR) {
            if (!C.samplerun) wweight[(size_t)wlk * nbins + t] = sweight[t];
            whist[(size_t)wlk * nbins + t] = shist[t];
            if (C.samplerun) wuhist[(size_t)wlk * nbins + t] = suhist[t];
        }
    }
    if (tid == 0) {
        wls[wlk] = C.ls; wmu[wlk] = C.ls_mu; wacc[wlk] += C.acc; wswitch[wlk] += C.nsw; wshift[wlk] += C.gauge;
        wvol[2 * wlk] += C.nvol_try; wvol[2 * wlk + 1] += C.nvol_acc;
        if (C.flag) wflag[wlk] |= C.flag;                   // bit 0: image-vector table outgrown, bit 1: 'dd' walker not in its window at eq_mc_cycles
        if (L == 2) { wfac[wlk] = C.wlf; wsum[wlk] = C.sumh; }
        if (C.dd) winflag[wlk] = C.mg.in_window;
        energy[box0] = C.men0;
        if (L == 2) energy[box0 + 1] = C.men1;
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
