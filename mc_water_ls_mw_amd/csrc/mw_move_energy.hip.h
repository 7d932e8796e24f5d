// mw_move_energy.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine:
// compute_local_real_energy (molint.F90:220-404) and the fused old/new evaluation of a trial move:
// local_energy_wave, move_energy_wave, k_move_energy, k_local_energy_single.
#pragma once
#include <type_traits>

#include "mw_common.hip.h"

namespace mw {

// =====================================================================================
// Local energy of one molecule = every pair and every triplet it takes part in
// (as centre or as end), the building block of a single-move Delta E.
// One 64-wide wavefront per request; lane l owns slot l of a neighbour list
// (maxneigh <= 64).  Pass 0: the lanes hold imol's own list and evaluate the pair
// term and g for the in-range lanes.  Then for every in-range j (a wave-uniform
// loop over the ballot mask):
//   * j--i--k triplets: lanes above j that are in range combine with j's
//     broadcast vector (molint.F90:302-318: the remaining entries of imol's list);
//   * i--j--k triplets: the lanes re-load jmol's list, shifted by j's image
//     (molint.F90:324-343), and each evaluates its k.
// A slot whose cos(theta) >= 0.99 contributes 0 (molint.F90:367-371; this is how
// the k == i self term drops out) and so does an out-of-range slot (G2).
//
// A request may carry up to two position overrides {index, xyz}: the molecule
// itself at a trial position, and (single-call drop-in path) the previously
// queried molecule whose host copy may have been reverted.  Overrides are used
// from registers wherever that index is gathered; with `commit` they are also
// written to the mirrored positions for later launches.
// =====================================================================================
struct Override { int idx; double x, y, z; };   // idx < 0: none (0-based molecule index)

// COHERENT = true (the resident server below): positions are read past the CU's vector L1 (agent scope, served by
// L2), because the server itself rewrites single positions between requests while its wavefront lives on.
template <bool COHERENT = false>
__device__ __forceinline__ void load_pos(const double* __restrict__ P, int j, const Override& o1, const Override& o2,
                                         double& x, double& y, double& z)
{
    const double* p = P + 3 * (size_t)j;
    if constexpr (COHERENT) {
        x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        x = p[0]; y = p[1]; z = p[2];
    }
    if (j == o1.idx) { x = o1.x; y = o1.y; z = o1.z; }
    if (j == o2.idx) { x = o2.x; y = o2.y; z = o2.z; }
}

// Returns the local energy in every lane.  `ninter` / `nslots` (wave-uniform) receive the number
// of in-range interactions as the reference enumerates them (pairs + triplet slots with
// cos(theta) < 0.99) and the number of list slots visited (n_i + sum of n_j over in-range j),
// which prices the call's algorithmic bytes.
template <bool COHERENT = false>
__device__ __forceinline__ double local_energy_wave(const double* __restrict__ P, const double* __restrict__ IV,
                                                    const uint32_t* __restrict__ LM, const int* __restrict__ NN,
                                                    int i, const Override& o1, const Override& o2, int lane,
                                                    unsigned int& ninter, unsigned int& nslots)
{
    double xi, yi, zi;
    load_pos<COHERENT>(P, i, o1, o2, xi, yi, zi);                         // molint.F90:258
    const int n_i = NN[i];

    // pass 0: imol's own list, one slot per lane
    const bool has = lane < n_i;
    const uint32_t e = has ? LM[(size_t)i * kRow + lane] : 0u;
    const int j = (int)(e & kJMask), kimg = (int)(e >> kJBits);
    double xj, yj, zj;
    load_pos<COHERENT>(P, j, o1, o2, xj, yj, zj);
    const double jvx = IV[3 * kimg], jvy = IV[3 * kimg + 1], jvz = IV[3 * kimg + 2];
    const double qx = xj + jvx, qy = yj + jvy, qz = zj + jvz;             // :269 position of j's image
    const double dx = qx - xi, dy = qy - yi, dz = qz - zi;                // :272
    const double r2 = dx * dx + dy * dy + dz * dz;                        // :273
    const bool inr = has && (r2 < kRcSq);                                 // :276
    double rinv = 0.0, e1 = 0.0, g = 0.0;
    if (inr) pair_terms(r2, rinv, e1, g);
    const double q = kSigSq * rinv * rinv;
    double acc2 = inr ? (kAeps * (kBigB * (q * q) - 1.0)) * e1 : 0.0;     // :294-297
    double acc3 = 0.0;
    unsigned int ntl = 0;            // per-lane count of triplet slots that contribute

    unsigned long long mask = __ballot(inr);
    ninter = (unsigned int)__popcll(mask);
    nslots = (unsigned int)n_i;
    while (mask) {                                                        // wave-uniform loop over in-range j
        const int jl = __ffsll((long long)mask) - 1;
        mask &= mask - 1ull;
        const double ajx = readlane_f64(dx, jl), ajy = readlane_f64(dy, jl), ajz = readlane_f64(dz, jl);   // jl is wave-uniform
        const double rinv_j = readlane_f64(rinv, jl), g_j = readlane_f64(g, jl);

        // j--i--k: later in-range slots of imol's own list                 :302-318
        if (inr && lane > jl) {
            const double ct = ((ajx * dx + ajy * dy + ajz * dz) * rinv_j) * rinv;     // :316,365
            if (ct < 0.99) { const double d = ct - kCos0; acc3 += g_j * (g * (d * d)); ++ntl; }   // :367-368,385-387
        }

        // i--j--k: jmol's list, translated by j's image                    :324-343
        const int jj = __builtin_amdgcn_readlane(j, jl);
        const double sjx = readlane_f64(jvx, jl), sjy = readlane_f64(jvy, jl), sjz = readlane_f64(jvz, jl);
        const double pjx = readlane_f64(qx, jl), pjy = readlane_f64(qy, jl), pjz = readlane_f64(qz, jl);
        const int n_j = NN[jj];
        nslots += (unsigned int)n_j;
        if (lane < n_j) {
            const uint32_t e2 = LM[(size_t)jj * kRow + lane];
            const int kk = (int)(e2 & kJMask), k2 = (int)(e2 >> kJBits);
            double xk, yk, zk;
            load_pos<COHERENT>(P, kk, o1, o2, xk, yk, zk);
            const double bx = ((xk + IV[3 * k2]) + sjx) - pjx;            // :332,334
            const double by = ((yk + IV[3 * k2 + 1]) + sjy) - pjy;
            const double bz = ((zk + IV[3 * k2 + 2]) + sjz) - pjz;
            const double s2 = bx * bx + by * by + bz * bz;                // :335
            if (s2 < kRcSq) {                                             // :361
                double rinv_k, e1_k, g_k;
                pair_terms(s2, rinv_k, e1_k, g_k);
                const double ct = (-(ajx * bx + ajy * by + ajz * bz) * rinv_j) * rinv_k;   // :320,341,365
                if (ct < 0.99) { const double d = ct - kCos0; acc3 += g_j * (g_k * (d * d)); ++ntl; }
            }
        }
    }
    const double tot = readlane_f64(dpp_wave_sum(acc2 + kLamEps * acc3), 63);                       // :397
    ninter += (unsigned int)__builtin_amdgcn_readlane(dpp_wave_sum_i32((int)ntl), 63);
    return tot;
}

// The same evaluation laid out for LATENCY (the resident server of the drop-in single call): the rows and third-body
// positions of up to eight in-range neighbours are requested together, so the whole call is four dependent memory
// round trips (row of imol -> positions of its entries -> rows of the in-range j -> positions of their entries)
// instead of three per in-range neighbour.  Same terms, same per-term arithmetic as local_energy_wave.
template <bool COHERENT>
__device__ __forceinline__ double local_energy_wave_batched(const double* __restrict__ P, const double* __restrict__ IV,
                                                            const uint32_t* __restrict__ LM, const int* __restrict__ NN,
                                                            int i, const Override& o1, const Override& o2, int lane,
                                                            unsigned int& ninter, unsigned int& nslots)
{
    constexpr int B = 8;
    double xi, yi, zi;
    load_pos<COHERENT>(P, i, o1, o2, xi, yi, zi);                         // molint.F90:258
    const int n_i = NN[i];
    const uint32_t e = LM[(size_t)i * kRow + lane];                       // rows are 64 entries long in memory: no need to wait for n_i
    const bool has = lane < n_i;
    const int j = has ? (int)(e & kJMask) : 0, kimg = has ? (int)(e >> kJBits) : 0;
    double xj, yj, zj;
    load_pos<COHERENT>(P, j, o1, o2, xj, yj, zj);
    const double jvx = IV[3 * kimg], jvy = IV[3 * kimg + 1], jvz = IV[3 * kimg + 2];
    const double qx = xj + jvx, qy = yj + jvy, qz = zj + jvz;             // :269
    const double dx = qx - xi, dy = qy - yi, dz = qz - zi;                // :272
    const double r2 = dx * dx + dy * dy + dz * dz;                        // :273
    const bool inr = has && (r2 < kRcSq);                                 // :276
    double rinv = 0.0, e1 = 0.0, g = 0.0;
    if (inr) pair_terms(r2, rinv, e1, g);
    const double q = kSigSq * rinv * rinv;
    double acc2 = inr ? (kAeps * (kBigB * (q * q) - 1.0)) * e1 : 0.0;     // :294-297
    double acc3 = 0.0;
    unsigned int ntl = 0;

    unsigned long long mask = __ballot(inr);
    ninter = (unsigned int)__popcll(mask);
    nslots = (unsigned int)n_i;
    while (mask) {                                                        // wave-uniform: batches of B in-range j
        // straight-line code, no branches between the loads: a batch shorter than B repeats its last neighbour
        // (harmless duplicate loads) so that every load of a stage is in flight before the first one is waited for
        int jls[B], jjs[B], njs[B];
        uint32_t e2s[B];
        const int left = __popcll(mask);
        const int cb = left < B ? left : B;
        int jlast = 0;
#pragma unroll
        for (int r = 0; r < B; ++r) {
            const int jl = mask ? __ffsll((long long)mask) - 1 : jlast;
            mask = mask ? (mask & (mask - 1ull)) : 0ull;
            jls[r] = jl; jlast = jl;
            jjs[r] = __builtin_amdgcn_readlane(j, jl);
        }
#pragma unroll
        for (int r = 0; r < B; ++r) { njs[r] = NN[jjs[r]]; e2s[r] = LM[(size_t)jjs[r] * kRow + lane]; }
        double xk[B], yk[B], zk[B], kx[B], ky[B], kz[B];
#pragma unroll
        for (int r = 0; r < B; ++r) {                                     // (stale slots past a row's end hold valid old entries)
            const int kk = (int)(e2s[r] & kJMask), k2 = (int)(e2s[r] >> kJBits);
            load_pos<COHERENT>(P, kk, o1, o2, xk[r], yk[r], zk[r]);
            kx[r] = IV[3 * k2]; ky[r] = IV[3 * k2 + 1]; kz[r] = IV[3 * k2 + 2];
        }
#pragma unroll
        for (int r = 0; r < B; ++r) {
            if (r < cb) {
                const int jl = jls[r];
                const double ajx = readlane_f64(dx, jl), ajy = readlane_f64(dy, jl), ajz = readlane_f64(dz, jl);
                const double rinv_j = readlane_f64(rinv, jl), g_j = readlane_f64(g, jl);
                if (inr && lane > jl) {                                               // j--i--k  :302-318
                    const double ct = ((ajx * dx + ajy * dy + ajz * dz) * rinv_j) * rinv;
                    if (ct < 0.99) { const double d = ct - kCos0; acc3 += g_j * (g * (d * d)); ++ntl; }
                }
                const double sjx = readlane_f64(jvx, jl), sjy = readlane_f64(jvy, jl), sjz = readlane_f64(jvz, jl);
                const double pjx = readlane_f64(qx, jl), pjy = readlane_f64(qy, jl), pjz = readlane_f64(qz, jl);
                nslots += (unsigned int)njs[r];
                if (lane < njs[r]) {                                                  // i--j--k  :324-343
                    const double bx = ((xk[r] + kx[r]) + sjx) - pjx;
                    const double by = ((yk[r] + ky[r]) + sjy) - pjy;
                    const double bz = ((zk[r] + kz[r]) + sjz) - pjz;
                    const double s2 = bx * bx + by * by + bz * bz;
                    if (s2 < kRcSq) {
                        double rinv_k, e1_k, g_k;
                        pair_terms(s2, rinv_k, e1_k, g_k);
                        const double ct = (-(ajx * bx + ajy * by + ajz * bz) * rinv_j) * rinv_k;
                        if (ct < 0.99) { const double d = ct - kCos0; acc3 += g_j * (g_k * (d * d)); ++ntl; }
                    }
                }
            }
        }
    }
    const double tot = readlane_f64(dpp_wave_sum(acc2 + kLamEps * acc3), 63);                       // :397
    ninter += (unsigned int)__builtin_amdgcn_readlane(dpp_wave_sum_i32((int)ntl), 63);
    return tot;
}

// -------------------------------------------------------------------------------------
// Batched single-move path: old AND new local energy of a trial move in one pass.
//
// What the two evaluations share is most of the work: the same list rows, the same
// gathered positions and -- for the i--j--k triplets -- the same r_jk, g_jk (only the
// molecule itself sits somewhere else), so each exp(.) of a third body is evaluated
// once and used for both.  Lanes are packed across ALL in-range neighbours j at once:
// the rows of the in-range j's are laid end to end (sum of nn(j) ~ 150 slots) and dealt
// to the 64 lanes, so a pass is ~80 % full instead of one partly filled pass per j.
// Each lane finds the j that owns its slot from the (wave-uniform) prefix sums and
// pulls that j's vector/weights from the owning lane with cross-lane reads.
//
// Cases where a periodic image of the molecule itself takes part: as third body
// (k == i through a non-identical image) both geometries are evaluated in line; a
// molecule that neighbours its own image (cells narrower than the list radius) takes
// the plain one-evaluation-at-a-time routine above.  The k == i self term is skipped
// explicitly (the reference drops it through its cos(theta) >= 0.99 rule).
// -------------------------------------------------------------------------------------
struct MoveRes { double eo, en; unsigned int io, so, in_, sn; };

// Per-wavefront LDS scratch: the in-range neighbours of the molecule, compacted by rank, so
// that any lane can pull neighbour `r`'s record with plain LDS reads (a broadcast when lanes
// of one group read the same record).
constexpr int kCap = 24;                       // more in-range neighbours than this: plain routine
struct WaveScratch {
    double q[3][kCap];                         // position of j's image            (molint.F90:269)
    double c[3][kCap];                         // j's image vector minus that position: takes r_k + ivect(k) into j's frame in ONE add
    double rinvo[kCap], rinvn[kCap];           // 1/r_ij at the old / trial position
    double go[kCap], gn[kCap];                 // exp(gamma sigma/(r_ij - a sigma)) old / trial
    int flag[kCap];                            // bit0 = in range of the old position, bit1 = of the trial position
    unsigned long long cm[kCap];               // bit p of the end-to-end slot numbering set: a row ends at slot p
    uint32_t qe[64];                           // queue of in-range third bodies: packed list entry ...
    int qown[64];                              // ... and rank | (image, inverse image, flags of that rank) << 5 of the j whose row it came from
};
static_assert(sizeof(WaveScratch) % 16 == 0, "scratch records must keep 16-byte alignment");

// Returns false (nothing written) when the request needs the plain routine.
// `row(j, s)` returns list entry s of molecule j and `nnof(j)` its row length: global memory (molecule-major
// list) or, for small systems in the sweep driver, LDS copies.
// SELFIMG = false: the caller guarantees that no periodic image of a molecule can be a third body of its own neighbours
// (cells at least three list radii wide along every cell vector -- every box that goes through the cell-grid builder):
// a row entry with k == i is then the molecule itself, and the both-geometries branch and the inverse-image bookkeeping
// behind it fall away (25 vector instructions per move).
#if defined(MW_SWEEP_STAMPS)   // a diagnostic build of the library only (tools/sweep_stamps.py): shader-clock cycles per stage of walker 0's
                              // first wavefront, summed over the launch -- g_sweep_stamps[16 + k] = cycles between stamp k - 1 and stamp k
__device__ unsigned long long g_sweep_stamps[48];
#define MW_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x < 64) { const unsigned long long mw_t = clock64(); \
                         if (lane == 0 && (k) > 0) g_sweep_stamps[16 + (k)] += mw_t - mw_tprev; mw_tprev = mw_t; } } while (0)
#elif defined(MW_LAT_STAMPS)      // tools/kbench built with -DMW_LAT_STAMPS only: where one wavefront's time goes (100 MHz ticks)
__device__ unsigned long long g_lat_stamps[16];
#define MW_STAMP(k) do { if (lane == 0) g_lat_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MW_STAMP(k) do { } while (0)
#endif

// NOTH > 0 (the Monte Carlo driver's look-ahead, mw_sweep.hip.h): `oth` holds the molecules that moves EARLIER in the chain
// are trying to move at the same time (-1: none); bit o of `cmask` comes back set when this evaluation read the position of
// oth[o] -- it is then only valid if that earlier move is rejected.  (The molecule's own index is the caller's to compare.)
// COUNTS = false (the Monte Carlo driver, which has no use for them): the interaction and slot counts of `res` are left unset and
// their bookkeeping -- a counter per item, a prefix sum's upper half, a wave-wide integer sum -- falls away.
template <bool SELFIMG = true, int NOTH = 0, bool COUNTS = true, typename PosFn, typename IvFn, typename RowFn, typename NnFn>
__device__ __forceinline__ bool move_energy_wave(PosFn getpos, IvFn getiv, RowFn row, NnFn nnof,
                                                 WaveScratch* __restrict__ ws, int niv,
                                                 int i, int n_i, uint32_t e,
                                                 double xo, double yo, double zo,
                                                 double xn, double yn, double zn, int lane, MoveRes& res,
                                                 const int* oth = nullptr, unsigned* cmask = nullptr)
{
    unsigned cm = 0u;                  // (per lane until the end: one wave-wide OR per evaluation, not a ballot per gather step and slot)
#ifdef MW_SWEEP_STAMPS
    unsigned long long mw_tprev = 0ull;
#endif
    // ---- pass 0: imol's own row; lanes 0..31 take slot l against the OLD position, lanes 32..63 the same
    // slot against the TRIAL position, so that one rsqrt/reciprocal/exp sequence serves both evaluations.
    // `e` arrives as entry (lane & 31) of imol's row, fetched by the caller ahead of time (whatever the row
    // length: rows are padded).  Rows longer than 32 entries take the plain routine, and so does a molecule
    // that neighbours one of its own periodic images.
    MW_STAMP(0);
    if (n_i > 32) return false;
    const int half = lane >> 5, sl = lane & 31;
    const bool has = sl < n_i;
    const int j = has ? (int)(e & kJMask) : 0, kimg = has ? (int)(e >> kJBits) : 0;
    if (SELFIMG && __ballot(has && j == i) != 0ull) return false;
    if constexpr (NOTH > 0) {
#pragma unroll
        for (int o = 0; o < NOTH; ++o) cm |= (has && j == oth[o]) ? 1u << o : 0u;
    }
    double xj, yj, zj, jvx, jvy, jvz;
    getpos(j, xj, yj, zj);
    getiv(kimg, jvx, jvy, jvz);
    const int nnj = has ? nnof(j) : 0;
    const double qx = xj + jvx, qy = yj + jvy, qz = zj + jvz;                 // molint.F90:269
    const double rix = half ? xn : xo, riy = half ? yn : yo, riz = half ? zn : zo;
    const double ax = qx - rix, ay = qy - riy, az = qz - riz;                 // :272
    const double r2 = ax * ax + ay * ay + az * az;
    const bool in = has && (r2 < kRcSq);                                      // :276
    const unsigned long long B = __ballot(in);
    const unsigned int mo_ = (unsigned int)B, mn_ = (unsigned int)(B >> 32);  // in range of the old / trial position, by slot
    const unsigned int U = mo_ | mn_;
    const int cntU = __popc(U);
    if (cntU > kCap) return false;
    MW_STAMP(1);

    double t3o = 0.0, t3n = 0.0;
    unsigned int nto = 0, ntn = 0;

    // ---- compact the in-range neighbours (of either position) into the wave's scratch ------------
    const bool inu = (U >> sl) & 1u;
    // (the in-range slots below this lane's: v_mbcnt counts them without a per-lane mask held in a register from move to move)
    const int rank = half ? (int)__builtin_amdgcn_mbcnt_hi(U, 0u) : (int)__builtin_amdgcn_mbcnt_lo(U, 0u);
    // The rows of the in-range j are laid end to end (slots 0..T-1).  An inclusive prefix sum over the 32
    // slot lanes of each half gives every j its first slot, and in its upper 16 bits the list slots each
    // evaluation visits (half 0: old position, half 1: trial position).
    const int mine = (inu ? nnj : 0) | (COUNTS ? ((in ? nnj : 0) << 16) : 0);
    int inc = mine;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);        // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);        // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);        // row_shr:4
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);        // row_shr:8
    const int r15 = __builtin_amdgcn_readlane(inc, 15), r47 = __builtin_amdgcn_readlane(inc, 47);
    inc += (lane & 16) ? (half ? r47 : r15) : 0;
    const int tot0 = __builtin_amdgcn_readlane(inc, 31), tot1 = __builtin_amdgcn_readlane(inc, 63);
    const int T = tot0 & 0xffff;
    const unsigned int so = (unsigned int)n_i + (unsigned int)(tot0 >> 16), sn = (unsigned int)n_i + (unsigned int)(tot1 >> 16);
    const int start = (inc & 0xffff) - (inu ? nnj : 0);
    // lane r of these two holds, for the in-range neighbour of rank r, its molecule and the first slot of
    // its row: the scan below locates a slot's owner from registers alone (no LDS round trips in front of
    // the row fetch).  Lanes that own no record aim at lane 63, which no rank reaches (cntU <= kCap).
    const int dstl = (inu && half == 0) ? rank : 63;
    const int jv  = __builtin_amdgcn_ds_permute(dstl << 2, j);
    const int stv = __builtin_amdgcn_ds_permute(dstl << 2, start);
    // the image that undoes `kimg`: cells are numbered centre first, then lexicographically without the
    // centre (compute_ivects, molint.F90:174-217), so the opposite cell is the mirror position
    const int cc = (niv - 1) >> 1;
    int kinv = 0;
    if constexpr (SELFIMG) {
        const int lin = kimg <= cc ? kimg - 1 : kimg, linv = niv - 1 - lin;
        kinv = kimg == 0 ? 0 : (linv < cc ? linv + 1 : linv);
    }
    // image (10 bits) | inverse image (10 bits) | in range of old, trial position (2 bits), by rank like jv
    const int flg = (int)((mo_ >> sl) & 1u) | (int)(((mn_ >> sl) & 1u) << 1);
    const int wv = __builtin_amdgcn_ds_permute(dstl << 2, kimg | (kinv << 10) | (flg << 20));
    // row-end marks: chunk c of the scan reads mask cm[c]; a slot's owner is the number of marks before it
    {   // (the address is worked out here, from a lane number the compiler cannot hoist: as a loop invariant of the callers'
        //  move loops it was one more register held from move to move -- the one that tipped a build of the driver into a spill)
        int lz = lane;
        asm volatile("" : "+v"(lz));
        if (lz < kCap) ws->cm[lz] = 0ull;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (inu && half == 0 && rank > 0 && start > 0)      // (rows are never empty: j lists i back)
        __hip_atomic_fetch_or(&ws->cm[(start - 1) >> 6], 1ull << ((start - 1) & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- rows of the in-range j: fetched ahead ----------------------------------------------------
    // The i--j--k stage below walks the rows of all in-range j laid end to end, 64 slots per chunk.  A
    // chunk's slot -> (owner rank, owner's packed word, row entry) fetch is issued TWO CHUNKS AHEAD of its
    // evaluation -- the first two right here, BEFORE the pair terms of pass 0 (a rsqrt, a reciprocal and an exp per
    // lane: the arithmetic the row fetch from global memory hides behind) -- so the scan never waits for a row.
    MW_STAMP(2);
    int nbefore = 0;                                         // row ends in the chunks already fetched (wave-uniform)
    auto fetch = [&](int t, int& own, int& wj, uint32_t& ent) {
        const unsigned long long M = ws->cm[t >> 6];         // one address for the whole wave
        const unsigned int mlo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)M);
        const unsigned int mhi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(M >> 32));
        own = nbefore + (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        nbefore += __popc(mlo) + __popc(mhi);
        const int jj = __builtin_amdgcn_ds_bpermute(own << 2, jv);
        const int st = __builtin_amdgcn_ds_bpermute(own << 2, stv);
        wj = __builtin_amdgcn_ds_bpermute(own << 2, wv);
        ent = t < T ? row(jj, t - st) : 0u;
    };
    int own_a = 0, own_b = 0, w_a = 0, w_b = 0; uint32_t ent_a = 0u, ent_b = 0u;
    if (T > 0) fetch(lane, own_a, w_a, ent_a);
    if (T > 64) fetch(64 + lane, own_b, w_b, ent_b);

    // ---- pass 0's pair terms; the in-range neighbours' records into the scratch --------------------
    double rinv = 0.0, e1 = 0.0, g = 0.0;
    if (in) pair_terms(r2, rinv, e1, g);
    const double qq = kSigSq * rinv * rinv;
    const double accp = in ? (kAeps * (kBigB * (qq * qq) - 1.0)) * e1 : 0.0;  // :294-297 (old in lanes 0..31, trial in 32..63)
    if (inu) {
        if (half == 0) {
            ws->q[0][rank] = qx; ws->q[1][rank] = qy; ws->q[2][rank] = qz;
            ws->c[0][rank] = jvx - qx; ws->c[1][rank] = jvy - qy; ws->c[2][rank] = jvz - qz;
            ws->rinvo[rank] = rinv; ws->go[rank] = g;
            ws->flag[rank] = flg;
        } else {
            ws->rinvn[rank] = rinv; ws->gn[rank] = g;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- the triplets, one ITEM per lane ------------------------------------------------------------
    // Two kinds of item share one instruction stream (what differs sits in two short sections that a pass without
    // such items skips):
    //  F  an in-range third body k of an in-range neighbour j, queued by the scan below: the i--j--k triplet
    //     (molint.F90:324-343) -- gather k, rsqrt / reciprocal / exp of r_jk, then cos(theta) at j in both geometries;
    //  P  a pair (a < b) of in-range neighbours: the j--i--k triplet (:302-318; a is the earlier list slot, so cos is
    //     formed in the reference's order) -- everything it needs is in the scratch already.
    // In both, cos = (A . B) r_A r_B with A = r_i - q_A from the molecule to neighbour A's image (A = j for F, a for P),
    // and the term is g_A g_B (cos - cos0)^2.
    MW_STAMP(3);
    const int npairs = cntU * (cntU - 1) / 2;
    int nq = 0;                                              // queued F items (wave-uniform)
    auto items = [&](int nP) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int total = nq + nP;
        for (int base = 0; base < total; base += 64) {
            const int idx = base + lane;
            const bool isF = idx < nq, isP = !isF && idx < total;
            int ia = 0, fl = 0;
            double box_ = 0.0, boy_ = 0.0, boz_ = 0.0, bnx = 0.0, bny = 0.0, bnz = 0.0;
            double rbo = 0.0, rbn = 0.0, gbo = 0.0, gbn = 0.0;
            if (isF) {
                const uint32_t e2 = ws->qe[idx];
                const int qw = ws->qown[idx];
                ia = qw & 31; fl = qw >> 25;
                const int kk = (int)(e2 & kJMask), k2 = (int)(e2 >> kJBits);
                // r_jk = (r_k + ivect(k)) + (ivect(j) - q_j)   (:332,334, the last two terms taken together in pass 0: the very
                // expression the scan used for the in-range decision)
                double xk, yk, zk, kvx, kvy, kvz;
                getpos(kk, xk, yk, zk);
                getiv(k2, kvx, kvy, kvz);
                box_ = (xk + kvx) + ws->c[0][ia];
                boy_ = (yk + kvy) + ws->c[1][ia];
                boz_ = (zk + kvz) + ws->c[2][ia];
                const double s2 = box_ * box_ + boy_ * boy_ + boz_ * boz_;       // :335 (in range: tested at scan)
                double rk, gk, e1k;
                pair_terms(s2, rk, e1k, gk);
                bnx = box_; bny = boy_; bnz = boz_;
                rbo = rk; rbn = rk; gbo = gk; gbn = gk;
            } else if (isP) {
                const int p = idx - nq;
                int b = (int)((1.0f + __builtin_amdgcn_sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);   // (v_sqrt_f32, 1 ulp: the two fix-ups below absorb it;
                                                                                                 //  the IEEE sqrtf expansion costs ~20 instructions)
                if (b * (b - 1) / 2 > p) --b;
                if ((b + 1) * b / 2 <= p) ++b;
                ia = p - b * (b - 1) / 2;
                fl = ws->flag[ia] & ws->flag[b];
                const double qbx = ws->q[0][b], qby = ws->q[1][b], qbz = ws->q[2][b];
                box_ = xo - qbx; boy_ = yo - qby; boz_ = zo - qbz;
                bnx = xn - qbx; bny = yn - qby; bnz = zn - qbz;
                rbo = ws->rinvo[b]; rbn = ws->rinvn[b]; gbo = ws->go[b]; gbn = ws->gn[b];
            }
            if (isF || isP) {
                const double pax = ws->q[0][ia], pay = ws->q[1][ia], paz = ws->q[2][ia];
                const double rao = ws->rinvo[ia], ran = ws->rinvn[ia], gao = ws->go[ia], gan = ws->gn[ia];
                if (fl & 1) {
                    const double ct = (((xo - pax) * box_ + (yo - pay) * boy_ + (zo - paz) * boz_) * rao) * rbo;   // :316,320,341,365
                    if (ct < 0.99) { const double d = ct - kCos0; t3o += gao * (gbo * (d * d)); if constexpr (COUNTS) ++nto; }    // :367-368,385-387
                }
                if (fl & 2) {
                    const double ct = (((xn - pax) * bnx + (yn - pay) * bny + (zn - paz) * bnz) * ran) * rbn;
                    if (ct < 0.99) { const double d = ct - kCos0; t3n += gan * (gbn * (d * d)); if constexpr (COUNTS) ++ntn; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        nq = 0;
    };

    // The P items run HERE, between the row fetch and the first use of what it brings: without them the scan's first
    // chunk waits for global memory (measured: one merged pass of F and P items at the end saves ~60 instructions per
    // move and LOSES 5 % -- the fetch latency it exposes costs more than the instructions it saves).
    if (npairs > 0) items(npairs);

    // ---- i--j--k triplets (molint.F90:324-343): the rows of all in-range j, end to end --------
    // SCAN: every slot gets the cheap part (gather, distance test); the ~1/3 that are in range are queued (entry + owner
    // rank, 8 bytes) in the wave's scratch as F items; whenever 64 are queued a pass of F items runs with every lane busy.
    MW_STAMP(4);
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < T;
        const int own = own_a, wj = w_a;
        const uint32_t e2 = ent_a;
        own_a = own_b; w_a = w_b; ent_a = ent_b;
        if (t0 + 128 < T) fetch(t + 128, own_b, w_b, ent_b);
        const int kk = (int)(e2 & kJMask), k2 = (int)(e2 >> kJBits);
        if constexpr (NOTH > 0) {
#pragma unroll
            for (int o = 0; o < NOTH; ++o) cm |= (valid && kk == oth[o]) ? 1u << o : 0u;
        }
        double xk, yk, zk, kvx, kvy, kvz;
        getpos(kk, xk, yk, zk);
        getiv(k2, kvx, kvy, kvz);
        const double cjx = ws->c[0][own], cjy = ws->c[1][own], cjz = ws->c[2][own];
        const bool self = valid && (kk == i);
        bool selfmove = false;
        if constexpr (SELFIMG) {
            const bool selfimg = self && (k2 == ((wj >> 10) & 1023));   // the molecule itself, not an image: k's shift undoes j's
            selfmove = self && !selfimg;
        }
        const double box_ = (xk + kvx) + cjx;                                    // :332,334 (see the F items)
        const double boy_ = (yk + kvy) + cjy;
        const double boz_ = (zk + kvz) + cjz;
        const double s2o = box_ * box_ + boy_ * boy_ + boz_ * boz_;              // :335
        if (SELFIMG && __ballot(selfmove) != 0ull) {
            // an image of the molecule itself as third body moves with it: both geometries, in line (rare)
            if (selfmove) {
                const int fl = wj >> 20;
                const double pjx = ws->q[0][own], pjy = ws->q[1][own], pjz = ws->q[2][own];
                const double bnx = (xn + kvx) + cjx, bny = (yn + kvy) + cjy, bnz = (zn + kvz) + cjz;
                const double s2n = bnx * bnx + bny * bny + bnz * bnz;
                double rk, gk, e1k;
                if ((s2o < kRcSq) && (fl & 1)) {
                    pair_terms(s2o, rk, e1k, gk);
                    const double ct = (-((pjx - xo) * box_ + (pjy - yo) * boy_ + (pjz - zo) * boz_) * ws->rinvo[own]) * rk;
                    if (ct < 0.99) { const double d = ct - kCos0; t3o += ws->go[own] * (gk * (d * d)); ++nto; }
                }
                if ((s2n < kRcSq) && (fl & 2)) {
                    pair_terms(s2n, rk, e1k, gk);
                    const double ct = (-((pjx - xn) * bnx + (pjy - yn) * bny + (pjz - zn) * bnz) * ws->rinvn[own]) * rk;
                    if (ct < 0.99) { const double d = ct - kCos0; t3n += ws->gn[own] * (gk * (d * d)); ++ntn; }
                }
            }
        }
        const bool inq = valid && !self && (s2o < kRcSq);                        // :361; the k == i self term is dropped
        const unsigned long long mq = __ballot(inq);
        const int c = __popcll(mq);
        if (nq + c > 64) items(0);
        if (inq) {
            const int slot = nq + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mq, 0u));
            ws->qe[slot] = e2; ws->qown[slot] = own | (wj << 5);
        }
        nq += c;
    }
    MW_STAMP(5);
    if (nq > 0) items(0);                                    // what the scan left in the queue
    __builtin_amdgcn_wave_barrier();                          // scratch is reused by the wave's next request
    MW_STAMP(6);

    // Wave sums on the DPP network (no LDS round trips): afterwards lane 63 holds the totals.
    double eo, en;                                                                                 // :397
    dpp_wave_sum2(kLamEps * t3o + (half == 0 ? accp : 0.0), kLamEps * t3n + (half == 1 ? accp : 0.0), eo, en);
    res.eo = eo; res.en = en;
    if constexpr (COUNTS) {
        const unsigned int cs = (unsigned int)__builtin_amdgcn_readlane(dpp_wave_sum_i32((int)(nto | (ntn << 16))), 63);
        nto = cs & 0xffffu; ntn = cs >> 16;
        res.io = (unsigned int)__popc(mo_) + nto; res.in_ = (unsigned int)__popc(mn_) + ntn;
        res.so = so; res.sn = sn;
    }
    if constexpr (NOTH > 0) {          // OR over the lanes, on the DPP network
        unsigned v = cm;
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);      // row_shr:8
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);      // row_bcast:15 into rows 1 and 3
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);      // row_bcast:31 into rows 2 and 3
        *cmask = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    }
    MW_STAMP(7);
    return true;
}

// -------------------------------------------------------------------------------------
// THE MOMENT PATH of the batched single-move kernel (round 4).  The i--j--k triplets -- the rows of all in-range neighbours j
// scanned slot by slot, ~115 distance tests and ~25 rsqrt / exp per evaluation, two thirds of the fused routine's instructions --
// are sums over j's OTHER neighbours, and those sums do not depend on where i sits:
//   sum_{k != i} g_jk (u_ji . u_jk - c0)^2 = u^T S2' u - 2 c0 u . S1' + c0^2 S0',     u = unit vector j -> i,
// with S0', S1', S2' the moments of j's in-range neighbourhood (mw_common.hip.h: kMomStride; computed for every molecule of the box
// by the full-box pass, k_model_energy's `mom` output) less i's own contribution at its mirrored position.  A request then costs
// pass 0 (i's own row, as before), one item per in-range neighbour and geometry for the i--j--k sums (a 96-byte read and ~60
// multiply-adds) and the j--i--k pairs: O(neighbours) like the full-box kernel, instead of O(neighbours^2).
//
// The reference's local-energy path drops a triplet slot whose cos(theta) >= 0.99 (molint.F90:367-371; the rule that removes the
// k == i self term) -- a moment sum cannot drop a term.  But a third body k with cos(theta_ijk) >= 0.99 lies within the cutoff of
// i itself (|ik|^2 = a^2 + b^2 - 2ab cos < max(a, b)^2 for an angle below 8.2 degrees), i.e. k is one of i's OWN in-range neighbours:
// the pair pass, which walks all pairs (a, b) of those anyway, tests both of them as centres -- is b within the cutoff of a, and
// cos(theta_iab) >= 0.99 (less a 1e-9 margin, on squares: no square root)? -- and a request with such a triplet is DECLINED to the
// plain routine, like the other cases the fused routine does not take.  On ice, and in any physical configuration of this model,
// there is none.  For boxes whose cells are at least three list radii wide only (SELFIMG = false: an image of i is never a third
// body of i's neighbours, so each neighbour holds exactly one contribution of i).
// -------------------------------------------------------------------------------------
// `ptab[p]` = (a | b << 8) of the p-th pair a < b (a table in LDS: decoding p costs a dozen instructions otherwise).  The interaction
// and slot counts are ADDED, lane by lane, to `acc` = {interactions old, slots old, interactions new, slots new}: the caller sums
// them over the lanes once per work item instead of once per request.
//
// SWEEP = true: the Monte Carlo driver's small walkers (mw_sweep.hip.h), whose cells are narrower than three list radii and whose
// moments change under the routine's feet.  (1) A neighbour j that is in range through TWO of its images holds two contributions of
// i -- images of i as each other's third bodies at j; one trial move in twenty of the reference's 48-molecule Ih cell, 7.7 A wide --
// which the pair pass, meeting every pair of in-range entries anyway, finds and accounts for (see there).  A molecule that lists an
// image of itself (a cell narrower than the list radius) is declined.  (2) No pair table (the driver's LDS is counted in bytes): pair
// p of the triangular numbering is decoded arithmetically.  (3) No counts.
// (4) The record of rank r keeps its molecule (ws->qown[r]) and `cnt_u` returns the number of records: the caller's
// moments_commit() brings the moments up to date from them when the move is accepted.  (5) Look-ahead (NOTH > 0, as
// move_energy_wave): bit o of `cmask` comes back set when the evaluation read the position of oth[o] OR the moments of a molecule
// that lists oth[o] -- `lmask[j]` = the molecules of j's row as a bit mask (N <= 64), fetched with j's position.
// WHEN the moments are asked for: 1 = once the in-range entries are known (a 96-byte read per in-range neighbour and geometry); 0 =
// after the pair terms (see below); 2 = WITH the positions, for every row entry whether in range or not -- one dependent load level
// fewer, which is what a lone wavefront reading global memory pays for (the resident server: ~0.5 us a level).
// `lmask` == nullptr with NOTH > 0 (walkers with more than 64 molecules, where a row's molecules do not fit a bit mask): no dependence
// test in here -- the caller decides by distance (mw_sweep.hip.h) and `cmask` comes back 0.
template <bool SWEEP = false, int NOTH = 0, int WHEN = (SWEEP && NOTH == 0) ? 0 : 1, typename PosFn, typename IvFn, typename NnFn>
__device__ __forceinline__ bool move_energy_mom_wave(PosFn getpos, IvFn getiv, NnFn nnof, const double* __restrict__ MOM,
                                                     WaveScratch* __restrict__ ws, const unsigned short* __restrict__ ptab, int i, int n_i, uint32_t e,
                                                     double xo, double yo, double zo, double xn, double yn, double zn,
                                                     int lane, MoveRes& res, unsigned int (&acc)[4], int* cnt_u = nullptr,
                                                     const unsigned long long* __restrict__ lmask = nullptr, const int* oth = nullptr,
                                                     unsigned* cmask = nullptr)
{
    // ---- pass 0: as move_energy_wave -- lanes 0..31 slot l of i's row against the OLD position, lanes 32..63 against the TRIAL one
#ifdef MW_SWEEP_STAMPS
#define MW_MOM_WHY(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sweep_stamps[44 + (k)] += 1ull; } while (0)
#else
#define MW_MOM_WHY(k) do { } while (0)
#endif
    if (n_i > 32) { MW_MOM_WHY(0); return false; }
    const int half = lane >> 5, sl = lane & 31;
    const bool has = sl < n_i;
    const int j = has ? (int)(e & kJMask) : 0, kimg = has ? (int)(e >> kJBits) : 0;
    if (SWEEP && __ballot(has && j == i) != 0ull) { MW_MOM_WHY(1); return false; }
    unsigned cm = 0u;
    if constexpr (NOTH > 0) {
        if (lmask != nullptr) {
#pragma unroll
            for (int o = 0; o < NOTH; ++o) cm |= (has && j == oth[o]) ? 1u << o : 0u;
        }
    }
    double xj, yj, zj, jvx, jvy, jvz;
    double M[10];
    if constexpr (WHEN == 2) {
        const double2* Mj = reinterpret_cast<const double2*>(MOM + (size_t)(has ? j : i) * kMomStride);
#pragma unroll
        for (int c = 0; c < 5; ++c) { const double2 v = Mj[c]; M[2 * c] = v.x; M[2 * c + 1] = v.y; }
    }
    getpos(j, xj, yj, zj);
    getiv(kimg, jvx, jvy, jvz);
    [[maybe_unused]] unsigned long long lmj = 0ull;
    if constexpr (NOTH > 0) lmj = lmask != nullptr ? lmask[j] : 0ull;
    const int nnj = (has && !SWEEP) ? nnof(j) : 0;
    const double qx = xj + jvx, qy = yj + jvy, qz = zj + jvz;                 // molint.F90:269
    const double rix = half ? xn : xo, riy = half ? yn : yo, riz = half ? zn : zo;
    const double ax = qx - rix, ay = qy - riy, az = qz - riz;                 // :272
    const double r2 = ax * ax + ay * ay + az * az;
    const bool in = has && (r2 < kRcSq);                                      // :276
    const unsigned long long B = __ballot(in);
    const unsigned int mo_ = (unsigned int)B, mn_ = (unsigned int)(B >> 32);
    const unsigned int U = mo_ | mn_;
    const int cntU = __popc(U);
    if (cntU > kCap) { MW_MOM_WHY(2); return false; }
    if constexpr (NOTH > 0) {          // whose moments this evaluation reads: those of the in-range j -- which hold every molecule of j's row
#pragma unroll
        for (int o = 0; o < NOTH; ++o) cm |= (in && oth[o] >= 0 && ((lmj >> (oth[o] & 63)) & 1ull) != 0ull) ? 1u << o : 0u;
    }
    const bool inu = (U >> sl) & 1u;
    const int rank = half ? (int)__builtin_amdgcn_mbcnt_hi(U, 0u) : (int)__builtin_amdgcn_mbcnt_lo(U, 0u);
    // j's moments: requested NOW, by the lane that holds j and this geometry, and used after the pair terms (whose rsqrt /
    // reciprocal / exp the read hides behind)
    // (the driver's one-move-at-a-time builds -- thousands of walkers, sixteen wavefronts per compute unit to hide a read behind, and
    //  a budget of 128 vector registers -- ask for them AFTER the pair terms instead: twenty registers fewer held across those)
    constexpr bool kLateMoments = WHEN == 0;
    auto load_moments = [&]() {
        const double2* Mj = reinterpret_cast<const double2*>(MOM + (size_t)(in ? j : i) * kMomStride);   // (a lane without an in-range j reads i's own: harmless, unused)
#pragma unroll
        for (int c = 0; c < 5; ++c) { const double2 v = Mj[c]; M[2 * c] = v.x; M[2 * c + 1] = v.y; }
    };
    if constexpr (WHEN == 1) load_moments();
    double rinv = 0.0, e1 = 0.0, g = 0.0;
    if (in) pair_terms(r2, rinv, e1, g);
    const double qq = kSigSq * rinv * rinv;
    const double accp = in ? (kAeps * (kBigB * (qq * qq) - 1.0)) * e1 : 0.0;  // :294-297 (old in lanes 0..31, trial in 32..63)
    const int flg = (int)((mo_ >> sl) & 1u) | (int)(((mn_ >> sl) & 1u) << 1) | (SWEEP ? j << 2 : 0);
    if (inu) {                                            // the in-range neighbours' records by rank, for the pair pass
        if (half == 0) {
            ws->q[0][rank] = qx; ws->q[1][rank] = qy; ws->q[2][rank] = qz;
            ws->rinvo[rank] = rinv; ws->go[rank] = g;
            ws->flag[rank] = flg;
            if constexpr (SWEEP) ws->qown[rank] = j;
        } else {
            ws->rinvn[rank] = rinv; ws->gn[rank] = g;
        }
    }

    // ---- i--j--k: one item per in-range neighbour and geometry, in the lane that holds them ---------------------------------
    // i's own term inside j's moments belongs to the OLD position (the one the full-box pass saw): lanes of the trial geometry take
    // the old 1/r and g from the lane 32 below
    if constexpr (kLateMoments) load_moments();
    const int lsrc = (lane & 31) << 2;
    const double g_old = __hiloint2double(__builtin_amdgcn_ds_bpermute(lsrc, __double2hiint(g)), __builtin_amdgcn_ds_bpermute(lsrc, __double2loint(g)));
    const double r_old = __hiloint2double(__builtin_amdgcn_ds_bpermute(lsrc, __double2hiint(rinv)), __builtin_amdgcn_ds_bpermute(lsrc, __double2loint(rinv)));
    double t3 = 0.0;
    unsigned int nt = 0u;
    if (in) {
        double S0 = M[0], S1x = M[1], S1y = M[2], S1z = M[3];
        double Sxx = M[4], Syy = M[5], Sxy = M[6], Sxz = M[7], Syz = M[8];
        double Szz = (S0 - Sxx) - Syy;                       // (trace of sum g u u^T = sum g)
        double cn = M[9];
        if ((mo_ >> sl) & 1u) {      // j's moments hold i at its mirrored (old) position: that term is not a third body
            const double ux = (xo - qx) * r_old, uy = (yo - qy) * r_old, uz = (zo - qz) * r_old;   // unit vector j -> i (old)
            const double hx = g_old * ux, hy = g_old * uy, hz = g_old * uz;
            S0 -= g_old; S1x -= hx; S1y -= hy; S1z -= hz;
            Sxx -= hx * ux; Syy -= hy * uy; Szz -= hz * uz; Sxy -= hx * uy; Sxz -= hx * uz; Syz -= hy * uz;
            cn -= 1.0;
        }
        const double vx = -ax * rinv, vy = -ay * rinv, vz = -az * rinv;                            // unit vector j -> i, this geometry
        const double wx = Sxx * vx + Sxy * vy + Sxz * vz, wy = Sxy * vx + Syy * vy + Syz * vz, wz = Sxz * vx + Syz * vy + Szz * vz;
        const double quad = vx * wx + vy * wy + vz * wz, lin = vx * S1x + vy * S1y + vz * S1z;
        t3 = g * ((quad - 2.0 * kCos0 * lin) + kCos0 * kCos0 * S0);                                // :324-343,385-387 summed over k
        nt = (unsigned int)(cn + 0.5);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- j--i--k: the pairs (a < b) of in-range neighbours (:302-318), one item per pair AND geometry (item t: pair t >> 1, geometry
    // t & 1), and the triplets the 0.99 rule could touch ---------------------------------------------------------------------------
    double t3p = 0.0;
    unsigned int ntp = 0u;
    bool hard = false;
    [[maybe_unused]] unsigned long long anysame = 0ull;
    const int gq = lane & 1;                                  // this lane's geometry in the pair pass
    const double rqx = gq ? xn : xo, rqy = gq ? yn : yo, rqz = gq ? zn : zo;
    const double* rinvq = gq ? ws->rinvn : ws->rinvo;
    const double* gvq = gq ? ws->gn : ws->go;
    const int nitems = cntU * (cntU - 1);                     // 2 x pairs
    // pair p = b (b - 1) / 2 + a of the triangular numbering, a < b, without a table: b from a single-precision square root (exact
    // enough for p < 2^20; two integer corrections make it exact), a dozen instructions per pass of the wavefront
    [[maybe_unused]] auto pair_of = [](int p, int& a_, int& b_) {
        int bb = (int)((1.0f + __fsqrt_rn(1.0f + 8.0f * (float)p)) * 0.5f);
        if (((bb * (bb - 1)) >> 1) > p) --bb;
        if ((((bb + 1) * bb) >> 1) <= p) ++bb;
        b_ = bb; a_ = p - ((bb * (bb - 1)) >> 1);
    };
    for (int base = 0; base < nitems; base += 64) {
        const int t = base + lane;
        const bool live = t < nitems;
        int ia, b;
        if constexpr (SWEEP) {
            pair_of(live ? t >> 1 : 0, ia, b);
        } else {
            const unsigned int ab = live ? (unsigned int)ptab[t >> 1] : 0x0100u;
            ia = (int)(ab & 0xffu); b = (int)(ab >> 8);
        }
        const int fa = ws->flag[ia], fb = ws->flag[b];
        const bool act = live && (((fa & fb) >> gq) & 1);
        const double pax = ws->q[0][ia], pay = ws->q[1][ia], paz = ws->q[2][ia];
        const double pbx = ws->q[0][b], pby = ws->q[1][b], pbz = ws->q[2][b];
        const double ra = rinvq[ia], rb = rinvq[b];
        const double Ax = rqx - pax, Ay = rqy - pay, Az = rqz - paz, Bx = rqx - pbx, By = rqy - pby, Bz = rqz - pbz;
        const double ct = ((Ax * Bx + Ay * By + Az * Bz) * ra) * rb;                               // :316,365
        if (act && ct < 0.99) { const double d = ct - kCos0; t3p += gvq[ia] * (gvq[b] * (d * d)); ++ntp; }   // :367-368,385-387
        if constexpr (SWEEP) anysame |= __ballot(live && (fa >> 2) == (fb >> 2));      // two records of ONE molecule: see below
        // a and b as each other's third bodies: only when they lie within the cutoff of each other -- on ice a molecule's in-range
        // neighbours do not (first shell 2.76 A, its members 4.5 A apart, cutoff 4.31 A), so the wavefront usually skips this
        const double dx = pbx - pax, dy = pby - pay, dz = pbz - paz;          // a -> b
        const double r2ab = dx * dx + dy * dy + dz * dz;
        const bool abin = act && r2ab < kRcSq;
        if (__ballot(abin) != 0ull) {
            constexpr double kC2 = (0.99 - 1e-9) * (0.99 - 1e-9);
            const double da = Ax * dx + Ay * dy + Az * dz, db = -(Bx * dx + By * dy + Bz * dz);    // (a->i).(a->b), (b->i).(b->a)
            hard = hard || (abin && ((da > 0.0 && (da * ra) * (da * ra) >= kC2 * r2ab) || (db > 0.0 && (db * rb) * (db * rb) >= kC2 * r2ab)));
        }
    }
    if constexpr (SWEEP) {
        // Records a and b that are two images of ONE molecule j (a cell narrower than two cutoffs: the reference's Ih example, 7.7 A;
        // one trial move in twenty there): j has two arms to i, and its moments hold both at their OLD ends.  For the old geometry that
        // is what the reference's loops see (entry a meets the other arm as a third body, entry b likewise); for the trial geometry the
        // other arm has moved too -- the sums are linear in the moments, so the pair's trial item puts that right arm by arm.  (No 0.99
        // rule: the arms are a cell vector apart.)  A pass of its own, after the main one: its registers are not the main pass's.
        if (anysame != 0ull) {
            for (int base = 0; base < nitems; base += 64) {
                const int t = base + lane;
                const bool live = t < nitems;
                int ia, b;
                pair_of(live ? t >> 1 : 0, ia, b);
                const int fa = ws->flag[ia], fb = ws->flag[b];
                if (live && (fa >> 2) == (fb >> 2)) {
                    if (gq == 1) {
                        const double pax = ws->q[0][ia], pay = ws->q[1][ia], paz = ws->q[2][ia];
                        const double pbx = ws->q[0][b], pby = ws->q[1][b], pbz = ws->q[2][b];
                        const double ran = ws->rinvn[ia], rbn = ws->rinvn[b], gan = ws->gn[ia], gbn = ws->gn[b];
                        const double rao = ws->rinvo[ia], rbo = ws->rinvo[b], gao = ws->go[ia], gbo = ws->go[b];
                        const double nax = (xn - pax) * ran, nay = (yn - pay) * ran, naz = (zn - paz) * ran;                       // j -> i, trial
                        const double nbx = (xn - pbx) * rbn, nby = (yn - pby) * rbn, nbz = (zn - pbz) * rbn;
                        const double oax = (xo - pax) * rao, oay = (yo - pay) * rao, oaz = (zo - paz) * rao;                       // j -> i, old
                        const double obx = (xo - pbx) * rbo, oby = (yo - pby) * rbo, obz = (zo - pbz) * rbo;
                        const bool ao = fa & 1, an = fa & 2, bo = fb & 1, bn = fb & 2;
                        const double dnn = (nax * nbx + nay * nby + naz * nbz) - kCos0;
                        const double dab = (nax * obx + nay * oby + naz * obz) - kCos0, dba = (nbx * oax + nby * oay + nbz * oaz) - kCos0;
                        double corr = 0.0;
                        if (an && bn) corr += 2.0 * (gan * (gbn * (dnn * dnn)));
                        if (an && bo) corr -= gan * (gbo * (dab * dab));
                        if (bn && ao) corr -= gbn * (gao * (dba * dba));
                        t3p += corr;
                    }
                }
            }
        }
    }
    const bool decline = __ballot(hard) != 0ull;
    __builtin_amdgcn_wave_barrier();                          // scratch is reused by the wave's next request
    if (decline) { MW_MOM_WHY(3); return false; }
    if constexpr (SWEEP) *cnt_u = anysame != 0ull ? -cntU : cntU;      // (negative: some molecule holds more than one record -- moments_commit)

    double eo, en;                                                                                 // :397
    dpp_wave_sum2(kLamEps * ((gq == 0 ? t3p : 0.0) + (half == 0 ? t3 : 0.0)) + (half == 0 ? accp : 0.0),
                  kLamEps * ((gq == 1 ? t3p : 0.0) + (half == 1 ? t3 : 0.0)) + (half == 1 ? accp : 0.0), eo, en);
    res.eo = eo; res.en = en;
    if constexpr (!SWEEP) {
        // this request's interactions (in-range pairs + triplet slots that contribute) and list slots (n_i + the rows of its in-range
        // neighbours: what prices its algorithmic bytes), left in the lanes that know them
        const unsigned int ci = (in ? 1u : 0u) + nt, cs = (in ? (unsigned int)nnj : 0u) + (sl == 0 ? (unsigned int)n_i : 0u);
        acc[0] += (half == 0 ? ci : 0u) + (gq == 0 ? ntp : 0u); acc[1] += half == 0 ? cs : 0u;
        acc[2] += (half == 1 ? ci : 0u) + (gq == 1 ? ntp : 0u); acc[3] += half == 1 ? cs : 0u;
    }
    if constexpr (NOTH > 0) {          // OR over the lanes, on the DPP network
        unsigned v = cm;
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);      // row_shr:8
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);      // row_bcast:15 into rows 1 and 3
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);      // row_bcast:31 into rows 2 and 3
        *cmask = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    }
    return true;
}

// An accepted move's moments (the Monte Carlo driver; the records of the evaluation that preceded it are still in `ws`): the moments
// of every molecule that had or now has i within the cutoff lose i's old contribution and gain the new one -- lane r, the record of
// rank r -- and i's own are the sum over its new neighbourhood.  (Szz is not stored: S0 - Sxx - Syy, mw_common.hip.h.)
__device__ __forceinline__ void moments_commit(double* __restrict__ MOM, WaveScratch* __restrict__ ws, int i, int cntU,
                                               double xo, double yo, double zo, double xn, double yn, double zn, int lane)
{
    // Lane r applies record r to its molecule's moments.  When a molecule holds several records (cnt_u < 0: a cell so narrow that
    // two -- or, narrow in two directions, up to four -- images of it are in range), the lane of its FIRST record applies them all,
    // in rank order, and the others none: one read-modify-write per molecule.
    const bool multi = cntU < 0;
    cntU = multi ? -cntU : cntU;
    if (lane < cntU) {
        const int j = ws->qown[lane];
        double2* Mj = reinterpret_cast<double2*>(MOM + (size_t)j * kMomStride);
        double M[10];
#pragma unroll
        for (int c = 0; c < 5; ++c) { const double2 v = Mj[c]; M[2 * c] = v.x; M[2 * c + 1] = v.y; }
        auto apply = [&](int r) {
            const int f = ws->flag[r];
            const double qx = ws->q[0][r], qy = ws->q[1][r], qz = ws->q[2][r];
            if (f & 1) {
                const double ri = ws->rinvo[r], g = ws->go[r];
                const double ux = (xo - qx) * ri, uy = (yo - qy) * ri, uz = (zo - qz) * ri;      // unit vector j -> i (old)
                const double hx = g * ux, hy = g * uy, hz = g * uz;
                M[0] -= g; M[1] -= hx; M[2] -= hy; M[3] -= hz;
                M[4] -= hx * ux; M[5] -= hy * uy; M[6] -= hx * uy; M[7] -= hx * uz; M[8] -= hy * uz; M[9] -= 1.0;
            }
            if (f & 2) {
                const double ri = ws->rinvn[r], g = ws->gn[r];
                const double ux = (xn - qx) * ri, uy = (yn - qy) * ri, uz = (zn - qz) * ri;      // unit vector j -> i (new)
                const double hx = g * ux, hy = g * uy, hz = g * uz;
                M[0] += g; M[1] += hx; M[2] += hy; M[3] += hz;
                M[4] += hx * ux; M[5] += hy * uy; M[6] += hx * uy; M[7] += hx * uz; M[8] += hy * uz; M[9] += 1.0;
            }
        };
        bool first = true;
        if (!multi) apply(lane);
        else {
            for (int r = 0; r < cntU; ++r) {
                if (ws->qown[r] == j) {
                    if (r < lane) first = false;
                    if (first) apply(r);
                }
            }
        }
        if (first) {
#pragma unroll
            for (int c = 0; c < 5; ++c) Mj[c] = make_double2(M[2 * c], M[2 * c + 1]);
        }
    }
    // i's own: the records' contributions at the trial position, seven records at a time through 70 doubles of the scratch (ws->c,
    // which this path does not use otherwise): lane u of a chunk writes its ten numbers, lane c < 10 then adds up component c --
    // in rank order, so the sum does not depend on anything but the records -- and stores it
    double* T = &ws->c[0][0];
    double Sc = 0.0;
    for (int r0 = 0; r0 < cntU; r0 += 7) {
        const int r = r0 + lane;
        if (lane < 7 && r < cntU) {
            double v[10] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (ws->flag[r] & 2) {
                const double ri = ws->rinvn[r], g = ws->gn[r];
                const double ux = (ws->q[0][r] - xn) * ri, uy = (ws->q[1][r] - yn) * ri, uz = (ws->q[2][r] - zn) * ri;   // unit vector i -> j
                const double hx = g * ux, hy = g * uy, hz = g * uz;
                v[0] = g; v[1] = hx; v[2] = hy; v[3] = hz; v[4] = hx * ux; v[5] = hy * uy; v[6] = hx * uy; v[7] = hx * uz; v[8] = hy * uz; v[9] = 1.0;
            }
#pragma unroll
            for (int c = 0; c < 10; ++c) T[c * 7 + lane] = v[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 10) {
            double t[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) t[u] = T[lane * 7 + u];
#pragma unroll
            for (int u = 0; u < 7; ++u) Sc += (r0 + u < cntU) ? t[u] : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane < 10) MOM[(size_t)i * kMomStride + lane] = Sc;
}

// One workgroup per work item {box, first request, last request+1}: the requests are
// sorted by box on upload, so the workgroup stages that box's positions in LDS once
// (LDSPOS) and its 16 wavefronts then serve the item's requests from LDS gathers.
//   mode bit 0: write e_old (mirrored positions), bit 1: write e_new (trial position)
constexpr int kMoveChunk = 2048;   // requests per work item when the box is staged in LDS

// (LAYOUT: SoA measures 1.4 % faster than the paired layout here -- 1288 vs 1306 us, tools/kbench -- now that the scan
// reads one vector less per slot; the full-box kernel keeps the paired layout, where it is the faster one)
// MOMPATH = true (with LDSPOS, SELFIMG = false): the moment path above; `mom` = the box moments [box][N][kMomStride] of the launch's boxes.
#ifndef MW_MOVE_WHEN
#define MW_MOVE_WHEN 1     // when the batched kernel asks for the moments (move_energy_mom_wave: WHEN); 0 and 2 measured: see DESIGN 3.2
#endif
template <bool LDSPOS, int LAYOUT = kLayoutSoA, bool SELFIMG = true, bool MOMPATH = false>
__global__ __launch_bounds__(1024)
void k_move_energy(const double* __restrict__ pos, const double* __restrict__ ivect,
                   const int* __restrict__ nivect, const uint32_t* __restrict__ listm,
                   const int* __restrict__ nn, const int4* __restrict__ work,
                   const int* __restrict__ req_imol, const double* __restrict__ req_trial,
                   const int* __restrict__ perm,
                   double* __restrict__ e_old, double* __restrict__ e_new,
                   unsigned int* __restrict__ counts,   // [nreq][4]: inter_old, slots_old, inter_new, slots_new
                   int* __restrict__ declined,          // [0], [1] = number of requests left to k_move_fallback (the word of this launch's parity,
                                                        // mode bit 2), then {request, box} pairs
                   int N, int ivcap, int mode, const double* __restrict__ mom = nullptr,
                   unsigned int* __restrict__ mtot = nullptr)   // MOMPATH: [work item][4] = {interactions old, slots old, interactions new, slots new} of the item's served requests
{
    static_assert(!MOMPATH || (LDSPOS && !SELFIMG), "the moment path serves boxes staged in LDS whose cells hold no self-images");
    __shared__ unsigned short s_ptab[MOMPATH ? kCap * (kCap - 1) / 2 : 1];           // pair p -> (a | b << 8), a < b
    if constexpr (MOMPATH) {
        for (int p = threadIdx.x; p < kCap * (kCap - 1) / 2; p += 1024) {
            int b = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
            if (b * (b - 1) / 2 > p) --b;
            if ((b + 1) * b / 2 <= p) ++b;
            s_ptab[p] = (unsigned short)((p - b * (b - 1) / 2) | (b << 8));
        }
    }
    unsigned int acc[4] = {0u, 0u, 0u, 0u};
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int4 w = work[blockIdx.x];
    const int b = w.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* P  = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    const uint32_t* LM = listm + (size_t)b * N * kRow;
    const int* NN = nn + (size_t)b * N;
    const int niv = nivect[b];

    // dynamic LDS: [positions when LDSPOS][image vectors][16 wave scratches] and, when LDSPOS, [row lengths, one
    // byte per molecule][the molecules of the item's requests]; positions and image vectors in the layout LAYOUT
    // (LdsVecs, mw_common.hip.h: the paired layout gathers a vector in 6 LDS cycles instead of 10)
    double* spos = smem;
    double* siv = smem + (LDSPOS ? lds_vec_bytes((size_t)N) / 8 : 0);
    WaveScratch* ws = reinterpret_cast<WaveScratch*>(siv + lds_vec_bytes((size_t)ivcap) / 8) + wave;
    unsigned char* snn = reinterpret_cast<unsigned char*>(reinterpret_cast<WaveScratch*>(siv + lds_vec_bytes((size_t)ivcap) / 8) + 16);
    int* simol = reinterpret_cast<int*>(snn + (((size_t)N + 7) & ~(size_t)7));
    __shared__ int s_next;                                                   // next request nobody has taken yet
    const int nreq = w.z - w.y;                                              // <= kMoveChunk when LDSPOS
    const double iv_first = stage_iv_begin<1024>(IV, niv, tid);
    if (LDSPOS) {
        // every staging load of the item in flight before the first wait (a loop of load / wait / store per element costs a
        // dozen HBM round trips in a row): row lengths and request molecules first, into registers, then the box
        constexpr int kB = 4;
        for (int base = 0; base < N; base += kB * 1024) {
            int v[kB];
#pragma unroll
            for (int k = 0; k < kB; ++k) { const int t = base + tid + k * 1024; v[k] = NN[t < N ? t : N - 1]; }
#pragma unroll
            for (int k = 0; k < kB; ++k) { const int t = base + tid + k * 1024; if (t < N) snn[t] = (unsigned char)v[k]; }   // maxneigh <= 64
        }
        for (int base = 0; base < nreq; base += kB * 1024) {
            int v[kB];
#pragma unroll
            for (int k = 0; k < kB; ++k) { const int t = base + tid + k * 1024; v[k] = req_imol[w.y + (t < nreq ? t : nreq - 1)]; }
#pragma unroll
            for (int k = 0; k < kB; ++k) { const int t = base + tid + k * 1024; if (t < nreq) simol[t] = v[k]; }
        }
        stage_vecs<LAYOUT, 1024>(spos, P, N, N, tid);
        if (tid == 0) s_next = 16;
    }
    stage_iv_end<LAYOUT, 1024>(siv, IV, niv, ivcap, tid, iv_first);
    __syncthreads();

    const LdsVecs<LAYOUT> vpos{spos, N}, viv{siv, ivcap};
    auto getiv = [&](int k, double& x, double& y, double& z) { viv.get(k, x, y, z); };
    auto getpos = [&](int jx, double& x, double& y, double& z) {
        if constexpr (LDSPOS) vpos.get(jx, x, y, z);
        else { const double* p = P + 3 * (size_t)jx; x = p[0]; y = p[1]; z = p[2]; }
    };
    auto row = [&](int jx, int sl) { return LM[(size_t)jx * kRow + sl]; };
    auto nnof = [&](int jx) { return LDSPOS ? (int)snn[jx] : NN[jx]; };
    auto imol_of = [&](int q) { return LDSPOS ? simol[q] : req_imol[w.y + q]; };

    // Requests are handed out dynamically: the first sixteen go to the sixteen wavefronts, after that a wavefront
    // takes the next untaken one (an LDS counter) when it STARTS a request, and fetches entry (lane & 31) of that
    // molecule's own row right away -- a request never begins by waiting on memory, and a wavefront that drew
    // cheap requests simply serves more of them (static dealing left ~8 % of the wave-time idle at the item's end).
    // (Without LDS staging an item holds at most 16 requests: one per wavefront.)
    int cur = wave;
    int i = cur < nreq ? imol_of(cur) : 0;
    uint32_t e = cur < nreq ? row(i, lane & 31) : 0u;
    double tx = 0.0, ty = 0.0, tz = 0.0;                     // the request's trial position, fetched with its row entry
    if ((mode & 2) && cur < nreq) { const double* t3 = req_trial + 3 * (size_t)(w.y + cur); tx = t3[0]; ty = t3[1]; tz = t3[2]; }
    while (cur < nreq) {
        int nxt = 0;
        if (lane == 0) nxt = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        nxt = LDSPOS ? __builtin_amdgcn_readfirstlane(nxt) : nreq;
        int i_nx = 0; uint32_t e_nx = 0u;
        double tx_nx = 0.0, ty_nx = 0.0, tz_nx = 0.0;
        if (nxt < nreq) {
            i_nx = imol_of(nxt); e_nx = row(i_nx, lane & 31);
            if (mode & 2) { const double* t3 = req_trial + 3 * (size_t)(w.y + nxt); tx_nx = t3[0]; ty_nx = t3[1]; tz_nx = t3[2]; }
        }

        const int m = w.y + cur;
        double xo, yo, zo;
        getpos(i, xo, yo, zo);
        double xn = xo, yn = yo, zn = zo;
        if (mode & 2) { xn = tx; yn = ty; zn = tz; }

        MoveRes r;
        bool fast;
        if constexpr (MOMPATH) fast = move_energy_mom_wave<false, 0, MW_MOVE_WHEN>(getpos, getiv, nnof, mom + (size_t)b * N * kMomStride, ws, s_ptab, i, nnof(i), e, xo, yo, zo, xn, yn, zn, lane, r, acc);
        else fast = move_energy_wave<SELFIMG>(getpos, getiv, row, nnof, ws, niv, i, nnof(i), e, xo, yo, zo, xn, yn, zn, lane, r);
        if (!fast) {
            // a request the fused routine declines (a row longer than 32 entries, more than kCap in-range neighbours, a
            // molecule that neighbours its own image -- never on ice) is left to k_move_fallback: with the plain routine
            // inlined here its registers counted against this loop (37 scalar registers spilled to vector lanes, ~30
            // vector instructions per request on moving them), and calling it out of line costs scratch (+8 % time)
            if (lane == 0) {
                const int k = atomicAdd(&declined[(mode >> 2) & 1], 1);
                declined[2 + 2 * k] = m; declined[3 + 2 * k] = b;
            }
        } else if (lane == 0) {
            const size_t o = (size_t)perm[m];
            if constexpr (MOMPATH) { r.io = r.so = r.in_ = r.sn = 0u; }       // (the counts of served requests go to `mtot`, summed per work item)
            if (mode & 1) { e_old[o] = r.eo; counts[4 * o] = r.io; counts[4 * o + 1] = r.so; }
            if (mode & 2) { e_new[o] = r.en; counts[4 * o + 2] = r.in_; counts[4 * o + 3] = r.sn; }
        }
        cur = nxt; i = i_nx; e = e_nx; tx = tx_nx; ty = ty_nx; tz = tz_nx;
    }
    if constexpr (MOMPATH) {         // the item's counts: lanes -> wavefront -> workgroup, ONE plain store per item (thousands of wavefronts adding to
        __shared__ unsigned int s_tot[16][4];                              // four global words serialise: +0.3 ms on a 0.9 ms launch)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int tot = dpp_wave_sum_i32((int)acc[c]);
            if (lane == 63) s_tot[wave][c] = (unsigned int)tot;
        }
        __syncthreads();
        if (tid < 4) {
            unsigned int t = 0u;
            for (int wv = 0; wv < 16; ++wv) t += s_tot[wv][tid];
            mtot[4 * (size_t)blockIdx.x + tid] = t;
        }
    }
}

// The requests k_move_energy declined, one wavefront each with the plain routine (positions and rows from global memory).
//   grid = any, block = 256; exits at once when nothing was declined
// The list has two count words used by alternate launches (mode bit 2): this kernel zeroes the OTHER one, which the next
// launch's k_move_energy will count in -- a memset per launch (a fill kernel and its dispatch gap, ~10 us) is saved.
__global__ __launch_bounds__(256)
void k_move_fallback(const double* __restrict__ pos, const double* __restrict__ ivect,
                     const uint32_t* __restrict__ listm, const int* __restrict__ nn,
                     const int* __restrict__ req_imol, const double* __restrict__ req_trial, const int* __restrict__ perm,
                     double* __restrict__ e_old, double* __restrict__ e_new, unsigned int* __restrict__ counts,
                     int* __restrict__ declined, int N, int ivcap, int mode)
{
    const int par = (mode >> 2) & 1;
    const int n = declined[par];
    if (blockIdx.x == 0 && threadIdx.x == 0) declined[par ^ 1] = 0;
    const int lane = threadIdx.x & 63;
    const int wave = (int)(blockIdx.x * (blockDim.x >> 6)) + (int)(threadIdx.x >> 6), nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    for (int k = wave; k < n; k += nwaves) {
        const int m = declined[2 + 2 * k], b = declined[3 + 2 * k];
        const double* P  = pos + (size_t)b * N * 3;
        const double* IV = ivect + (size_t)b * ivcap * 3;
        const uint32_t* LM = listm + (size_t)b * N * kRow;
        const int* NN = nn + (size_t)b * N;
        const int i = req_imol[m];
        Override none; none.idx = -1; none.x = none.y = none.z = 0.0;
        Override tr; tr.idx = -1; tr.x = tr.y = tr.z = 0.0;
        if (mode & 2) { tr.idx = i; tr.x = req_trial[3 * (size_t)m]; tr.y = req_trial[3 * (size_t)m + 1]; tr.z = req_trial[3 * (size_t)m + 2]; }
        MoveRes r;
        r.eo = local_energy_wave(P, IV, LM, NN, i, none, none, lane, r.io, r.so);
        r.en = local_energy_wave(P, IV, LM, NN, i, tr, none, lane, r.in_, r.sn);
        if (lane == 0) {
            const size_t o = (size_t)perm[m];
            if (mode & 1) { e_old[o] = r.eo; counts[4 * o] = r.io; counts[4 * o + 1] = r.so; }
            if (mode & 2) { e_new[o] = r.en; counts[4 * o + 2] = r.in_; counts[4 * o + 3] = r.sn; }
        }
    }
}

// Single request with by-value overrides (the drop-in compute_local_real_energy call):
// one wave, result written straight to host-visible memory.
__global__ __launch_bounds__(64)
void k_local_energy_single(double* __restrict__ pos, const double* __restrict__ ivect,
                           const uint32_t* __restrict__ listm, const int* __restrict__ nn,
                           int b, int i, Override o1, Override o2, int commit,
                           double* __restrict__ e_out, int N, int ivcap,
                           unsigned long long* __restrict__ done, unsigned long long seq)
{
    const int lane = threadIdx.x;
    double* P = pos + (size_t)b * N * 3;
    unsigned int ni, ns;
    const double e = local_energy_wave(P, ivect + (size_t)b * ivcap * 3, listm + (size_t)b * N * kRow,
                                       nn + (size_t)b * N, i, o1, o2, lane, ni, ns);
    if (lane == 0) {
        *e_out = e;
        if (commit) {   // these two indices are never read from memory in this launch (overrides win)
            if (o1.idx >= 0) { P[3 * o1.idx] = o1.x; P[3 * o1.idx + 1] = o1.y; P[3 * o1.idx + 2] = o1.z; }
            if (o2.idx >= 0 && o2.idx != o1.idx) { P[3 * o2.idx] = o2.x; P[3 * o2.idx + 1] = o2.y; P[3 * o2.idx + 2] = o2.z; }
        }
        // the host spins on `done` (host-visible memory) instead of going through a stream synchronisation
        __threadfence_system();
        __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// =====================================================================================
// Resident server for the drop-in single call (compute_local_real_energy behind the unchanged Fortran call
// sites, mc_moves.F90:1010,1083): a kernel launch plus its completion cost ~30 us, fifteen times what the
// reference spends on the whole evaluation, so the engine keeps ONE small kernel resident instead -- started by
// the first single call, stopped by any entry point that changes device state behind its back (uploads, list
// builds, the batch kernels) and by mw_finalize, and by itself after `idle_limit` empty polls.  Workgroup w (one
// wavefront) serves mail slot w (lattice ils goes to slot (ils - 1) % nslots, so the two lattices of a move can
// be evaluated concurrently from two host threads): it polls the slot's request lines (device memory the host
// writes through the BAR, or host-mapped memory), evaluates the request with move_energy_wave (positions read
// past the L1: this kernel itself commits the two overridden positions between requests), and writes energy +
// sequence word to the reply line in host memory.  The cost of a call is a posted PCIe write each way, a poll
// and the evaluation, not a launch.
// One workgroup per slot, not one wavefront of a shared workgroup: a compute unit's vector memory pipeline returns
// data in order, and with eight wavefronts polling across PCIe (1.3 us a read) every gather of the one that is
// working queued behind their polls -- 8.9 us an evaluation against 3.6 us on a compute unit of its own.
// =====================================================================================
struct MailSlot {                       // 64-byte aligned; one per served slot (request lines and reply line may live in different copies)
    // request, line A (the host writes the fields of both lines, then seq_a, then seq_b, a store fence between them:
    // a line that shows the new sequence word shows its new fields, and seq_b == seq_a says both lines are in)
    unsigned long long seq_a;
    int box, imol;                      // 0-based
    double x1, y1, z1;                  // position of imol, if flags & 2
    int flags, prev;                    // bit 0: commit the positions, bit 1: x1.. present, bit 2: x2.. present; prev 0-based
    unsigned long long pad_a[2];
    // request, line B
    double x2, y2, z2;                  // position of the previously queried molecule, if flags & 4
    unsigned long long pad_b[4];
    unsigned long long seq_b;
    // reply line (device -> host)
    unsigned long long rep_seq;         // the request this reply belongs to (written last)
    double energy;
    unsigned int ninter, nslots;
    unsigned long long pad_c[5];
};
static_assert(sizeof(MailSlot) == 192, "two request lines and one reply line");
struct MailHead { int quit; int exited; int pad[14]; };

template <bool COHERENT>
__global__ __launch_bounds__(64)
void k_local_server(MailHead* __restrict__ head, MailSlot* __restrict__ slots, const MailSlot* __restrict__ reqs,
                    double* __restrict__ pos, const double* __restrict__ ivect, const int* __restrict__ nivect,
                    const uint32_t* __restrict__ listm, const int* __restrict__ nn,
                    int N, int ivcap, long long idle_limit, int stamps,
                    double* mom, double* pm, int* momok)   // the moment path (below), or nullptr
{
    __shared__ WaveScratch ws;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x;
    MailSlot* m = slots + w;
    const unsigned long long* words = reinterpret_cast<const unsigned long long*>(reqs + w);    // request lines
    unsigned long long last = __hip_atomic_load(&m->rep_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    long long idle = 0;
    auto word = [&](unsigned long long v, int l) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
        return ((unsigned long long)hi << 32) | lo;
    };
    for (;;) {                                                        // every exit condition is reached by every wavefront
        // one load instruction fetches both request lines (lane l reads word l & 15): two PCIe reads in flight together
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long v = __hip_atomic_load(words + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long seq = word(v, 0);
        if (seq != last && word(v, 15) == seq) {
            const unsigned long long w1 = word(v, 1), w5 = word(v, 5);
            const int b = (int)(unsigned int)w1, i = (int)(unsigned int)(w1 >> 32);
            const int flags = (int)(unsigned int)w5, prev = (int)(unsigned int)(w5 >> 32);
            Override o1, o2;
            o1.idx = (flags & 2) ? i : -1;
            o1.x = __longlong_as_double((long long)word(v, 2)); o1.y = __longlong_as_double((long long)word(v, 3)); o1.z = __longlong_as_double((long long)word(v, 4));
            o2.idx = (flags & 4) ? prev : -1;
            o2.x = __longlong_as_double((long long)word(v, 8)); o2.y = __longlong_as_double((long long)word(v, 9)); o2.z = __longlong_as_double((long long)word(v, 10));
            double* P = pos + (size_t)b * N * 3;
            const double* IVb = ivect + (size_t)b * ivcap * 3;
            const uint32_t* LMb = listm + (size_t)b * N * kRow;
            const int* NNb = nn + (size_t)b * N;
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            // The lane-packed evaluation of the batched kernel (old and trial position both = the molecule's position): a
            // third of the instructions of the plain routine, which matters for ONE wavefront on its own; the plain
            // routine, with its loads batched, takes the cases that one declines.
            double xi, yi, zi;
            load_pos<COHERENT>(P, i, o1, o2, xi, yi, zi);
            auto getpos = [&](int jx, double& x, double& y, double& z) { load_pos<COHERENT>(P, jx, o1, o2, x, y, z); };
            auto getiv = [&](int k, double& x, double& y, double& z) { x = IVb[3 * k]; y = IVb[3 * k + 1]; z = IVb[3 * k + 2]; };
            auto row = [&](int jx, int sl) { return LMb[(size_t)jx * kRow + sl]; };
            auto nnof = [&](int jx) { return NNb[jx]; };
            MoveRes res;
            double e;
            // THE MOMENT PATH (round 4; move_energy_mom_wave<SWEEP>, as in the Monte Carlo driver): `mom` = every molecule's moments of
            // the served boxes, made by the full-box kernel when the server starts, and `pm` = the positions they were made FROM.  The
            // host changes positions only through the requests' own overrides -- the queried molecule and the one queried before it
            // (anything else is an exclusive entry point, which stops the server) -- so at most `prev` can have moved since: if its
            // committed position is no longer the one in `pm`, its neighbours' moments and its own are brought up to date first
            // (moments_commit: an accepted move of the host's chain, one request in four at most), then the queried molecule is
            // evaluated with pm[i] as the "old" position -- the arm the moments hold -- and the request's as the trial one.
            // Every neighbour is read from `pm`.  A request this does not cover (an uncommitted override of prev, a decline) takes the
            // routines below; one that leaves the moments behind (a declined update) switches the path off for the box.
            bool served = false;
            if (mom != nullptr) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (this wavefront's own earlier writes to mom / pm: past the L1)
                double* MOMb = mom + (size_t)b * N * kMomStride;
                double* PMb = pm + (size_t)b * N * 3;
                auto getpm = [&](int jx, double& x, double& y, double& z) { const double* q = PMb + 3 * (size_t)jx; x = q[0]; y = q[1]; z = q[2]; };
                // ONE round trip for everything the path needs to get going (a lone wavefront pays ~0.5 us per dependent load level)
                const int pv = o2.idx >= 0 ? prev : i;
                const int mk = momok[b];
                double px, py, pz, qx, qy, qz;
                getpm(pv, px, py, pz);
                getpm(i, qx, qy, qz);
                const uint32_t erow = row(i, lane & 31);
                const int ni = nnof(i);
                unsigned int nocounts[4];
                int cnt = 0;
                bool ok = true;
                if (o2.idx >= 0 && o2.idx != i && (px != o2.x || py != o2.y || pz != o2.z)) {       // prev has moved since its moments were made
                    if (!(flags & 1) || mk == 0) ok = false;       // (an override that is not committed: the moments must not follow it)
                    else {
                        MoveRes r2;
                        if (move_energy_mom_wave<true, 0, 2>(getpm, getiv, nnof, MOMb, &ws, nullptr, prev, nnof(prev), row(prev, lane & 31),
                                                                 px, py, pz, o2.x, o2.y, o2.z, lane, r2, nocounts, &cnt)) {
                            moments_commit(MOMb, &ws, prev, cnt, px, py, pz, o2.x, o2.y, o2.z, lane);
                            if (lane == 0) { PMb[3 * prev] = o2.x; PMb[3 * prev + 1] = o2.y; PMb[3 * prev + 2] = o2.z; }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        } else {
                            ok = false;
                            if (lane == 0) momok[b] = 0;           // (the moments no longer follow the positions: off for this box until the server restarts)
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        }
                    }
                }
                if (ok && move_energy_mom_wave<true, 0, 2>(getpm, getiv, nnof, MOMb, &ws, nullptr, i, ni, erow,
                                                               qx, qy, qz, xi, yi, zi, lane, res, nocounts, &cnt) && mk != 0) { e = res.en; served = true; }
            }
            if (served) {
            } else if (move_energy_wave(getpos, getiv, row, nnof, &ws, nivect[b], i, nnof(i), row(i, lane & 31), xi, yi, zi, xi, yi, zi, lane, res)) {
                e = res.eo;
            } else {
                unsigned int ni, ns;
                e = local_energy_wave_batched<COHERENT>(P, IVb, LMb, NNb, i, o1, o2, lane, ni, ns);
            }
            const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) {
                // the reply: energy and sequence word in ONE 16-byte store (one PCIe write: the host reads the word, then the energy)
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const unsigned long long eb = (unsigned long long)__double_as_longlong(e);
                const u32x4 rep = {(unsigned int)seq, (unsigned int)(seq >> 32), (unsigned int)eb, (unsigned int)(eb >> 32)};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(&m->rep_seq), "v"(rep) : "memory");
                if (stamps) {     // 100 MHz stamps for tools/kbench (poll issued -> request decoded -> evaluated): diagnostics only
                    __hip_atomic_store(&m->pad_c[0], t1 - t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&m->pad_c[1], t2 - t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (flags & 1) {  // after the reply is on its way: written through to L2 (agent scope); the kernel's end makes
                                  // them visible to every later launch
                    if (o1.idx >= 0) {
                        __hip_atomic_store(P + 3 * o1.idx, o1.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(P + 3 * o1.idx + 1, o1.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(P + 3 * o1.idx + 2, o1.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (o2.idx >= 0 && o2.idx != o1.idx) {
                        __hip_atomic_store(P + 3 * o2.idx, o2.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(P + 3 * o2.idx + 1, o2.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(P + 3 * o2.idx + 2, o2.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            last = seq;
            idle = 0;
        } else {
            const int q = __hip_atomic_load(&head->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (q != 0 || ++idle > idle_limit) break;
        }
    }
    if (lane == 0) __hip_atomic_fetch_add(&head->exited, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace mw
