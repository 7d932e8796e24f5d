// mw_full_energy.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine:
// compute_model_energy (molint.F90:407-499): k_model_energy, k_sum_partials.
#pragma once

#include "mw_common.hip.h"
#include <type_traits>

namespace mw {

// =====================================================================================
// Full-box energy.
//
// Per atom i with in-range neighbours j (r_ij < rc), unit vectors u_j, weights
// g_j = exp(gamma*sigma/(r_ij - a*sigma)):
//   E_i = 1/2 sum_j phi2(r_ij) + lambda*eps * sum_{j<k} g_j g_k (u_j.u_k - cos0)^2
// The reference walks all pairs j<k (molint.F90:467-487).  Here the triplet sum
// comes from moments accumulated in ONE pass over the neighbours:
//   S0 = sum g, S1 = sum g u, S2 = sum g u u^T, Q = sum g^2
//   sum_{j<k} g_j g_k (c_jk - c0)^2 = 1/2 [ (|S2|_F^2 - Q) - 2 c0 (|S1|^2 - Q) + c0^2 (S0^2 - Q) ]
// (c_jj = 1 gives the three Q terms).  No per-neighbour storage, so nothing
// spills and the loop is O(neighbours).  Cancellation is harmless at this
// tolerance: the terms are O(S0^2) ~ 0.4 while the parity bar is 1e-10 relative
// on E_i ~ 2e-2 -- fourteen digits are left over.
//
// Both exponentials of a pair come from one (pair_terms, mw_common.hip.h).
//
// One molecule per lane, and the lanes of a wavefront hold molecules that do the SAME amount of work:
// the list builder sorts the molecules of a box by (in-range neighbours, row length) at build time
// (k_list_order) and stores the slot-major list in that order -- column t of the list belongs to
// molecule order[t], nns[t] is its row length, cmax[t / 64] the longest row of its group of 64.  The order
// is a layout hint only: every in-range decision is taken here, on the current positions.
//
// Divergence control inside a wavefront: phase 1 runs the cheap distance test over all list slots
// (list read eight slots at a time, so eight coalesced loads are in flight) and parks the in-range
// entries in a per-thread LDS queue; phase 2 runs the expensive part only over that queue, so a
// wavefront's trip count is its largest in-range count -- which the sorted order keeps within one of
// the mean (7 instead of 10 passes on the thermal 4096-molecule boxes) -- rather than its longest row.
//
// LDSPOS = true : one workgroup stages the whole box's positions in LDS
//                 (N*24 B: 96 KiB at N = 4096) and gathers r_j from there (layouts: LdsVecs below).
// LDSPOS = false: r_j gathered from global memory (L2-resident for the sizes
//                 that do not fit LDS, e.g. 786 KiB at N = 32768).
//   grid = (nsplit, nboxes_in_launch); each block takes list columns [split*chunk, ...), chunk % 64 == 0
// =====================================================================================
struct AtomSum { double e; int cnt; };   // energy of one molecule, its in-range neighbours

constexpr int kQCap = 12;   // in-range entries per molecule parked in LDS between the two phases

constexpr double kAepsBSig4 = kAeps * kBigB * kSigSq * kSigSq;   // A eps B sigma^4 (molint.F90:460)

// A molecule's running sums over its in-range neighbours and what becomes of them: ONE piece of arithmetic for every routine that
// adds a neighbour (atom_energy below; the Monte Carlo driver's volume moves spread over several wavefronts, mw_sweep.hip.h, whose
// sum must be atom_energy's bit for bit).  add(): neighbour at d = r_j + ivect - r_i with 1/r, e1, g of pair_terms (:456-462).
// |d|^2 with the multiply-adds spelled out: wherever a routine squares a separation that another routine squares too (the two must
// agree to the bit -- see MomentSums), the fusion is not left to the compiler's contraction, which decides per inlined copy.
__device__ __forceinline__ double dist2(double dx, double dy, double dz) { return __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)); }

struct MomentSums {
    double e2 = 0.0, S0 = 0.0, Q = 0.0, S1x = 0.0, S1y = 0.0, S1z = 0.0;
    double Sxx = 0.0, Syy = 0.0, Szz = 0.0, Sxy = 0.0, Sxz = 0.0, Syz = 0.0;
    __device__ __forceinline__ void add(double dx, double dy, double dz, double rinv, double e1, double g)
    {
        // (no contraction in here: g arrives as the PRODUCT t^4 t^2 when pair_terms is inlined next to this, and "S0 += g" then fuses
        //  into fma(t^4, t^2, S0) -- one rounding fewer than where g comes out of memory.  The driver's split volume move found it:
        //  one ulp of a molecule's energy between two routines made of the same source lines.)
#pragma clang fp contract(off)
        const double ri2 = rinv * rinv, ri4 = ri2 * ri2;
        e2 = __builtin_fma(fma_sc(ri4, kAepsBSig4, -kAeps), e1, e2);                  // A eps (B (sigma/r)^4 - 1) e1  :460-461
        // moments of g u with u = d / r: S1 += (g/r) d, S2 += (g/r^2) d d^T
        const double w1 = g * rinv, w2 = g * ri2;
        const double hx = w2 * dx, hy = w2 * dy, hz = w2 * dz;
        S0 += g;  Q = __builtin_fma(g, g, Q);
        S1x = __builtin_fma(w1, dx, S1x); S1y = __builtin_fma(w1, dy, S1y); S1z = __builtin_fma(w1, dz, S1z);
        Sxx = __builtin_fma(hx, dx, Sxx); Syy = __builtin_fma(hy, dy, Syy); Szz = __builtin_fma(hz, dz, Szz);
        Sxy = __builtin_fma(hx, dy, Sxy); Sxz = __builtin_fma(hx, dz, Sxz); Syz = __builtin_fma(hy, dz, Syz);
    }
    // the molecule's energy (:464,483); its moments to mom_out (per lane, or nullptr; 16-byte stores: a divergent store costs its 64
    // addresses, whatever their width)
    __device__ __forceinline__ double finish(int cnt, double* __restrict__ mom_out) const
    {
#pragma clang fp contract(off)
        // (every multiply-add explicit: one rounding sequence wherever this is inlined)
        const double D2 = dist2(Sxx, Syy, Szz), O2 = dist2(Sxy, Sxz, Syz);
        const double F2 = __builtin_fma(2.0, O2, D2);
        const double F1 = dist2(S1x, S1y, S1z);
        const double A = F2 - Q, B = F1 - Q, C = __builtin_fma(S0, S0, -Q);
        const double T = 0.5 * __builtin_fma(kCos0 * kCos0, C, __builtin_fma(-2.0 * kCos0, B, A));
        if (mom_out) {
            double2* m2 = reinterpret_cast<double2*>(mom_out);
            m2[0] = make_double2(S0, S1x); m2[1] = make_double2(S1y, S1z); m2[2] = make_double2(Sxx, Syy);
            m2[3] = make_double2(Sxy, Sxz); m2[4] = make_double2(Syz, (double)cnt);
        }
        return __builtin_fma(0.5, e2, kLamEps * T);
    }
};

// The list streams through buffer loads: the row of slot s is a scalar offset, the column a per-lane offset, and a
// lane without a column (offset kNoColumn) reads 0 -- no address arithmetic in vector registers at all.
using ListRsrc = __amdgpu_buffer_rsrc_t;
constexpr uint32_t kNoColumn = 0x7fffffffu;
__device__ __forceinline__ ListRsrc list_rsrc(const uint32_t* L, int N, int S)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(L), 0, (int)((size_t)N * S * sizeof(uint32_t)), 0x00020000);
}
__device__ __forceinline__ uint32_t list_load(ListRsrc rs, uint32_t column_bytes, int slot, int N, int S)
{
    const int s = slot < S ? slot : S - 1;                                   // (uniform; such a slot is never live)
    return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)column_bytes, s * N * (int)sizeof(uint32_t), 0);
}

// `queue` points at this thread's column of an LDS array [QCAP + 1][BLOCK] (entry q at queue[q*BLOCK]:
// consecutive threads, consecutive banks; row QCAP takes what does not fit -- those entries are found again by a rescan of
// the column and accumulated after the queued ones, in list order: the sum is the same whatever QCAP is).  `col` is the byte offset of the
// thread's list column (kNoColumn: none), `mol` the molecule it belongs to, `n` its row length, `nmax`
// (wave-uniform) the longest row among the wavefront's columns and `c0min` (wave-uniform) the number of
// leading slots that hold a central-image entry in EVERY column of the wavefront (columns list their central
// entries first and are zero-padded to nmax): those slots skip the image-vector gather.  The list is read eight
// slots at a time and ONE CHUNK AHEAD: `cur` arrives holding this column's first eight entries; while a chunk is
// being tested the next one -- of this column, or the first of the thread's next column `col_next` -- is already
// in flight, so the HBM latency of the list stream hides behind the LDS gathers and distance tests.
// LEAN = true (the Monte Carlo driver's volume moves, where this routine is a guest in a kernel sized for something else):
// no chunk-ahead list prefetch and no double-buffered gathers -- twenty vector registers fewer, `cur` / `col_next` unused.
// `ent` (LEAN only): where a lane's list entries come from when not from the slot-major list in global memory -- the Monte Carlo driver's
// small walkers keep their rows in LDS (`ent(s)` = entry s of this lane's molecule, 0 past its end; then rs / col are unused).
struct ListFromGlobal { static constexpr bool kGlobal = true; __device__ uint32_t operator()(int) const { return 0u; } };
template <int BLOCK, bool BATCH4, bool LEAN = false, int QCAP = kQCap, typename PosFn, typename IvFn, typename EntFn = ListFromGlobal>
__device__ __forceinline__ AtomSum atom_energy(ListRsrc rs, uint32_t col, uint32_t col_next, int mol, int n, int nmax, int c0min,
                                               int N, int S, uint32_t* __restrict__ queue, PosFn getpos, IvFn getiv,
                                               uint32_t (&cur)[8], double* __restrict__ mom_out = nullptr, EntFn ent = ListFromGlobal())
{
    constexpr bool kEnt = !std::is_same<EntFn, ListFromGlobal>::value;
    static_assert(!kEnt || LEAN, "list entries from a functor: the lean variant only");
    auto entry = [&](int s) -> uint32_t { if constexpr (kEnt) return ent(s); else return list_load(rs, col, s, N, S); };
    double xi, yi, zi;
    getpos(mol, xi, yi, zi);

    // phase 1: cheap distance test over all list slots; the in-range entries are parked in LDS.
    int cnt = 0;
    for (int s0 = 0; s0 < nmax || s0 == 0; s0 += 8) {
        uint32_t nxt[8];
        if constexpr (LEAN) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = entry(s0 + u);
        } else {
            const bool last = s0 + 8 >= nmax;                 // wave-uniform
            const uint32_t pc = last ? col_next : col;        // whose chunk comes next
            const int ps = last ? 0 : s0 + 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) nxt[u] = list_load(rs, pc, ps + u, N, S);
        }
        if constexpr (BATCH4) {
        // four slots at a time: their gathers are issued together, then the four tests (a slot past the longest
        // row holds whatever the prefetch brought -- never live, and an LDS gather cannot fault)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sb = s0 + 4 * h;
            if (sb < nmax) {                                  // wave-uniform
                double d[4][3];
                if (sb + 4 <= c0min) {                        // wave-uniform: central image only, its vector is exactly 0 (molint.F90:197)
#pragma unroll
                    for (int u = 0; u < 4; ++u) getpos((int)(cur[4 * h + u] & kJMask), d[u][0], d[u][1], d[u][2]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { d[u][0] -= xi; d[u][1] -= yi; d[u][2] -= zi; }
                } else {
                    double iv[4][3];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        getpos((int)(cur[4 * h + u] & kJMask), d[u][0], d[u][1], d[u][2]);
                        getiv((int)(cur[4 * h + u] >> kJBits), iv[u][0], iv[u][1], iv[u][2]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {                                             // molint.F90:447,450
                        d[u][0] = (d[u][0] + iv[u][0]) - xi; d[u][1] = (d[u][1] + iv[u][1]) - yi; d[u][2] = (d[u][2] + iv[u][2]) - zi;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double r2 = dist2(d[u][0], d[u][1], d[u][2]);
                    if (sb + u < n && r2 < kRcSq) {                                           // :454
                        queue[(cnt < QCAP ? cnt : QCAP) * BLOCK] = cur[4 * h + u];
                        ++cnt;
                    }
                }
            }
        }
        } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < nmax) {                              // wave-uniform
                const uint32_t e = cur[u];                    // past the row's end: 0 = molecule 0, central image; not live
                double xj, yj, zj;
                getpos((int)(e & kJMask), xj, yj, zj);
                double dx, dy, dz;
                if (s0 + u < c0min) {                         // wave-uniform: the central image vector is exactly 0 (molint.F90:197)
                    dx = xj - xi; dy = yj - yi; dz = zj - zi;
                } else {
                    double ix, iy, iz;
                    getiv((int)(e >> kJBits), ix, iy, iz);
                    dx = (xj + ix) - xi; dy = (yj + iy) - yi; dz = (zj + iz) - zi;            // molint.F90:447,450
                }
                const double r2 = dist2(dx, dy, dz);
                if (s0 + u < n && r2 < kRcSq) {                                               // :454
                    queue[(cnt < QCAP ? cnt : QCAP) * BLOCK] = e;
                    ++cnt;
                }
            }
        }
        }
        if constexpr (!LEAN) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        }
    }

    // phase 2: pair term and moments over the in-range entries only
    MomentSums ms;
    // The gathers of entry q+1 are issued before entry q is evaluated (one LDS round trip hidden per entry).
    auto gather = [&](uint32_t e, double (&v)[6]) {
        getpos((int)(e & kJMask), v[0], v[1], v[2]);
        getiv((int)(e >> kJBits), v[3], v[4], v[5]);
    };
    auto accumulate = [&](const double (&v)[6]) {
        const double dx = (v[0] + v[3]) - xi, dy = (v[1] + v[4]) - yi, dz = (v[2] + v[5]) - zi;
        const double r2 = dist2(dx, dy, dz);
        double rinv, e1, g;
        pair_terms(r2, rinv, e1, g);                                                  // :456-462
        ms.add(dx, dy, dz, rinv, e1, g);
    };
    const int nq = cnt < QCAP ? cnt : QCAP;
    if constexpr (LEAN && !BATCH4) {
        for (int q = 0; q < nq; ++q) { double v[6]; gather(queue[q * BLOCK], v); accumulate(v); }
    } else if (nq > 0) {
        double va[6], vb[6];
        gather(queue[0], va);
        for (int q = 0; q < nq; ++q) {
            const uint32_t en = queue[(q + 1 < nq ? q + 1 : q) * BLOCK];
            gather(en, vb);
            accumulate(va);
#pragma unroll
            for (int c = 0; c < 6; ++c) va[c] = vb[c];
        }
    }
    if (cnt > QCAP) {           // rare (dense configurations): rescan the column for the in-range entries the queue had no room for
        int seen = 0;
        for (int s = 0; s < n; ++s) {
            double v[6];
            gather(entry(s), v);
            const double dx = (v[0] + v[3]) - xi, dy = (v[1] + v[4]) - yi, dz = (v[2] + v[5]) - zi;
            if (dist2(dx, dy, dz) < kRcSq) { if (seen >= QCAP) accumulate(v); ++seen; }
        }
    }
    AtomSum out;
    out.e  = ms.finish(cnt, mom_out);
    out.cnt = cnt;
    return out;
}

constexpr int kMaxGroups = 128;          // groups of 64 list columns per workgroup (chunk <= 8192)
constexpr int kStageHold = 1;            // staging tickets a wavefront holds in registers (12 doubles per lane each; 2 or 3 spill at 1024 threads: measured slower)
constexpr int kStageTicket = 64 * 12;    // doubles of the next box per staging ticket

// The wavefronts of a workgroup draw groups of 64 list columns from an LDS ticket, heaviest group first (the
// columns are sorted by work, ascending): a wavefront that drew cheap groups serves more of them, and the
// workgroup's tail is made of the cheapest groups.
//
// A workgroup is PERSISTENT over boxes: it takes boxes blockIdx.y, blockIdx.y + gridDim.y, ... of the launch (the host
// makes gridDim.y the number of compute units when whole boxes are staged in LDS -- one such workgroup fills a CU's LDS
// -- and the number of boxes otherwise, which is the plain one-box-per-workgroup launch).  The next box cannot be staged
// while the current one is in use (2 x 96 KiB do not fit), but it can be READ: a wavefront that has run out of groups
// draws staging tickets (768 doubles of the next box each) and holds them in registers -- free now that its evaluation
// is over -- until the slowest wavefront arrives at the barrier; then the registers go to LDS.  The HBM round trip of
// the staging overlaps the workgroup's tail instead of following it; so do the first list reads of the next box.
// MOMOUT: the build that also leaves every molecule's moments behind (`mom`); the plain build carries none of its registers.
template <bool LDSPOS, int BLOCK, int LAYOUT, bool BATCH4 = false, bool MOMOUT = false>
__global__ __launch_bounds__(BLOCK)
void k_model_energy(const double* __restrict__ pos, const double* __restrict__ ivect,
                    const int* __restrict__ nivect, const uint32_t* __restrict__ list,
                    const int* __restrict__ order, const int* __restrict__ nns, const int* __restrict__ cmax,
                    double* __restrict__ partial, unsigned long long* __restrict__ cpartial,
                    double* __restrict__ energy, unsigned long long* __restrict__ counts,
                    int N, int S, int ivcap, int box0, int nsplit, int chunk, int count,
                    double* __restrict__ mom = nullptr,   // [box][N][kMomStride]: every molecule's moments too (the single-move kernel's moment path)
                    int write_energy = 1)                 // 0: a pass for the moments only -- energies and counts of the boxes stay as they are
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double gsum[kMaxGroups];                  // per group of 64 columns: summed in group order at the end,
    __shared__ unsigned long long red_p[BLOCK / 64], red_t[BLOCK / 64];   // so the energy does not depend on who drew what
    __shared__ int s_ticket;
    __shared__ int s_stage[2];                           // staging tickets, one counter per parity of the workgroup's box number

    const int split = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ngroups = (N + 63) >> 6;

    // dynamic LDS: [positions when LDSPOS][image vectors][in-range queue kQCap x BLOCK u32]
    double* spos = smem;
    double* siv = smem + (LDSPOS ? lds_vec_bytes((size_t)N) / 8 : 0);
    uint32_t* queue = reinterpret_cast<uint32_t*>(siv + lds_vec_bytes((size_t)ivcap) / 8) + tid;   // [kQCap + 1][BLOCK]
    const int a0 = split * chunk;                        // a multiple of 64: whole groups (at most kMaxGroups)
    const int a1 = min(N, a0 + chunk);
    const int g0 = a0 >> 6, G = ((a1 + 63) >> 6) - g0;   // this workgroup's groups: g0 .. g0 + G - 1
    const int w0 = __builtin_amdgcn_readfirstlane(wid);
    const int first_grp = w0 < G ? g0 + G - 1 - w0 : -1; // first tickets: one per wavefront (wave-uniform, in a scalar register)
    const int nstage = LDSPOS ? (3 * N + kStageTicket - 1) / kStageTicket : 0;

    // first list reads of this wavefront's first group: nothing in them depends on the staged box, so they are in flight
    // while it is being staged (behind the barrier they would be one more exposed HBM round trip per workgroup)
    uint32_t cur[8];
    int n_cur = 0, mol = 0;
    uint32_t col = kNoColumn;
    auto first_reads = [&](int b) {
        col = kNoColumn; n_cur = 0; mol = 0;
        if (first_grp >= 0) {
            const int t = first_grp * 64 + lane;
            if (t < a1) { col = (uint32_t)t * 4u; n_cur = nns[(size_t)b * N + t]; mol = order[(size_t)b * N + t]; }
            const ListRsrc r0 = list_rsrc(list + (size_t)b * S * N, N, S);
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = list_load(r0, col, u, N, S);
        }
    };
    int bi = blockIdx.y;                                 // the workgroup's current box, counted within the launch
    {   // the workgroup's first box: staged straight from global memory, all of a thread's loads in flight together
        const int b = box0 + bi;
        const double* IV = ivect + (size_t)b * ivcap * 3;
        const int niv = nivect[b];
        first_reads(b);
        const double iv_first = stage_iv_begin<BLOCK>(IV, niv, tid);
        if (LDSPOS) stage_vecs<LAYOUT, BLOCK>(spos, pos + (size_t)b * N * 3, N, N, tid);
        stage_iv_end<LAYOUT, BLOCK>(siv, IV, niv, ivcap, tid, iv_first);
        if (tid == 0) { s_ticket = BLOCK / 64; s_stage[0] = 0; s_stage[1] = 0; }
    }
    __syncthreads();

    const LdsVecs<LAYOUT> vpos{spos, N}, viv{siv, ivcap};
    auto getiv = [&](int k, double& x, double& y, double& z) { viv.get(k, x, y, z); };

    for (int it = 0;; ++it) {                            // the workgroup's boxes
        const int b = box0 + bi;
        const bool more = LDSPOS && bi + (int)gridDim.y < count;   // workgroup-uniform (boxes gathered through L2 are launched one per workgroup)
        const double* P  = pos + (size_t)b * N * 3;
        const int* ORD = order + (size_t)b * N;
        const int* NNS = nns + (size_t)b * N;
        const int* CM = cmax + (size_t)b * ngroups;
        auto getpos = [&](int j, double& x, double& y, double& z) {
            if constexpr (LDSPOS) vpos.get(j, x, y, z);
            else { const double* p = P + 3 * (size_t)j; x = p[0]; y = p[1]; z = p[2]; }
        };
        const ListRsrc rs = list_rsrc(list + (size_t)b * S * N, N, S);

        unsigned int np = 0, nt = 0;                     // directed in-range pairs, i-centred triplets of this lane's molecules
        int grp = first_grp;
        while (grp >= 0) {                               // wave-uniform
            int tk = 0;
            if (lane == 0) tk = __hip_atomic_fetch_add(&s_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            tk = __builtin_amdgcn_readfirstlane(tk);
            const int gnext = tk < G ? g0 + G - 1 - tk : -1;
            const bool act = col != kNoColumn;
            uint32_t col_next = kNoColumn;
            int n_next = 0, mol_next = 0;
            if (gnext >= 0) {                                                    // one group ahead, like the list chunks
                const int tn = gnext * 64 + lane;
                if (tn < a1) { col_next = (uint32_t)tn * 4u; n_next = NNS[tn]; mol_next = ORD[tn]; }
            }
            const int cm = __builtin_amdgcn_readfirstlane(CM[grp]);
            double* mo = (MOMOUT && act) ? mom + ((size_t)b * N + mol) * kMomStride : nullptr;
            AtomSum a = atom_energy<BLOCK, BATCH4>(rs, col, col_next, mol, act ? (n_cur & 0xff) : 0, cm & 0xff, cm >> 8, N, S,
                                           queue, getpos, getiv, cur, mo);
            if (act) { np += (unsigned int)a.cnt; nt += (unsigned int)(a.cnt * (a.cnt - 1) / 2); }
            const double ge = dpp_wave_sum(act ? a.e : 0.0);             // fixed tree; total in lane 63
            if (lane == 63) gsum[grp - g0] = ge;
            n_cur = n_next; mol = mol_next; col = col_next; grp = gnext;
        }
        const unsigned long long wp = wave_sum_u64(np), wt = wave_sum_u64(nt);
        if (lane == 0) { red_p[wid] = wp; red_t[wid] = wt; }

        // out of groups.  If the workgroup has another box: its first list reads, its image vectors, and as many staging
        // tickets as fit in registers -- all in flight while the other wavefronts finish
        const int bn = box0 + bi + (int)gridDim.y;
        double hold[kStageHold][12], iv_next = 0.0;
        int held[kStageHold];
        int nivn = 0;
#pragma unroll
        for (int q = 0; q < kStageHold; ++q) held[q] = -1;
        if (more) {
            first_reads(bn);
            nivn = nivect[bn];
            iv_next = stage_iv_begin<BLOCK>(ivect + (size_t)bn * ivcap * 3, nivn, tid);
            if constexpr (LDSPOS) {
                const double* Pn = pos + (size_t)bn * N * 3;
#pragma unroll
                for (int q = 0; q < kStageHold; ++q) {
                    int tk = 0;
                    if (lane == 0) tk = __hip_atomic_fetch_add(&s_stage[it & 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    tk = __builtin_amdgcn_readfirstlane(tk);
                    held[q] = tk < nstage ? tk : -1;
                    if (held[q] >= 0) {                  // (the box's last ticket may read past its end: d_pos is padded by one ticket)
                        const double* src = Pn + (size_t)held[q] * kStageTicket + lane;
#pragma unroll
                        for (int k = 0; k < 12; ++k) hold[q][k] = src[k * 64];
                    }
                }
            }
        }
        __syncthreads();                                 // the box is done: its sums are in LDS, its staged positions free
        if (tid == 0) {
            double e = 0.0; unsigned long long p = 0, t = 0;
            for (int k = 0; k < G; ++k) e += gsum[k];
            for (int w = 0; w < BLOCK / 64; ++w) { p += red_p[w]; t += red_t[w]; }
            if (!write_energy) {
            } else if (nsplit == 1) {                    // one workgroup per box: model_energy(ils) and its counts directly
                energy[b] = e; counts[2 * b] = p; counts[2 * b + 1] = t;
            } else {                                     // a split box: k_sum_partials adds the partials in split order
                const size_t o = (size_t)(b) * nsplit + split;
                partial[o] = e; cpartial[2 * o] = p; cpartial[2 * o + 1] = t;
            }
            s_ticket = BLOCK / 64;
            s_stage[(it + 1) & 1] = 0;                   // (the counter of THIS box is still in use below)
        }
        if (!more) break;
        stage_iv_end<LAYOUT, BLOCK>(siv, ivect + (size_t)bn * ivcap * 3, nivn, ivcap, tid, iv_next);
        if constexpr (LDSPOS) {
            const double* Pn = pos + (size_t)bn * N * 3;
#pragma unroll
            for (int q = 0; q < kStageHold; ++q) {
                if (held[q] >= 0) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        const int t = held[q] * kStageTicket + k * 64 + lane;
                        if (t < 3 * N) spos[LdsVecs<LAYOUT>::slot(t, N)] = hold[q][k];
                    }
                }
            }
            for (;;) {                                   // tickets nobody had registers for (all wavefronts arrived together, or a large box)
                int tk = 0;
                if (lane == 0) tk = __hip_atomic_fetch_add(&s_stage[it & 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                tk = __builtin_amdgcn_readfirstlane(tk);
                if (tk >= nstage) break;
                double v[12];
                const double* src = Pn + (size_t)tk * kStageTicket + lane;
#pragma unroll
                for (int k = 0; k < 12; ++k) v[k] = src[k * 64];
#pragma unroll
                for (int k = 0; k < 12; ++k) { const int t = tk * kStageTicket + k * 64 + lane; if (t < 3 * N) spos[LdsVecs<LAYOUT>::slot(t, N)] = v[k]; }
            }
        }
        __syncthreads();
        bi += (int)gridDim.y;
    }
}

// Fixed-order sum of the per-workgroup partials of SPLIT boxes (nsplit > 1): model_energy(ils) and its counts.
// (A "last workgroup sums" scheme inside k_model_energy needs agent-scope release/acquire around a counter -- on
// gfx950 an L2 write-back and invalidate per workgroup -- which took the 64 x 32768 launch from 0.19 to 0.46 ms by
// throwing away the L2 lines its position gathers live on; a second tiny launch is cheaper.)
__global__ __launch_bounds__(64)
void k_sum_partials(const double* __restrict__ partial, const unsigned long long* __restrict__ cpartial,
                    double* __restrict__ energy, unsigned long long* __restrict__ counts,
                    int box0, int count, int nsplit)
{
    // one wavefront per box: lane l adds the partials l, l + 64, ... in that order, then a fixed DPP tree -- the same
    // sum whatever the launch looked like, and one coalesced read instead of a chain of nsplit dependent loads per box
    const int b = box0 + (int)blockIdx.x, lane = threadIdx.x;
    if ((int)blockIdx.x >= count) return;
    double e = 0.0; unsigned long long p = 0, q = 0;
    for (int s = lane; s < nsplit; s += 64) {
        const size_t o = (size_t)b * nsplit + s;
        e += partial[o]; p += cpartial[2 * o]; q += cpartial[2 * o + 1];
    }
    e = readlane_f64(dpp_wave_sum(e), 63);
    p = wave_sum_u64(p); q = wave_sum_u64(q);
    if (lane == 0) { energy[b] = e; counts[2 * b] = p; counts[2 * b + 1] = q; }
}

}  // namespace mw
