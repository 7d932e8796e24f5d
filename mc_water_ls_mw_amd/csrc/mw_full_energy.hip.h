// mw_full_energy.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine:
// compute_model_energy (molint.F90:407-499): k_model_energy, k_sum_partials.
#pragma once

#include "mw_common.hip.h"

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
//                 (N*24 B: 96 KiB at N = 4096) and gathers r_j from there.
// LDSPOS = false: r_j gathered from global memory (L2-resident for the sizes
//                 that do not fit LDS, e.g. 786 KiB at N = 32768).
//   grid = (nsplit, nboxes_in_launch); each block takes list columns [split*chunk, ...), chunk % 64 == 0
// =====================================================================================
struct AtomSum { double e; unsigned long long np, nt; };

constexpr int kQCap = 12;   // in-range entries per molecule parked in LDS between the two phases

constexpr double kAepsBSig4 = kAeps * kBigB * kSigSq * kSigSq;   // A eps B sigma^4 (molint.F90:460)

// `queue` points at this thread's column of an LDS array [kQCap][BLOCK] (entry q at queue[q*BLOCK]:
// consecutive threads, consecutive banks).  `t` is the thread's list column (-1: none), `mol` the molecule
// it belongs to, `n` its row length and `nmax` (wave-uniform) the longest row among the wavefront's
// columns.  The list is read eight slots at a time and ONE CHUNK AHEAD: `cur` arrives holding this
// column's first eight entries; while a chunk is being tested the next one -- of this column, or the first
// of the thread's next column `tnext` -- is already in flight, so the HBM latency of the list stream hides
// behind the LDS gathers and distance tests.
template <int BLOCK, typename PosFn, typename IvFn>
__device__ __forceinline__ AtomSum atom_energy(int t, int mol, int n, int nmax, const uint32_t* __restrict__ L, int N, int S,
                                               uint32_t* __restrict__ queue, PosFn getpos, IvFn getiv,
                                               uint32_t (&cur)[8], int tnext)
{
    double xi, yi, zi;
    getpos(mol, xi, yi, zi);

    // phase 1: cheap distance test over all list slots; the in-range entries are parked in LDS.
    int cnt = 0;
    unsigned long long over = 0ull;             // in-range slots beyond the LDS queue (re-read later)
    for (int s0 = 0; s0 < nmax || s0 == 0; s0 += 8) {
        uint32_t nxt[8];
        const bool last = s0 + 8 >= nmax;                     // wave-uniform
        const int pt = last ? tnext : t;                      // whose chunk comes next
        const int ps = last ? 0 : s0 + 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt[u] = (pt >= 0 && ps + u < S) ? L[(size_t)(ps + u) * N + pt] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < nmax) {                              // wave-uniform
                const bool live = s0 + u < n;
                const uint32_t e = live ? cur[u] : 0u;        // a slot past the row's end: molecule 0, central image, ignored
                double xj, yj, zj, ix, iy, iz;
                getpos((int)(e & kJMask), xj, yj, zj);
                getiv((int)(e >> kJBits), ix, iy, iz);
                const double dx = (xj + ix) - xi, dy = (yj + iy) - yi, dz = (zj + iz) - zi;   // molint.F90:447,450
                const double r2 = dx * dx + dy * dy + dz * dz;
                if (live && r2 < kRcSq) {                                                     // :454
                    if (cnt < kQCap) queue[cnt * BLOCK] = e;
                    else over |= 1ull << (s0 + u);
                    ++cnt;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
    }

    // phase 2: pair term and moments over the in-range entries only
    double e2 = 0.0, S0 = 0.0, Q = 0.0, S1x = 0.0, S1y = 0.0, S1z = 0.0;
    double Sxx = 0.0, Syy = 0.0, Szz = 0.0, Sxy = 0.0, Sxz = 0.0, Syz = 0.0;
    // The gathers of entry q+1 are issued before entry q is evaluated (one LDS round trip hidden per entry).
    auto gather = [&](uint32_t e, double (&v)[6]) {
        getpos((int)(e & kJMask), v[0], v[1], v[2]);
        getiv((int)(e >> kJBits), v[3], v[4], v[5]);
    };
    auto accumulate = [&](const double (&v)[6]) {
        const double dx = (v[0] + v[3]) - xi, dy = (v[1] + v[4]) - yi, dz = (v[2] + v[5]) - zi;
        const double r2 = dx * dx + dy * dy + dz * dz;
        double rinv, e1, g;
        pair_terms(r2, rinv, e1, g);                                                  // :456-462
        const double ri2 = rinv * rinv, ri4 = ri2 * ri2;
        e2 = __builtin_fma(fma_sc(ri4, kAepsBSig4, -kAeps), e1, e2);                  // A eps (B (sigma/r)^4 - 1) e1  :460-461
        // moments of g u with u = d / r: S1 += (g/r) d, S2 += (g/r^2) d d^T
        const double w1 = g * rinv, w2 = g * ri2;
        const double hx = w2 * dx, hy = w2 * dy, hz = w2 * dz;
        S0 += g;  Q = __builtin_fma(g, g, Q);
        S1x = __builtin_fma(w1, dx, S1x); S1y = __builtin_fma(w1, dy, S1y); S1z = __builtin_fma(w1, dz, S1z);
        Sxx = __builtin_fma(hx, dx, Sxx); Syy = __builtin_fma(hy, dy, Syy); Szz = __builtin_fma(hz, dz, Szz);
        Sxy = __builtin_fma(hx, dy, Sxy); Sxz = __builtin_fma(hx, dz, Sxz); Syz = __builtin_fma(hy, dz, Syz);
    };
    const int nq = cnt < kQCap ? cnt : kQCap;
    if (nq > 0) {
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
    while (over) {
        const int s = __ffsll((long long)over) - 1;
        over &= over - 1ull;
        double v[6];
        gather(L[(size_t)s * N + t], v);
        accumulate(v);
    }
    const double F2 = Sxx * Sxx + Syy * Syy + Szz * Szz + 2.0 * (Sxy * Sxy + Sxz * Sxz + Syz * Syz);
    const double F1 = S1x * S1x + S1y * S1y + S1z * S1z;
    const double T = 0.5 * ((F2 - Q) - 2.0 * kCos0 * (F1 - Q) + kCos0 * kCos0 * (S0 * S0 - Q));
    AtomSum out;
    out.e  = 0.5 * e2 + kLamEps * T;                                                   // :464,483
    out.np = (unsigned long long)cnt;
    out.nt = (unsigned long long)(cnt * (cnt - 1) / 2);
    return out;
}

template <bool LDSPOS, int BLOCK>
__global__ __launch_bounds__(BLOCK)
void k_model_energy(const double* __restrict__ pos, const double* __restrict__ ivect,
                    const int* __restrict__ nivect, const uint32_t* __restrict__ list,
                    const int* __restrict__ order, const int* __restrict__ nns, const int* __restrict__ cmax,
                    double* __restrict__ partial, unsigned long long* __restrict__ cpartial,
                    int N, int S, int ivcap, int box0, int nsplit, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_e[BLOCK / 64];
    __shared__ unsigned long long red_p[BLOCK / 64], red_t[BLOCK / 64];

    const int b = box0 + blockIdx.y;
    const int split = blockIdx.x;
    const int tid = threadIdx.x;
    const int ngroups = (N + 63) >> 6;
    const double* P  = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    const uint32_t* L = list + (size_t)b * S * N;
    const int* ORD = order + (size_t)b * N;
    const int* NNS = nns + (size_t)b * N;
    const int* CM = cmax + (size_t)b * ngroups;
    const int niv = nivect[b];

    // dynamic LDS: [positions when LDSPOS][image vectors][in-range queue kQCap x BLOCK u32]; the positions
    // sit at offset 0 so that a gather's address is one multiply and the ds_read offsets are immediates.
    double* spos = smem;
    double* siv = smem + (LDSPOS ? 3 * (size_t)N : 0);
    uint32_t* queue = reinterpret_cast<uint32_t*>(siv + (size_t)ivcap * 3) + tid;
    for (int t = tid; t < niv * 3; t += BLOCK) siv[t] = IV[t];
    if (LDSPOS) {
        for (int t = tid; t < 3 * N; t += BLOCK) spos[t] = P[t];   // flat, fully coalesced copy
    }
    __syncthreads();

    auto getiv = [&](int k, double& x, double& y, double& z) { x = siv[3 * k]; y = siv[3 * k + 1]; z = siv[3 * k + 2]; };
    auto getpos = [&](int j, double& x, double& y, double& z) {
        const double* p = LDSPOS ? (spos + 3 * (size_t)j) : (P + 3 * (size_t)j);
        x = p[0]; y = p[1]; z = p[2];
    };

    double esum = 0.0;
    unsigned long long np = 0, nt = 0;
    const int a0 = split * chunk;                        // a multiple of 64: a wavefront's columns are one group
    const int a1 = min(N, a0 + chunk);
    const int wbase = __builtin_amdgcn_readfirstlane(a0 + (tid & ~63));
    uint32_t cur[8];
    int n_cur = 0, mol = 0;
    {
        const int t = a0 + tid;
        if (t < a1) {
            n_cur = NNS[t]; mol = ORD[t];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = u < S ? L[(size_t)u * N + t] : 0u;
        }
    }
    for (int base = wbase; base < a1; base += BLOCK) {           // wave-uniform
        const int t = base + (tid & 63);
        const int tn = t + BLOCK;
        const bool act = t < a1;
        const int tnext = tn < a1 ? tn : -1;
        int n_next = 0, mol_next = 0;
        if (tnext >= 0) { n_next = NNS[tnext]; mol_next = ORD[tnext]; }      // one column ahead, like the list chunks
        const int nmax = CM[base >> 6];
        AtomSum a = atom_energy<BLOCK>(act ? t : -1, mol, act ? n_cur : 0, nmax, L, N, S, queue, getpos, getiv, cur, tnext);
        if (act) { esum += a.e; np += a.np; nt += a.nt; }
        n_cur = n_next; mol = mol_next;
    }

    esum = wave_sum(esum); np = wave_sum_u64(np); nt = wave_sum_u64(nt);
    const int wid = tid >> 6;
    if ((tid & 63) == 0) { red_e[wid] = esum; red_p[wid] = np; red_t[wid] = nt; }
    __syncthreads();
    if (tid == 0) {
        double e = 0.0; unsigned long long p = 0, t = 0;
        for (int w = 0; w < BLOCK / 64; ++w) { e += red_e[w]; p += red_p[w]; t += red_t[w]; }
        const size_t o = (size_t)(b) * nsplit + split;
        partial[o] = e; cpartial[2 * o] = p; cpartial[2 * o + 1] = t;
    }
}

// Fixed-order sum of the per-block partials: model_energy(ils) and its counts.
__global__ void k_sum_partials(const double* __restrict__ partial, const unsigned long long* __restrict__ cpartial,
                               double* __restrict__ energy, unsigned long long* __restrict__ counts,
                               int box0, int count, int nsplit)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int b = box0 + t;
    double e = 0.0; unsigned long long p = 0, q = 0;
    for (int s = 0; s < nsplit; ++s) {
        const size_t o = (size_t)b * nsplit + s;
        e += partial[o]; p += cpartial[2 * o]; q += cpartial[2 * o + 1];
    }
    energy[b] = e; counts[2 * b] = p; counts[2 * b + 1] = q;
}

}  // namespace mw
