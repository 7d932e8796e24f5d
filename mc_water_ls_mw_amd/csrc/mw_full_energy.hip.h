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
// Both exponentials of a pair come from one: with t = exp(0.2*sigma/(r - a*sigma)),
// exp(sigma/(r-a sigma)) = t^5 and g = exp(1.2 sigma/(r - a sigma)) = t^6.
//
// Divergence control: phase 1 runs the cheap distance test over all list slots
// (list read eight slots at a time, so eight coalesced loads are in flight) and
// parks the in-range entries in a per-thread LDS queue; phase 2 runs the expensive
// part only over that queue, so a wave's trip count is its largest in-range count
// (4-12) rather than its largest list length (16-25).
//
// LDSPOS = true : one workgroup stages the whole box's positions in LDS
//                 (N*24 B: 96 KiB at N = 4096) and gathers r_j from there.
// LDSPOS = false: r_j gathered from global memory (L2-resident for the sizes
//                 that do not fit LDS, e.g. 786 KiB at N = 32768).
//   grid = (nsplit, nboxes_in_launch); each block takes atoms [split*chunk, ...)
// =====================================================================================
struct AtomSum { double e; unsigned long long np, nt; };

constexpr int kQCap = 12;   // in-range entries per molecule parked in LDS between the two phases

// `queue` points at this thread's column of an LDS array [kQCap][BLOCK] (entry q at queue[q*BLOCK]:
// consecutive threads, consecutive banks).  The list is read eight slots at a time and ONE CHUNK
// AHEAD: `cur` arrives holding this molecule's first eight entries; while a chunk is being tested the
// next one -- of this molecule, or the first of the thread's next molecule `inext` -- is already in
// flight, so the HBM latency of the list stream hides behind the LDS gathers and distance tests.
template <int BLOCK, typename PosFn, typename IvFn>
__device__ __forceinline__ AtomSum atom_energy(int i, int n, const uint32_t* __restrict__ L, int N, int S,
                                               uint32_t* __restrict__ queue, PosFn getpos, IvFn getiv,
                                               uint32_t (&cur)[8], int inext)
{
    double xi, yi, zi;
    getpos(i, xi, yi, zi);

    // phase 1: cheap distance test over all list slots; the in-range entries are parked in LDS.
    int cnt = 0;
    unsigned long long over = 0ull;             // in-range slots beyond the LDS queue (re-read later)
    for (int s0 = 0; s0 < n || s0 == 0; s0 += 8) {
        uint32_t nxt[8];
        const bool last = s0 + 8 >= n;
        const int pi = last ? inext : i;                      // whose chunk comes next
        const int ps = last ? 0 : s0 + 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt[u] = (pi >= 0 && ps + u < S) ? L[(size_t)(ps + u) * N + pi] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < n) {
                double xj, yj, zj, ix, iy, iz;
                getpos((int)(cur[u] & kJMask), xj, yj, zj);
                getiv((int)(cur[u] >> kJBits), ix, iy, iz);
                const double dx = (xj + ix) - xi, dy = (yj + iy) - yi, dz = (zj + iz) - zi;   // molint.F90:447,450
                const double r2 = dx * dx + dy * dy + dz * dz;
                if (r2 < kRcSq) {                                                             // :454
                    if (cnt < kQCap) queue[cnt * BLOCK] = cur[u];
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
        const double rinv = fast_rsqrt(r2);
        const double den = fma_sc(r2, rinv, -kSigA);   // r - a sigma: < 0 inside the cutoff
        // r2 < rc^2 but r rounded onto rc: the pair's energy is exactly 0 in the limit
        const double w = fast_rcp(__builtin_fmin(den, -1.0e-300));
        const double t = fast_exp_neg(0.2 * kSigma * w);
        const double t2 = t * t, t4 = t2 * t2;
        const double e1 = t4 * t;                   // exp(sigma/(r - a sigma))       :459
        const double g  = t4 * t2;                  // exp(gamma sigma/(r - a sigma)) :462
        const double q = kSigSq * rinv * rinv;
        e2 += (kAeps * (kBigB * (q * q) - 1.0)) * e1;                                 // :460-461
        const double ux = dx * rinv, uy = dy * rinv, uz = dz * rinv;
        const double gx = g * ux, gy = g * uy, gz = g * uz;
        S0 += g;  Q += g * g;
        S1x += gx; S1y += gy; S1z += gz;
        Sxx += gx * ux; Syy += gy * uy; Szz += gz * uz;
        Sxy += gx * uy; Sxz += gx * uz; Syz += gy * uz;
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
        gather(L[(size_t)s * N + i], v);
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
                    const int* __restrict__ nn, double* __restrict__ partial,
                    unsigned long long* __restrict__ cpartial,
                    int N, int S, int ivcap, int box0, int nsplit, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_e[BLOCK / 64];
    __shared__ unsigned long long red_p[BLOCK / 64], red_t[BLOCK / 64];

    const int b = box0 + blockIdx.y;
    const int split = blockIdx.x;
    const int tid = threadIdx.x;
    const double* P  = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    const uint32_t* L = list + (size_t)b * S * N;
    const int* NN = nn + (size_t)b * N;
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
    const int a0 = split * chunk;
    const int a1 = min(N, a0 + chunk);
    int i = a0 + tid;
    uint32_t cur[8];
    int n_cur = 0;
    if (i < a1) {
        n_cur = NN[i];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = u < S ? L[(size_t)u * N + i] : 0u;
    }
    for (; i < a1; i += BLOCK) {
        const int inext = i + BLOCK < a1 ? i + BLOCK : -1;
        const int n_next = inext >= 0 ? NN[inext] : 0;          // one molecule ahead, like the list chunks
        AtomSum a = atom_energy<BLOCK>(i, n_cur, L, N, S, queue, getpos, getiv, cur, inext);
        esum += a.e; np += a.np; nt += a.nt;
        n_cur = n_next;
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
