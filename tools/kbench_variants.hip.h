// kbench_variants.hip.h -- experimental variants of the hot kernels, timed side by side by tools/kbench.hip on one
// MI355X (methodology: one process, interleaved rounds).  Nothing here ships: a variant that wins is moved into
// mc_water_ls_mw_amd/csrc/ and the loser is deleted from this file.
#pragma once

namespace kb {
using namespace mw;

constexpr int V_DYN = 1;       // wavefronts draw groups of 64 list columns from an LDS ticket, heaviest first
constexpr int V_PIPE = 2;      // phase 1 branch-free, four slots in flight
constexpr int V_PERSIST = 4;   // one workgroup per CU walks several boxes; next box's positions prefetched into registers
constexpr int V_NOP2 = 8;      // ablation: phase 2 skipped (timing only)
constexpr int V_SERP = 32;     // static hand-out, alternate passes reversed
constexpr int V_NOP1 = 64;     // ablation: phase 1 takes the first min(n, 7) slots without testing (timing only)

template <int VAR, int BLOCK, typename PosFn, typename IvFn>
__device__ __forceinline__ AtomSum atom_energy_v(int t, int mol, int n, int nmax, const uint32_t* __restrict__ L, int N, int S,
                                                 uint32_t* __restrict__ queue, PosFn getpos, IvFn getiv,
                                                 uint32_t (&cur)[8], int tnext)
{
    double xi, yi, zi;
    getpos(mol, xi, yi, zi);
    int cnt = 0;
    unsigned long long over = 0ull;
    for (int s0 = 0; s0 < nmax || s0 == 0; s0 += 8) {
        uint32_t nxt[8];
        const bool last = s0 + 8 >= nmax;
        const int pt = last ? tnext : t;
        const int ps = last ? 0 : s0 + 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt[u] = (pt >= 0 && ps + u < S) ? L[(size_t)(ps + u) * N + pt] : 0u;
        if constexpr (VAR & V_NOP1) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < n && s0 + u < 7) { queue[cnt * BLOCK] = cur[u]; ++cnt; }
        } else if constexpr (VAR & V_PIPE) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (s0 + 4 * h < nmax) {
                    uint32_t e[4];
                    double v[4][6];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        e[u] = (s0 + 4 * h + u < n) ? cur[4 * h + u] : 0u;
                        getpos((int)(e[u] & kJMask), v[u][0], v[u][1], v[u][2]);
                        getiv((int)(e[u] >> kJBits), v[u][3], v[u][4], v[u][5]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double dx = (v[u][0] + v[u][3]) - xi, dy = (v[u][1] + v[u][4]) - yi, dz = (v[u][2] + v[u][5]) - zi;
                        const double r2 = dx * dx + dy * dy + dz * dz;
                        const bool in = (s0 + 4 * h + u < n) && (r2 < kRcSq);
                        const int idx = cnt < kQCap ? cnt : kQCap;          // row kQCap is a dump row
                        if (in) queue[idx * BLOCK] = e[u];
                        cnt += in ? 1 : 0;
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (s0 + u < nmax) {
                    const bool live = s0 + u < n;
                    const uint32_t e = live ? cur[u] : 0u;
                    double xj, yj, zj, ix, iy, iz;
                    getpos((int)(e & kJMask), xj, yj, zj);
                    getiv((int)(e >> kJBits), ix, iy, iz);
                    const double dx = (xj + ix) - xi, dy = (yj + iy) - yi, dz = (zj + iz) - zi;
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    if (live && r2 < kRcSq) {
                        if (cnt < kQCap) queue[cnt * BLOCK] = e;
                        else over |= 1ull << (s0 + u);
                        ++cnt;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
    }

    double e2 = 0.0, S0 = 0.0, Q = 0.0, S1x = 0.0, S1y = 0.0, S1z = 0.0;
    double Sxx = 0.0, Syy = 0.0, Szz = 0.0, Sxy = 0.0, Sxz = 0.0, Syz = 0.0;
    auto gather = [&](uint32_t e, double (&v)[6]) {
        getpos((int)(e & kJMask), v[0], v[1], v[2]);
        getiv((int)(e >> kJBits), v[3], v[4], v[5]);
    };
    auto accumulate = [&](const double (&v)[6]) {
        const double dx = (v[0] + v[3]) - xi, dy = (v[1] + v[4]) - yi, dz = (v[2] + v[5]) - zi;
        const double r2 = dx * dx + dy * dy + dz * dz;
        double rinv, e1, g;
        pair_terms(r2, rinv, e1, g);
        const double ri2 = rinv * rinv, ri4 = ri2 * ri2;
        e2 = __builtin_fma(fma_sc(ri4, kAepsBSig4, -kAeps), e1, e2);
        const double w1 = g * rinv, w2 = g * ri2;
        const double hx = w2 * dx, hy = w2 * dy, hz = w2 * dz;
        S0 += g;  Q = __builtin_fma(g, g, Q);
        S1x = __builtin_fma(w1, dx, S1x); S1y = __builtin_fma(w1, dy, S1y); S1z = __builtin_fma(w1, dz, S1z);
        Sxx = __builtin_fma(hx, dx, Sxx); Syy = __builtin_fma(hy, dy, Syy); Szz = __builtin_fma(hz, dz, Szz);
        Sxy = __builtin_fma(hx, dy, Sxy); Sxz = __builtin_fma(hx, dz, Sxz); Syz = __builtin_fma(hy, dz, Syz);
    };
    if constexpr (VAR & V_NOP2) {
        AtomSum out;
        out.e = (double)cnt; out.np = (unsigned long long)cnt; out.nt = (unsigned long long)(cnt * (cnt - 1) / 2);
        return out;
    }
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
    if constexpr (VAR & V_PIPE) {
        if (cnt > kQCap) {          // rare: rescan the row for the in-range entries the queue had no room for
            int seen = 0;
            for (int s = 0; s < n; ++s) {
                double v[6];
                gather(L[(size_t)s * N + t], v);
                const double dx = (v[0] + v[3]) - xi, dy = (v[1] + v[4]) - yi, dz = (v[2] + v[5]) - zi;
                if (dx * dx + dy * dy + dz * dz < kRcSq) { if (seen >= kQCap) accumulate(v); ++seen; }
            }
        }
    } else {
        while (over) {
            const int s = __ffsll((long long)over) - 1;
            over &= over - 1ull;
            double v[6];
            gather(L[(size_t)s * N + t], v);
            accumulate(v);
        }
    }
    const double F2 = Sxx * Sxx + Syy * Syy + Szz * Szz + 2.0 * (Sxy * Sxy + Sxz * Sxz + Syz * Syz);
    const double F1 = S1x * S1x + S1y * S1y + S1z * S1z;
    const double T = 0.5 * ((F2 - Q) - 2.0 * kCos0 * (F1 - Q) + kCos0 * kCos0 * (S0 * S0 - Q));
    AtomSum out;
    out.e  = 0.5 * e2 + kLamEps * T;
    out.np = (unsigned long long)cnt;
    out.nt = (unsigned long long)(cnt * (cnt - 1) / 2);
    return out;
}

// LDS-staged boxes only (N*24 B + queue fit LDS), one workgroup per box (or per CU with V_PERSIST).
constexpr int kPre = 12;     // doubles per thread of the next box's positions (V_PERSIST; N <= 4096)

template <int VAR>
__global__ __launch_bounds__(1024)
void k_me(const double* __restrict__ pos, const double* __restrict__ ivect,
          const int* __restrict__ nivect, const uint32_t* __restrict__ list,
          const int* __restrict__ order, const int* __restrict__ nns, const int* __restrict__ cmax,
          double* __restrict__ partial, unsigned long long* __restrict__ cpartial,
          int N, int S, int ivcap, int nboxes)
{
    constexpr int BLOCK = 1024;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_e[16];
    __shared__ unsigned long long red_p[16], red_t[16];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = (N + 63) >> 6;
    double* spos = smem;
    double* siv = smem + 3 * (size_t)N;
    uint32_t* queue = reinterpret_cast<uint32_t*>(siv + (size_t)ivcap * 3) + tid;
    auto getiv = [&](int k, double& x, double& y, double& z) { x = siv[3 * k]; y = siv[3 * k + 1]; z = siv[3 * k + 2]; };
    auto getpos = [&](int j, double& x, double& y, double& z) { const double* p = spos + 3 * (size_t)j; x = p[0]; y = p[1]; z = p[2]; };

    double pre[kPre];
    const int bstep = (VAR & V_PERSIST) ? (int)gridDim.x : nboxes;
    int b = blockIdx.x;
    if constexpr (VAR & V_PERSIST) {
        const double* P = pos + (size_t)b * N * 3;
#pragma unroll
        for (int k = 0; k < kPre; ++k) { const int o = tid + k * BLOCK; pre[k] = o < 3 * N ? P[o] : 0.0; }
    }
    for (; b < nboxes; b += bstep) {
        const double* P  = pos + (size_t)b * N * 3;
        const double* IV = ivect + (size_t)b * ivcap * 3;
        const uint32_t* L = list + (size_t)b * S * N;
        const int* ORD = order + (size_t)b * N;
        const int* NNS = nns + (size_t)b * N;
        const int* CM = cmax + (size_t)b * G;
        const int niv = nivect[b];
        for (int t = tid; t < niv * 3; t += BLOCK) siv[t] = IV[t];
        if constexpr (VAR & V_PERSIST) {
#pragma unroll
            for (int k = 0; k < kPre; ++k) { const int o = tid + k * BLOCK; if (o < 3 * N) spos[o] = pre[k]; }
            const int bn = b + bstep;
            if (bn < nboxes) {
                const double* Pn = pos + (size_t)bn * N * 3;
#pragma unroll
                for (int k = 0; k < kPre; ++k) { const int o = tid + k * BLOCK; pre[k] = o < 3 * N ? Pn[o] : 0.0; }
            }
        } else {
            for (int t = tid; t < 3 * N; t += BLOCK) spos[t] = P[t];
        }
        if (tid == 0) s_ticket = 16;
        __syncthreads();

        double esum = 0.0;
        unsigned long long np = 0, nt = 0;
        uint32_t cur[8];
        // group hand-out
        int it = 0;
        auto group_of = [&](int ticket) -> int {                   // -1: none
            if constexpr (VAR & V_DYN) return ticket < G ? G - 1 - ticket : -1;
            else if constexpr (VAR & V_SERP) {
                const int pass = ticket >> 4, w = ticket & 15;
                const int g = pass * 16 + ((pass & 1) ? 15 - w : w);
                return g < G ? g : -1;
            } else return ticket < G ? ticket : -1;
        };
        int ticket = wid;
        int grp = group_of(ticket);
        int n_cur = 0, mol = 0;
        if (grp >= 0) {
            const int t = grp * 64 + lane;
            if (t < N) {
                n_cur = NNS[t]; mol = ORD[t];
#pragma unroll
                for (int u = 0; u < 8; ++u) cur[u] = u < S ? L[(size_t)u * N + t] : 0u;
            }
        }
        while (grp >= 0) {
            int tk;
            if constexpr (VAR & V_DYN) {
                tk = 0;
                if (lane == 0) tk = atomicAdd(&s_ticket, 1);
                tk = __builtin_amdgcn_readfirstlane(tk);
            } else tk = ticket + 16;
            const int gnext = group_of(tk);
            const int t = grp * 64 + lane;
            const bool act = t < N;
            int tnext = gnext >= 0 ? gnext * 64 + lane : -1;
            if (tnext >= N) tnext = -1;
            int n_next = 0, mol_next = 0;
            if (tnext >= 0) { n_next = NNS[tnext]; mol_next = ORD[tnext]; }
            const int nmax = CM[grp];
            AtomSum a = atom_energy_v<VAR, BLOCK>(act ? t : -1, mol, act ? n_cur : 0, nmax, L, N, S, queue, getpos, getiv, cur, tnext);
            if (act) { esum += a.e; np += a.np; nt += a.nt; }
            n_cur = n_next; mol = mol_next; grp = gnext; ticket = tk; ++it;
        }
        esum = wave_sum(esum); np = wave_sum_u64(np); nt = wave_sum_u64(nt);
        if (lane == 0) { red_e[wid] = esum; red_p[wid] = np; red_t[wid] = nt; }
        __syncthreads();                    // also: every wavefront is done with this box's LDS positions
        if (tid == 0) {
            double e = 0.0; unsigned long long p = 0, q = 0;
            for (int w = 0; w < 16; ++w) { e += red_e[w]; p += red_p[w]; q += red_t[w]; }
            partial[b] = e; cpartial[2 * b] = p; cpartial[2 * b + 1] = q;
        }
        if constexpr (VAR & V_PERSIST) __syncthreads();           // red_* are rewritten by the next box
    }
}

}  // namespace kb
