// mw_neighbours.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine:
// compute_neighbours (molint.F90:501-559): brute-force builder and the cell-grid builder, both
// bit-identical to the reference's list.
#pragma once

#include "mw_common.hip.h"

namespace mw {

// =====================================================================================
// Neighbour list, brute force over (j, image): the reference's own enumeration
// order (j ascending, image ascending) falls out of the loop nest, so the list is
// identical entry for entry.  r_j and the image vector are wave-uniform (scalar
// registers); only r_i and the running count live in vector registers.
// The distance arithmetic is kept unfused (no FMA contraction) so that the
// in/out decision at the list radius is bit-identical to the reference's
// molint.F90:529-537 evaluated on a CPU without FMA.
//   grid = (ceil(N/256), nboxes_in_launch), block = 256
// =====================================================================================
__global__ __launch_bounds__(256)
void k_build_neighbours(const double* __restrict__ pos, const double* __restrict__ ivect,
                        const int* __restrict__ nivect,
                        uint32_t* __restrict__ listm, int* __restrict__ nn, unsigned char* __restrict__ cin,
                        int* __restrict__ stats,
                        const int* __restrict__ use_grid, int N, int S, int ivcap, int box0)
{
#pragma clang fp contract(off)
    const int b = box0 + blockIdx.y;
    if (use_grid[b]) return;              // this box goes through k_cell_search
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double* P  = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    const int niv = nivect[b];
    uint32_t* LM = listm + ((size_t)b * N + (i < N ? i : 0)) * kRow;
    const bool active = i < N;
    const int ii = active ? i : 0;
    const double xi = P[3 * ii], yi = P[3 * ii + 1], zi = P[3 * ii + 2];   // molint.F90:522
    int cnt = 0, cin_ = 0, bnd_ = 0;       // cin_: entries already inside the energy cutoff, bnd_: any non-central image
                                           // (both sort keys of k_list_order)

    for (int j = 0; j < N; ++j) {                                           // :525
        const double vx = P[3 * j] - xi, vy = P[3 * j + 1] - yi, vz = P[3 * j + 2] - zi;   // :529
        for (int k = 0; k < niv; ++k) {                                     // :531
            const double tx = vx + IV[3 * k], ty = vy + IV[3 * k + 1], tz = vz + IV[3 * k + 2];   // :534
            const double r2 = tx * tx + ty * ty + tz * tz;                  // :535
            if (r2 < kRnSq && !(k == 0 && j == i)) {                        // :532,537
                if (active && cnt < S) LM[cnt] = pack_entry(j, k);
                ++cnt;
                cin_ += r2 < kRcSq ? 1 : 0;
                bnd_ |= k != 0 ? 1 : 0;
            }
        }
    }
    if (active) { nn[(size_t)b * N + i] = cnt < S ? cnt : S; cin[(size_t)b * N + i] = (unsigned char)((cin_ < 127 ? cin_ : 127) | (bnd_ << 7)); }

    // per-box statistics: min nn, max nn (max > S means overflow)
    int mn = wave_min_i(active ? cnt : 0x7fffffff);
    int mx = wave_max_i(active ? cnt : 0);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&stats[2 * b], mn);
        atomicMax(&stats[2 * b + 1], mx);
    }
}

// =====================================================================================
// Neighbour list through a cell grid: O(N) candidates instead of 27 N^2 tests, and still the
// reference's list entry for entry.
//   * A candidate (j, image) only ever comes from the 27 grid cells around molecule i; the grid
//     spacing is >= the list radius (with a 1e-9 margin for the rounding of the cell assignment),
//     so every pair the reference accepts is among the candidates.
//   * Each candidate is decided by the reference's own expression on the unwrapped positions,
//     |(r_j - r_i) + ivect_k|^2 < rn^2, unfused (molint.F90:529-537), with ivect_k taken from the
//     same table the reference builds -- an image outside that table is not a candidate, exactly
//     as the reference never tests it.
//   * The accepted entries are rank-sorted by (j, image) in LDS, which is the reference's
//     enumeration order, before they are written to the molecule-major list (k_list_order derives the
//     slot-major layout).
// Needs >= 3 grid cells along every cell vector; smaller boxes use k_build_neighbours.
// Four launches per batch: bin (count), scan, scatter, search.
// =====================================================================================
struct GridDesc {
    double hinv[9];        // s = hinv * r (row-major 3x3): fractional coordinates
    int nc[3];             // grid cells along h1, h2, h3 (0: box uses the brute-force kernel)
    int im[3];             // image-table half widths (molint.F90:189-191)
    int ncell;             // nc[0]*nc[1]*nc[2]
    int pad;
};

// {min, max} row length of each box of a build: reset on the device (no host round trip in front of a build)
__global__ void k_init_stats(int* __restrict__ stats, int box0, int count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < count) { stats[2 * (box0 + b)] = 0x7fffffff; stats[2 * (box0 + b) + 1] = 0; }
}

// shift (floor of the fractional coordinate) packed 10 bits per component, biased by 512
__device__ __forceinline__ int pack_shift(int a, int b, int c) { return (a + 512) | ((b + 512) << 10) | ((c + 512) << 20); }

__global__ __launch_bounds__(256)
void k_cell_bin(const double* __restrict__ pos, const GridDesc* __restrict__ grid,
                int* __restrict__ cellid, int* __restrict__ shift, int* __restrict__ count,
                int N, int cstride, int box0)
{
    const int b = box0 + blockIdx.y;
    const GridDesc& G = grid[b];
    if (G.nc[0] == 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double* p = pos + ((size_t)b * N + i) * 3;
    const double x = p[0], y = p[1], z = p[2];
    int c[3], f[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double sd = G.hinv[3 * d] * x + G.hinv[3 * d + 1] * y + G.hinv[3 * d + 2] * z;
        const double fl = floor(sd);
        int ci = (int)((sd - fl) * (double)G.nc[d]);
        ci = ci < 0 ? 0 : (ci >= G.nc[d] ? G.nc[d] - 1 : ci);
        c[d] = ci;
        int sh = (int)fl;
        f[d] = sh < -511 ? -511 : (sh > 511 ? 511 : sh);   // farther out than the image table reaches anyway
    }
    const int cid = (c[0] * G.nc[1] + c[1]) * G.nc[2] + c[2];
    cellid[(size_t)b * N + i] = cid;
    shift[(size_t)b * N + i] = pack_shift(f[0], f[1], f[2]);
    atomicAdd(&count[(size_t)b * cstride + cid], 1);
}

// exclusive scan of the per-cell counts -> start[0..ncell]; cursor = start.  One block per box.
__global__ __launch_bounds__(1024)
void k_cell_scan(const GridDesc* __restrict__ grid, const int* __restrict__ count,
                 int* __restrict__ start, int* __restrict__ cursor, int cstride, int box0)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int b = box0 + blockIdx.x;
    const int ncell = grid[b].nc[0] == 0 ? 0 : grid[b].ncell;
    const int* cnt = count + (size_t)b * cstride;
    int* st = start + (size_t)b * (cstride + 1);
    int* cu = cursor + (size_t)b * cstride;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ncell; base += 1024) {
        const int idx = base + tid;
        const int v = idx < ncell ? cnt[idx] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int excl = carry + woff + incl - v;
        if (idx < ncell) { st[idx] = excl; cu[idx] = excl; }
        __syncthreads();
        if (tid == 1023) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0 && ncell > 0) st[ncell] = carry;
}

__global__ __launch_bounds__(256)
void k_cell_scatter(const GridDesc* __restrict__ grid, const int* __restrict__ cellid,
                    int* __restrict__ cursor, int* __restrict__ sorted, int N, int cstride, int box0)
{
    const int b = box0 + blockIdx.y;
    if (grid[b].nc[0] == 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int cid = cellid[(size_t)b * N + i];
    const int slot = atomicAdd(&cursor[(size_t)b * cstride + cid], 1);
    sorted[(size_t)b * N + slot] = i;      // order inside a cell is arbitrary: the final lists are sorted
}

// One thread per molecule, taken in grid order so that a wavefront walks the same cells.
__global__ __launch_bounds__(256)
void k_cell_search(const double* __restrict__ pos, const double* __restrict__ ivect,
                   const GridDesc* __restrict__ grid, const int* __restrict__ cellid,
                   const int* __restrict__ shift, const int* __restrict__ start, const int* __restrict__ sorted,
                   uint32_t* __restrict__ listm,
                   int* __restrict__ nn, unsigned char* __restrict__ cin, int* __restrict__ stats,
                   int N, int S, int ivcap, int cstride, int box0)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    uint32_t* buf = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;     // column of [S][256] keys
    const int b = box0 + blockIdx.y;
    const GridDesc& G = grid[b];
    if (G.nc[0] == 0) return;                                             // wave-uniform
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool active = p < N;
    const double* P = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    const int* ST = start + (size_t)b * (cstride + 1);
    const int* SO = sorted + (size_t)b * N;
    const int* SH = shift + (size_t)b * N;
    const int i = active ? SO[p] : 0;
    const double xi = P[3 * i], yi = P[3 * i + 1], zi = P[3 * i + 2];      // molint.F90:522
    const int cid = cellid[(size_t)b * N + i];
    const int c2 = cid % G.nc[2], c1 = (cid / G.nc[2]) % G.nc[1], c0 = cid / (G.nc[2] * G.nc[1]);
    const int shi = SH[i];
    const int si0 = (shi & 1023) - 512, si1 = ((shi >> 10) & 1023) - 512, si2 = ((shi >> 20) & 1023) - 512;
    const int w1 = 2 * G.im[1] + 1, w2 = 2 * G.im[2] + 1;
    const int central = (G.im[0] * w1 + G.im[1]) * w2 + G.im[2];
    int cnt = 0, cin_ = 0, bnd_ = 0;

    if (active) {
        for (int d0 = -1; d0 <= 1; ++d0) {
            int n0 = c0 + d0, o0 = 0;
            if (n0 < 0) { n0 += G.nc[0]; o0 = -1; } else if (n0 >= G.nc[0]) { n0 -= G.nc[0]; o0 = 1; }
            for (int d1 = -1; d1 <= 1; ++d1) {
                int n1 = c1 + d1, o1 = 0;
                if (n1 < 0) { n1 += G.nc[1]; o1 = -1; } else if (n1 >= G.nc[1]) { n1 -= G.nc[1]; o1 = 1; }
                for (int d2 = -1; d2 <= 1; ++d2) {
                    int n2 = c2 + d2, o2 = 0;
                    if (n2 < 0) { n2 += G.nc[2]; o2 = -1; } else if (n2 >= G.nc[2]) { n2 -= G.nc[2]; o2 = 1; }
                    const int nc = (n0 * G.nc[1] + n1) * G.nc[2] + n2;
                    const int e0 = ST[nc], e1 = ST[nc + 1];
                    for (int q = e0; q < e1; ++q) {
                        const int j = SO[q];
                        const int shj = SH[j];
                        // the image of j that lies in this neighbouring grid cell: r_j + H m
                        const int m0 = o0 + si0 - ((shj & 1023) - 512);
                        const int m1 = o1 + si1 - (((shj >> 10) & 1023) - 512);
                        const int m2 = o2 + si2 - (((shj >> 20) & 1023) - 512);
                        if (m0 < -G.im[0] || m0 > G.im[0] || m1 < -G.im[1] || m1 > G.im[1] || m2 < -G.im[2] || m2 > G.im[2])
                            continue;                                   // not in the reference's image table
                        const int lin = ((m0 + G.im[0]) * w1 + (m1 + G.im[1])) * w2 + (m2 + G.im[2]);
                        const int k = lin == central ? 0 : (lin < central ? lin + 1 : lin);   // molint.F90:197-213
                        if (k == 0 && j == i) continue;                                       // :532
                        const double vx = P[3 * j] - xi, vy = P[3 * j + 1] - yi, vz = P[3 * j + 2] - zi;   // :529
                        const double tx = vx + IV[3 * k], ty = vy + IV[3 * k + 1], tz = vz + IV[3 * k + 2]; // :534
                        const double r2 = tx * tx + ty * ty + tz * tz;                                     // :535
                        if (r2 < kRnSq) {                                                                  // :537
                            if (cnt < S) buf[cnt * 256] = ((uint32_t)j << 10) | (uint32_t)k;   // sort key: j, then image
                            ++cnt;
                            cin_ += r2 < kRcSq ? 1 : 0;
                            bnd_ |= k != 0 ? 1 : 0;
                        }
                    }
                }
            }
        }
        // rank sort (keys are unique): entry a goes to slot #{keys smaller than key a}
        const int n = cnt < S ? cnt : S;
        uint32_t* LM = listm + ((size_t)b * N + i) * kRow;
        for (int a = 0; a < n; ++a) {
            const uint32_t ka = buf[a * 256];
            int r = 0;
            for (int c = 0; c < n; ++c) r += (buf[c * 256] < ka) ? 1 : 0;
            LM[r] = pack_entry((int)(ka >> 10), (int)(ka & 1023u));
        }
        nn[(size_t)b * N + i] = n;
        cin[(size_t)b * N + i] = (unsigned char)((cin_ < 127 ? cin_ : 127) | (bnd_ << 7));
    }
    int mn = wave_min_i(active ? cnt : 0x7fffffff);
    int mx = wave_max_i(active ? cnt : 0);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&stats[2 * b], mn);
        atomicMax(&stats[2 * b + 1], mx);
    }
}

// =====================================================================================
// Sorted slot-major copy of the list for the full-box kernel (one molecule per lane): the molecules of a box
// are ordered by (neighbours inside the energy cutoff at build time, interior / boundary, row length), so that
// the 64 lanes of a wavefront run the same number of cheap distance tests and the same number of expensive pair
// evaluations, and whole wavefronts of interior molecules never touch the image vectors.
// The sort runs inside SEGMENTS of `seg` consecutive molecules (seg % 64 == 0; the whole box when its positions
// are staged in LDS, 1024 molecules when they are gathered through the caches, where neighbours in index are
// neighbours in space and a wavefront's gathers should stay close together), one workgroup per segment.
// Stable counting sort -- a box always gets the same order, so energies stay bitwise reproducible:
//   A  histogram over (key, group of 64 consecutive molecules) in LDS,
//   scan in (key, group) order,
//   B  every molecule's destination = start of its (key, group) cell + its rank among the group's lanes
//      with the same key (ballots),
//   C  column t of the slot-major list <- row order[t] of the molecule-major list (coalesced stores), central-image
//      entries first, zero-padded to the longest row of the column's group of 64.
// kbits = number of key bits kept (the (key, group) table must fit kOrderSlots); kbits < 0: identity order.
//   grid = (segments, boxes), block = 1024
// =====================================================================================
constexpr int kOrderSlots = 32768;      // ints of dynamic LDS at most (128 KiB)

__global__ __launch_bounds__(1024)
void k_list_order(const uint32_t* __restrict__ listm, const int* __restrict__ nn, const unsigned char* __restrict__ cin,
                  const int* __restrict__ stats, uint32_t* __restrict__ list, int* __restrict__ order,
                  int* __restrict__ nns, int* __restrict__ cmax, int N, int S, int box0, int kbits, int seg)
{
    extern __shared__ __attribute__((aligned(16))) int hist[];       // [keys][groups of the segment], dynamic
    __shared__ int wsum[16];
    const int b = box0 + blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i0 = blockIdx.x * seg, i1 = min(N, i0 + seg);          // this workgroup's molecules = its list columns
    const int ngroups = (i1 - i0 + 63) >> 6;
    const int ngroups_box = (N + 63) >> 6;
    const uint32_t* LM = listm + (size_t)b * N * kRow;
    const int* NN = nn + (size_t)b * N;
    const unsigned char* CI = cin + (size_t)b * N;
    uint32_t* L = list + (size_t)b * S * N;
    int* ORD = order + (size_t)b * N;
    int* NNS = nns + (size_t)b * N;
    int* CM = cmax + (size_t)b * ngroups_box;
    const int nmin = stats[2 * b];

    // key = in-range neighbours at build time (4 bits) | has entries of a non-central image (1 bit) | row length (3 bits)
    auto keyof = [&](int i, int n) {
        const int c = min((int)(CI[i] & 0x7f), 15);
        const int bnd = CI[i] >> 7;
        const int nb = min(7, max(0, n - nmin) >> 1);
        return ((c << 4) | (bnd << 3) | nb) >> (8 - kbits);
    };

    if (kbits < 0) {
        for (int i = i0 + tid; i < i1; i += 1024) { ORD[i] = i; NNS[i] = min(NN[i], S); }
    } else {
        const int K = 1 << kbits, M = K * ngroups;
        for (int e = tid; e < M; e += 1024) hist[e] = 0;
        __syncthreads();
        for (int i = i0 + tid; i < i1; i += 1024) atomicAdd(&hist[keyof(i, NN[i]) * ngroups + ((i - i0) >> 6)], 1);
        __syncthreads();
        // exclusive scan of hist[0..M): `per` consecutive elements per thread
        const int per = (M + 1023) / 1024;
        const int e0 = min(M, tid * per), e1 = min(M, e0 + per);
        int local = 0;
        for (int e = e0; e < e1; ++e) local += hist[e];
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int run = incl - local;
        for (int w = 0; w < wid; ++w) run += wsum[w];
        for (int e = e0; e < e1; ++e) { const int v = hist[e]; hist[e] = run; run += v; }
        __syncthreads();
        for (int base = i0; base < i1; base += 1024) {
            const int i = base + tid;
            const bool valid = i < i1;
            const int n = valid ? NN[i] : 0;
            const int key = valid ? keyof(i, n) : 0;
            unsigned long long same = __ballot(valid);
            for (int bit = 0; bit < kbits; ++bit) {
                const unsigned long long m = __ballot((key >> bit) & 1);
                same &= ((key >> bit) & 1) ? m : ~m;
            }
            const int rank = __popcll(same & ((1ull << lane) - 1ull));
            if (valid) {
                const int dst = i0 + hist[key * ngroups + ((i - i0) >> 6)] + rank;
                ORD[dst] = i; NNS[dst] = min(n, S);
            }
        }
    }
    __syncthreads();   // ORD / NNS of this segment are read back below by other threads of this workgroup
    // Column t <- row ORD[t], entries of the central image first: the full-box kernel then skips the image-vector
    // gather for the slots every lane of a wavefront knows to be central (c0min).  NNS[t] = n | n0 << 8 (n0 = central
    // entries), CM[group] = longest row | smallest n0 << 8.
    for (int base = i0; base < i1; base += 1024) {
        const int t = base + tid;
        const bool valid = t < i1;
        const int i = valid ? ORD[t] : 0;
        const int n = valid ? NNS[t] : 0;
        const int nmax = __builtin_amdgcn_readfirstlane(wave_max_i(n));
        const uint4* row = reinterpret_cast<const uint4*>(LM + (size_t)i * kRow);
        int w = 0;                                            // next slot of the column
        for (int pass = 0; pass < 2; ++pass) {
            for (int s4 = 0; s4 < nmax; s4 += 4) {
                if (s4 < n) {
                    const uint4 v = row[s4 >> 2];
                    const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (s4 + u < n && ((e[u] >> kJBits) == 0u) == (pass == 0)) { L[(size_t)w * N + t] = e[u]; ++w; }
                }
            }
            if (pass == 1 && valid)
                for (; w < nmax; ++w) L[(size_t)w * N + t] = 0u;       // zero-padded to the group's longest row
            if (pass == 0) {
                const int n0 = w;
                if (valid) NNS[t] = n | (n0 << 8);
                const int c0min = __builtin_amdgcn_readfirstlane(wave_min_i(valid ? n0 : 0x7fff));
                if (lane == 0 && valid) CM[t >> 6] = nmax | ((c0min > 255 ? 255 : c0min) << 8);
            }
        }
    }
}

}  // namespace mw
