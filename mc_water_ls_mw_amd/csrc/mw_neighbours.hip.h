// mw_neighbours.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine:
// compute_neighbours (molint.F90:501-559): brute-force builder and the cell-grid builder, both
// bit-identical to the reference's list.
#pragma once
#include <type_traits>

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
//   grid = (ceil(N/block), nboxes_in_launch), block = 64 .. 256 (a multiple of 64, no larger than N needs)
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
    // (a wavefront without a molecule has nothing to do: with the reference's own 48-molecule cells three of a 256-thread
    // block's four wavefronts used to walk the whole 27 N^2 loop for nobody -- the launch uses blocks of one wavefront there)
    const bool idle = (int)(blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) >= N;      // (leaves below, after the block's barrier)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double* P  = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    const int niv = nivect[b];
    uint32_t* LM = listm + ((size_t)b * N + (i < N ? i : 0)) * kRow;
    const bool active = i < N;
    const int ii = active ? i : 0;
    const double xi = P[3 * ii], yi = P[3 * ii + 1], zi = P[3 * ii + 2];   // molint.F90:522
    int cnt = 0, cin_ = 0, bnd_ = 0;       // cin_: entries already inside the energy cutoff, bnd_: any non-central image
                                           // (both sort keys of k_list_order)

    // The image vectors are wave-uniform: nine at a time arrive in ONE batch of scalar loads (27 is what every cell wider than
    // the cutoff has).  Loaded one by one inside the test loop -- a scalar load and a wait per image, which no unrolling moved
    // across the loop's branches -- the ~200 cycles per test were the load's latency: 116 us for ONE 48-molecule box.
    constexpr int kIvBatch = 9;
    // Small boxes (the replica farm's 48-molecule cells: one wavefront per box, 16 384 boxes per build): the box's positions and
    // image vectors go through LDS once, and the loops read them back as broadcasts -- the scalar loads they were (three batches
    // per j, each a wait of its own) left the vector unit half idle (0.68 ms per 16 384 boxes).
    constexpr int kLdsN = 256, kLdsIv = 64;
    __shared__ double s_p[3 * kLdsN], s_iv[3 * kLdsIv];
    const bool staged = N <= kLdsN && niv <= kLdsIv;                        // (uniform)
    if (staged) {
        for (int t = threadIdx.x; t < 3 * N; t += blockDim.x) s_p[t] = P[t];
        for (int t = threadIdx.x; t < 3 * niv; t += blockDim.x) s_iv[t] = IV[t];
        __syncthreads();
    }
    if (idle) return;
    // (two copies of the loops, one per address space: a pointer chosen at run time between LDS and global memory would make every
    //  access through it a FLAT one)
    auto tests = [&](const double* __restrict__ Pj, const double* __restrict__ IVk) {
    for (int j = 0; j < N; ++j) {                                           // :525
        const double vx = Pj[3 * j] - xi, vy = Pj[3 * j + 1] - yi, vz = Pj[3 * j + 2] - zi;   // :529
        for (int k0 = 0; k0 < niv; k0 += kIvBatch) {                        // :531
            double iv[kIvBatch][3];
#pragma unroll
            for (int u = 0; u < kIvBatch; ++u) {
                const int kk = k0 + u < niv ? k0 + u : niv - 1;             // (uniform; a slot past the table is not tested)
                iv[u][0] = IVk[3 * kk]; iv[u][1] = IVk[3 * kk + 1]; iv[u][2] = IVk[3 * kk + 2];
            }
#pragma unroll
            for (int u = 0; u < kIvBatch; ++u) {
                const int k = k0 + u;
                if (k < niv) {                                              // (uniform)
                    const double tx = vx + iv[u][0], ty = vy + iv[u][1], tz = vz + iv[u][2];   // :534
                    const double r2 = tx * tx + ty * ty + tz * tz;          // :535
                    if (r2 < kRnSq && !(k == 0 && j == i)) {                // :532,537
                        if (active && cnt < S) LM[cnt] = pack_entry(j, k);
                        ++cnt;
                        cin_ += r2 < kRcSq ? 1 : 0;
                        bnd_ |= k != 0 ? 1 : 0;
                    }
                }
            }
        }
    }
    };
    if (staged) tests(s_p, s_iv); else tests(P, IV);
    if (active) { nn[(size_t)b * N + i] = cnt < S ? cnt : S; cin[(size_t)b * N + i] = (unsigned char)((cin_ < 127 ? cin_ : 127) | (bnd_ << 7)); }

    // per-box statistics: min nn, max nn (max > S means overflow)
    int mn = wave_min_i(active ? cnt : 0x7fffffff);
    int mx = wave_max_i(active ? cnt : 0);
    if ((threadIdx.x & 63) == 0) {          // (look first: see k_cell_pairs)
        if (mn < __hip_atomic_load(&stats[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&stats[2 * b], mn);
        if (mx > __hip_atomic_load(&stats[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[2 * b + 1], mx);
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
    double h[9];           // cell vectors (column-major hmatrix: h[3k + d] = component d of vector k)
    int nc[3];             // grid cells along h1, h2, h3 (0: box uses the brute-force kernel)
    int im[3];             // image-table half widths (molint.F90:189-191)
    int ncell;             // nc[0]*nc[1]*nc[2]
    float eps;             // bound on the error of the float pre-filter's squared distance (k_cell_pairs)
};

// One box's cell as the host hands it over (mw_set_cell): header, grid descriptor, then nivect image vectors (3 doubles each).
struct CellRecord { int niv, usegrid; double vol; double h[9]; GridDesc grid; };

// Files a CellRecord (in pinned host memory the device reads in place) into the engine's arrays.  grid = 1, block = 256
__global__ __launch_bounds__(256)
void k_set_cell(const CellRecord* __restrict__ rec, const double* __restrict__ iv_in, double* __restrict__ ivect, int* __restrict__ nivect,
                double* __restrict__ hmat, double* __restrict__ volume, GridDesc* __restrict__ grid, int* __restrict__ usegrid)
{
    const int t = threadIdx.x, niv = rec->niv;
    for (int k = t; k < 3 * niv; k += 256) ivect[k] = iv_in[k];
    if (t < 9) hmat[t] = rec->h[t];
    if (t == 9) { *nivect = niv; *usegrid = rec->usegrid; *volume = rec->vol; }
    constexpr int kWords = (int)(sizeof(GridDesc) / sizeof(int));
    static_assert(sizeof(GridDesc) % sizeof(int) == 0, "GridDesc is copied word by word");
    for (int k = t; k < kWords; k += 256) reinterpret_cast<int*>(grid)[k] = reinterpret_cast<const int*>(&rec->grid)[k];
}


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
                int* __restrict__ cellid, int* __restrict__ shift, float4* __restrict__ wrel, int* __restrict__ count,
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
    double u[3];           // position inside the grid cell, in units of the cell vectors / nc
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double sd = G.hinv[3 * d] * x + G.hinv[3 * d + 1] * y + G.hinv[3 * d + 2] * z;
        const double fl = floor(sd);
        int ci = (int)((sd - fl) * (double)G.nc[d]);
        ci = ci < 0 ? 0 : (ci >= G.nc[d] ? G.nc[d] - 1 : ci);
        c[d] = ci;
        u[d] = ((sd - fl) * (double)G.nc[d] - (double)ci) / (double)G.nc[d];
        int sh = (int)fl;
        f[d] = sh < -511 ? -511 : (sh > 511 ? 511 : sh);   // farther out than the image table reaches anyway
    }
    const int cid = (c[0] * G.nc[1] + c[1]) * G.nc[2] + c[2];
    cellid[(size_t)b * N + i] = cid;
    shift[(size_t)b * N + i] = pack_shift(f[0], f[1], f[2]);
    // the wrapped position relative to its grid cell's origin: small numbers, so single precision keeps ~1e-6 bohr
    // (k_cell_pairs' pre-filter); .w carries the molecule index
    wrel[(size_t)b * N + i] = make_float4((float)(u[0] * G.h[0] + u[1] * G.h[3] + u[2] * G.h[6]),
                                          (float)(u[0] * G.h[1] + u[1] * G.h[4] + u[2] * G.h[7]),
                                          (float)(u[0] * G.h[2] + u[1] * G.h[5] + u[2] * G.h[8]), __int_as_float(i));
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
void k_cell_scatter(const GridDesc* __restrict__ grid, const int* __restrict__ cellid, const int* __restrict__ shift,
                    const float4* __restrict__ wrel, int* __restrict__ cursor, int* __restrict__ sorted,
                    float4* __restrict__ wpos, int* __restrict__ wsh, int N, int cstride, int box0)
{
    const int b = box0 + blockIdx.y;
    if (grid[b].nc[0] == 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int cid = cellid[(size_t)b * N + i];
    const int slot = atomicAdd(&cursor[(size_t)b * cstride + cid], 1);
    sorted[(size_t)b * N + slot] = i;      // order inside a cell is arbitrary: the final lists are sorted
    wpos[(size_t)b * N + slot] = wrel[(size_t)b * N + i];    // cell-ordered copies: a cell's molecules are one contiguous load
    wsh[(size_t)b * N + slot] = shift[(size_t)b * N + i];
}

// Bin + scan + scatter of one box in ONE workgroup, for boxes whose cell-ordered records fit LDS (N <= kSortBoxMax):
// the three kernels above move every molecule's record through HBM twice (cell id, shift and cell-relative position
// out of k_cell_bin, back into k_cell_scatter, out again in cell order) and pay two grid-wide launches for ~100 KB of
// data per box.  Here a box's positions are read once (and once more from L2), the per-cell counters and the scan
// live in LDS, the cell-ordered records are assembled in LDS and leave in one coalesced stream:
//   HBM traffic per box = 24 N in + 20 N + 4 (ncell + 1) out   (k_cell_pairs reads exactly these 20 N + 4 ncell bytes).
// Order inside a cell = arrival order of the LDS atomics (arbitrary, as in k_cell_scatter: the rows are sorted later).
//   grid = boxes, block = 1024; dynamic LDS = N * 24 + (cstride + 1) * 4 bytes
constexpr int kSortBoxMax = 5120;

struct CellRec { int c[3]; int sh; float4 w; };

__device__ __forceinline__ CellRec cell_record(const GridDesc& G, const double* __restrict__ p, int i)
{
    const double x = p[0], y = p[1], z = p[2];
    CellRec r;
    int f[3];
    double u[3];           // position inside the grid cell, in units of the cell vectors / nc
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double sd = G.hinv[3 * d] * x + G.hinv[3 * d + 1] * y + G.hinv[3 * d + 2] * z;
        const double fl = floor(sd);
        int ci = (int)((sd - fl) * (double)G.nc[d]);
        ci = ci < 0 ? 0 : (ci >= G.nc[d] ? G.nc[d] - 1 : ci);
        r.c[d] = ci;
        u[d] = ((sd - fl) * (double)G.nc[d] - (double)ci) / (double)G.nc[d];
        int sh = (int)fl;
        f[d] = sh < -511 ? -511 : (sh > 511 ? 511 : sh);   // farther out than the image table reaches anyway
    }
    r.sh = pack_shift(f[0], f[1], f[2]);
    // the wrapped position relative to its grid cell's origin: small numbers, so single precision keeps ~1e-6 bohr
    // (k_cell_pairs' pre-filter); .w carries the molecule index
    r.w = make_float4((float)(u[0] * G.h[0] + u[1] * G.h[3] + u[2] * G.h[6]),
                      (float)(u[0] * G.h[1] + u[1] * G.h[4] + u[2] * G.h[7]),
                      (float)(u[0] * G.h[2] + u[1] * G.h[5] + u[2] * G.h[8]), __int_as_float(i));
    return r;
}

__global__ __launch_bounds__(1024)
void k_cell_sort_box(const double* __restrict__ pos, const GridDesc* __restrict__ grid,
                     int* __restrict__ start, float4* __restrict__ wpos, int* __restrict__ wsh, int* __restrict__ stats,
                     int N, int cstride, int box0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sortbox_lds[];
    __shared__ int wsum[16];
    __shared__ int carry;
    const int b = box0 + blockIdx.x;
    const GridDesc& G = grid[b];
    if (threadIdx.x == 0) { stats[2 * b] = 0x7fffffff; stats[2 * b + 1] = 0; }   // {min, max} row length of this build (k_init_stats' job)
    if (G.nc[0] == 0) return;                                             // this box keeps the brute-force kernel
    float4* s_w = reinterpret_cast<float4*>(sortbox_lds);                 // [N] records in cell order
    int* s_sh = reinterpret_cast<int*>(s_w + N);                          // [N]
    unsigned int* s_rc = reinterpret_cast<unsigned int*>(s_sh + N);       // [N] rank inside the cell << 16 | ... (cell id kept apart)
    int* s_cnt = reinterpret_cast<int*>(s_rc + N);                        // [cstride + 1] counts, then starts
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ncell = G.ncell;
    const double* P = pos + (size_t)b * N * 3;
    for (int c = tid; c <= ncell; c += 1024) s_cnt[c] = 0;
    if (tid == 0) carry = 0;
    __syncthreads();
    // ---- pass 1: cell of every molecule, its rank among the cell's molecules ----------------------------------------
    for (int i = tid; i < N; i += 1024) {
        const CellRec r = cell_record(G, P + 3 * (size_t)i, i);
        const int cid = (r.c[0] * G.nc[1] + r.c[1]) * G.nc[2] + r.c[2];
        const int rank = atomicAdd(&s_cnt[cid], 1);
        s_rc[i] = ((unsigned int)rank << 16) | (unsigned int)cid;         // (ncell <= cstride < 65536, a cell never holds 65536 molecules)
    }
    __syncthreads();
    // ---- exclusive scan of the counts, in place --------------------------------------------------------------------
    for (int base = 0; base < ncell; base += 1024) {
        const int idx = base + tid;
        const int v = idx < ncell ? s_cnt[idx] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int excl = carry + woff + incl - v;
        if (idx < ncell) s_cnt[idx] = excl;
        __syncthreads();
        if (tid == 1023) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0) s_cnt[ncell] = carry;
    __syncthreads();
    // ---- pass 2: records into their cell-ordered slots (LDS), then out in one coalesced stream -----------------------
    for (int i = tid; i < N; i += 1024) {
        const CellRec r = cell_record(G, P + 3 * (size_t)i, i);
        const unsigned int rc = s_rc[i];
        const int slot = s_cnt[rc & 0xffffu] + (int)(rc >> 16);
        s_w[slot] = r.w;
        s_sh[slot] = r.sh;
    }
    __syncthreads();
    float4* WP = wpos + (size_t)b * N;
    int* WS = wsh + (size_t)b * N;
    int* ST = start + (size_t)b * (cstride + 1);
    for (int t = tid; t < N; t += 1024) { WP[t] = s_w[t]; WS[t] = s_sh[t]; }
    for (int c = tid; c <= ncell; c += 1024) ST[c] = s_cnt[c];
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
    if ((threadIdx.x & 63) == 0) {          // (look first: see k_cell_pairs)
        if (mn < __hip_atomic_load(&stats[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&stats[2 * b], mn);
        if (mx > __hip_atomic_load(&stats[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[2 * b + 1], mx);
    }
}

// =====================================================================================
// The list builder proper: one WAVEFRONT per block of `bcells` consecutive grid cells along the third grid axis
// (~4 molecules per grid cell on ice, so a single cell is too little work to cover a wavefront's memory latencies).
//   * The candidates of the block -- the molecules of the 9 x (bcells + 2) grid cells around it, ~240 on ice for
//     bcells = 4 -- are loaded once (contiguous 16-byte records, cell-ordered by k_cell_scatter) and held in
//     registers, 64 per chunk: wrapped positions relative to the block's origin in single precision, and the list
//     entry (j, image) each would become.  The image number follows from the wrap offsets and the two molecules'
//     cell shifts (m = wrap + floor(s_i) - floor(s_j): positions are never wrapped, G8); a candidate whose image is
//     not in the reference's image table is dropped here, exactly as the reference never tests it.  Every molecule
//     of a block normally has the same shift; if not, the candidates are re-staged when the shift changes.
//   * For each molecule i of the block (wave-uniform) the 64 lanes test 64 candidates per step in single precision
//     against rn^2 + eps, ALL chunks of the batch in straight-line code (4 or 5 chunks; the compiler interleaves
//     their dependency chains -- skipping the chunks outside the molecule's three slabs, as an earlier version did,
//     put a branch in front of every chunk and cost more than it saved); the ~20 hits go straight into the molecule's
//     row in LDS as finished entries (ballot + mbcnt).
//   * A hit whose single-precision distance lies within eps of rn^2 -- about one molecule in 2000 -- flags its row:
//     such a row is re-decided entry by entry with the reference's own double-precision expression on the
//     unwrapped positions, unfused (molint.F90:529-537).  eps bounds the pre-filter's rounding error (make_grid),
//     so the list is the reference's entry for entry.
//   * Rows are rank-sorted by (j, image), the reference's enumeration order, two per pass, and written to the
//     molecule-major list; k_list_order derives the slot-major layout.
// Needs >= 3 grid cells along every cell vector and bcells <= nc[2] (each (cell, wrap) pair, hence each image of a
// molecule, is a candidate at most once).
//   grid = ceil(max blocks of cells / 4) * (boxes rounded up to 8), block = 256 (wavefront w of a box's workgroup x takes
//   cell block 4x + w);
//   64 VGPRs and 25.5 KB of LDS per workgroup: six workgroups = 24 wavefronts per CU
// =====================================================================================
constexpr int kPairChunks = 5;      // candidate chunks (of 64) held in registers per batch (more candidates: more batches)
constexpr int kPairIB = 20;         // molecules of the cell block per block of rows at most (more molecules: another block of rows)
constexpr int kPairRowWords = 1040;  // LDS words for a block of rows: rows of `rowcap` = maxneigh + 2 (rounded up to 4) entries, i.e.
                                     // 20 rows at maxneigh = 50, 15 at 64 (a wavefront's ~18 molecules then need one block, not two)
constexpr int kPairMaxB = 5;        // grid cells per wavefront at most
constexpr int kPairPieces = 9 * (kPairMaxB + 2);
constexpr int kPairLookup = 320;    // enumeration positions with a one-read piece lookup (beyond: binary search)
constexpr uint32_t kNoEntry = 0xffffffffu;

// (sized so that six workgroups of four wavefronts fit a CU's 160 KiB: the kernel is latency bound -- 13 % slower with
// four workgroups per CU than with five, 10 % faster with six)
struct PairsLds {
    __attribute__((aligned(16))) uint32_t rows[kPairRowWords];          // [rows][rowcap] hits as sort keys: j << 10 | image
    float4 own[kPairIB];                         // the block's molecules (position relative to the block origin, index)
    int ownsh[kPairIB];                          // their packed shifts
    int cnt[kPairIB];                            // hits per molecule (bit 30: an ambiguous hit, re-decide in double precision)
    int ncin[kPairIB];                           // hits already inside the energy cutoff (single precision: a sort key only)
    int pstart[kPairPieces + 1], pq[kPairPieces];   // pieces of the enumeration (one grid cell each): first position, first sorted slot
    int pk[kPairPieces];                         // image number of the piece's molecules for equal shifts (m = wrap offsets) | wrap offsets (2 bits each, biased by 1) << 10
    float poff[kPairPieces][3];                  // origin of the piece's grid cell relative to the block's
    unsigned char tpiece[kPairLookup];           // piece of enumeration position t (t < kPairLookup)
};

static_assert(4 * sizeof(PairsLds) <= 26 * 1024, "six workgroups of k_cell_pairs per CU");

__device__ __forceinline__ int piece_of(const PairsLds& W, int t, int npieces)
{
    if (t < kPairLookup) return W.tpiece[t];
    int lo = 0, hi = npieces;                    // the largest r with pstart[r] <= t (empty pieces share a start: the last one holds t)
#pragma unroll
    for (int it = 0; it < 6; ++it) { const int mid = (lo + hi) >> 1; if (W.pstart[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256)
void k_cell_pairs(const double* __restrict__ pos, const double* __restrict__ ivect,
                  const GridDesc* __restrict__ grid, const int* __restrict__ start,
                  const float4* __restrict__ wpos, const int* __restrict__ wsh,
                  uint32_t* __restrict__ listm, int* __restrict__ nn, unsigned char* __restrict__ cin,
                  int* __restrict__ stats, int N, int S, int ivcap, int cstride, int box0, int bcells_max,
                  int nwg_per_box, int count)
{
    __shared__ PairsLds lds[4];
    const int rowcap = (S + 2 + 3) & ~3;                                  // hits kept per molecule: maxneigh, itself, one to tell an overflow
    const int ibmax = min(kPairIB, kPairRowWords / rowcap);               // rows per block of rows
    // XCD-aware placement: the hardware deals consecutive workgroups to the 8 XCDs in turn, and each XCD has its own L2.
    // A box's cell-ordered records (80 KB) are read ~27 times over by the wavefronts around each cell, so all workgroups of
    // a box go to ONE XCD (box mod 8): workgroup L -> XCD L mod 8, the (L / 8)-th workgroup that XCD receives.
    const int L = (int)blockIdx.x;                                         // 1-D grid of nwg_per_box * (count rounded up to 8) workgroups
    const int xcd = L & 7, kq = L >> 3;
    const int bl = (kq / nwg_per_box) * 8 + xcd;                          // box of this launch
    const int wg = kq % nwg_per_box;                                      // its workgroup
    if (bl >= count) return;
    const int b = box0 + bl;
    const GridDesc& G = grid[b];
    if (G.nc[0] == 0) return;                                             // this box keeps the brute-force kernel
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nc0 = G.nc[0], nc1 = G.nc[1], nc2 = G.nc[2];
    const int B = bcells_max < nc2 ? bcells_max : nc2;                    // cells per wavefront
    const int nb2 = (nc2 + B - 1) / B;                                    // blocks along the third axis
    const int blk = wg * 4 + wave;
    if (blk >= nc0 * nc1 * nb2) return;
    PairsLds& W = lds[wave];
    const int* ST = start + (size_t)b * (cstride + 1);
    const float4* WP = wpos + (size_t)b * N;
    const int* WS = wsh + (size_t)b * N;
    const double* P = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    uint32_t* LM = listm + (size_t)b * N * kRow;
    const int c2a = (blk % nb2) * B, c1 = (blk / nb2) % nc1, c0 = blk / (nb2 * nc1);
    const int Bw = min(B, nc2 - c2a);                                     // this block's cells: (c0, c1, c2a .. c2a + Bw - 1)
    const int crow = (c0 * nc1 + c1) * nc2;
    const int qc0 = ST[crow + c2a], nmol = ST[crow + c2a + Bw] - qc0;     // its molecules: one contiguous range of sorted slots
    if (nmol == 0) return;

    // ---- the pieces of the candidate enumeration, slab by slab along the third axis: piece r = e * 9 + (d0 + 1) * 3 +
    // (d1 + 1) is the grid cell (c0 + d0, c1 + d1, c2a - 1 + e), wrapped; lane r describes piece r ---------------------
    const int span = Bw + 2, npieces = 9 * span;
    int pcnt = 0;
    if (lane < npieces) {
        const int e = lane / 9, dd = lane % 9;
        const int d0 = dd / 3 - 1, d1 = dd % 3 - 1, d2 = e - 1;            // d2: offset from the block's first cell
        int n0 = c0 + d0, n1 = c1 + d1, n2 = c2a + d2, o0 = 0, o1 = 0, o2 = 0;
        if (n0 < 0) { n0 += nc0; o0 = -1; } else if (n0 >= nc0) { n0 -= nc0; o0 = 1; }
        if (n1 < 0) { n1 += nc1; o1 = -1; } else if (n1 >= nc1) { n1 -= nc1; o1 = 1; }
        if (n2 < 0) { n2 += nc2; o2 = -1; } else if (n2 >= nc2) { n2 -= nc2; o2 = 1; }
        const int ncl = (n0 * nc1 + n1) * nc2 + n2;
        const int q0 = ST[ncl];
        pcnt = ST[ncl + 1] - q0;
        W.pq[lane] = q0;
        const int po_ = (o0 + 1) | ((o1 + 1) << 2) | ((o2 + 1) << 4);
        {
            const int im0_ = G.im[0], im1_ = G.im[1], im2_ = G.im[2];            // (>= 1: the wrap offsets are always in the table)
            const int w1_ = 2 * im1_ + 1, w2_ = 2 * im2_ + 1;
            const int central_ = (im0_ * w1_ + im1_) * w2_ + im2_;
            const int lin = ((o0 + im0_) * w1_ + (o1 + im1_)) * w2_ + (o2 + im2_);
            W.pk[lane] = (lin == central_ ? 0 : (lin < central_ ? lin + 1 : lin)) | (po_ << 10);                     // molint.F90:197-213; wrap offsets above the image number
        }
        const double f0 = (double)d0 / (double)nc0, f1 = (double)d1 / (double)nc1, f2 = (double)d2 / (double)nc2;
        W.poff[lane][0] = (float)(f0 * G.h[0] + f1 * G.h[3] + f2 * G.h[6]);
        W.poff[lane][1] = (float)(f0 * G.h[1] + f1 * G.h[4] + f2 * G.h[7]);
        W.poff[lane][2] = (float)(f0 * G.h[2] + f1 * G.h[5] + f2 * G.h[8]);
    }
    int incl = pcnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
    if (lane < npieces) {
        W.pstart[lane] = incl - pcnt;
        for (int t = incl - pcnt; t < incl && t < kPairLookup; ++t) W.tpiece[t] = (unsigned char)lane;
    }
    if (lane == npieces - 1) W.pstart[npieces] = incl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int T = W.pstart[npieces];
    const float rn2_hi = (float)kRnSq + G.eps;
    const float eps_ = G.eps + 8.0f * (float)kRnSq * 1.1920929e-7f;      // band of the ambiguity test: (rn^2 - eps, rn^2 + eps) and a few ulps, so that
                                                                          // every single-precision hit above rn^2 - eps is inside it whatever the rounding
    const int im0 = G.im[0], im1 = G.im[1], im2 = G.im[2];
    const int w1 = 2 * im1 + 1, w2 = 2 * im2 + 1;
    const int central = (im0 * w1 + im1) * w2 + im2;
    int wmin = 0x7fffffff, wmax = 0;                                      // row lengths seen by this wavefront

    for (int ib0 = 0; ib0 < nmol; ib0 += ibmax) {
        const int nib = min(ibmax, nmol - ib0);
        // ---- the block's molecules: cell (hence slab range), position relative to the block origin, shift ----------
        if (lane < nib) {
            const int q = qc0 + ib0 + lane;
            int cell = 0;                                                  // which of the block's cells holds it
            for (int k = 1; k < Bw; ++k) cell += (ST[crow + c2a + k] <= q) ? 1 : 0;
            const int r = (cell + 1) * 9 + 4;                              // its piece: slab cell + 1, (d0, d1) = (0, 0)
            float4 v = WP[q];
            v.x += W.poff[r][0]; v.y += W.poff[r][1]; v.z += W.poff[r][2];
            W.own[lane] = v;
            W.ownsh[lane] = WS[q];
            W.cnt[lane] = 0; W.ncin[lane] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (T > 65535) { if (lane == 0) atomicMax(&stats[2 * b + 1], 0x7ffffff0); return; }   // absurd density: reported as an overflow

        for (int t0 = 0; t0 < T; t0 += kPairChunks * 64) {
            float cx[kPairChunks], cy[kPairChunks], cz[kPairChunks];
            uint32_t ck[kPairChunks];                                      // the entry (j << 10 | image) each candidate would become
            int staged_t0 = -1, shcur = 0;                                 // (kept in registers only inside a batch: holding them across the
                                                                           //  sort of a block of rows costs 46 VGPRs, i.e. two wavefronts per SIMD)
            // ---- a batch of candidates into registers, for molecules of shift `sh` -------------------------------
            auto stage = [&](int sh) __attribute__((always_inline)) {
                const int s0 = (sh & 1023) - 512, s1 = ((sh >> 10) & 1023) - 512, s2 = ((sh >> 20) & 1023) - 512;
#pragma unroll
                for (int ch = 0; ch < kPairChunks; ++ch) {
                    const int t = t0 + ch * 64 + lane;
                    cx[ch] = 3.0e18f; cy[ch] = 3.0e18f; cz[ch] = 3.0e18f; ck[ch] = kNoEntry;   // no candidate: never within range
                    if (t0 + ch * 64 < T && t < T) {
                        const int r = piece_of(W, t, npieces);
                        const int q = W.pq[r] + (t - W.pstart[r]);
                        const float4 v = WP[q];
                        const int shj = WS[q];
                        const int pkr = W.pk[r];
                        int k = pkr & 1023;                                                      // same shift (the rule): the piece's image
                        if (__builtin_amdgcn_ballot_w64(shj != sh) != 0ull) {                    // wave-uniform
                            if (shj != sh) {
                                const int po = pkr >> 10;
                                const int m0 = ((po & 3) - 1) + s0 - ((shj & 1023) - 512);
                                const int m1 = (((po >> 2) & 3) - 1) + s1 - (((shj >> 10) & 1023) - 512);
                                const int m2 = (((po >> 4) & 3) - 1) + s2 - (((shj >> 20) & 1023) - 512);
                                k = -1;
                                if (!(m0 < -im0 || m0 > im0 || m1 < -im1 || m1 > im1 || m2 < -im2 || m2 > im2)) {   // in the image table
                                    const int lin = ((m0 + im0) * w1 + (m1 + im1)) * w2 + (m2 + im2);
                                    k = lin == central ? 0 : (lin < central ? lin + 1 : lin);                        // molint.F90:197-213
                                }
                            }
                        }
                        if (k >= 0) {
                            cx[ch] = v.x + W.poff[r][0]; cy[ch] = v.y + W.poff[r][1]; cz[ch] = v.z + W.poff[r][2];
                            ck[ch] = ((uint32_t)__builtin_bit_cast(int, v.w) << 10) | (uint32_t)k;
                        }
                    }
                }
            };
            const int nch = min(kPairChunks, (T - t0 + 63) >> 6);          // chunks of this batch that hold candidates
            // ---- every molecule of the block against the batch -------------------------------------------------
            // The chunk loop is compiled once per chunk count with NO branch inside: the chunks of a molecule are
            // independent dependency chains (subtract, square, compare, ballot, mbcnt, store), and only straight-line
            // code lets the compiler interleave them -- with a branch per chunk (the earlier "skip the chunks outside the
            // molecule's three slabs") every chain ran exposed, which cost more than the tests it saved (1.35 -> 1.25 ms
            // for the whole list build with all chunks tested).  What is needed once per molecule -- "some hit was too
            // close to call", "how many hits are inside the energy cutoff" -- is kept per lane and reduced once.
            auto run_batch = [&](auto nch_c) __attribute__((always_inline)) {
                constexpr int NCH = decltype(nch_c)::value;
                for (int il = 0; il < nib; ++il) {
                    const int shi = __builtin_amdgcn_readfirstlane(W.ownsh[il]);
                    if (staged_t0 != t0 || shi != shcur) { shcur = shi; staged_t0 = t0; stage(shcur); }   // (again only for a molecule that left the box unwrapped)
                    const float4 o = W.own[il];
                    const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, o.x)));
                    const float yi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, o.y)));
                    const float zi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, o.z)));
                    // (first batch: the rows are empty.  Later batches: the count WITHOUT the ambiguity flag in bit 30 -- with it,
                    //  every later hit of a flagged row landed "past the row's end", was counted but not stored, and the sort read
                    //  whatever LDS held in its place: a wild molecule index in the double-precision re-decision, a GPU fault)
                    int cnt = t0 == 0 ? 0 : (__builtin_amdgcn_readfirstlane(W.cnt[il]) & 0x3fffffff);
                    int inner_l = 0, amb_l = 0;
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        const float dx = cx[ch] - xi, dy = cy[ch] - yi, dz = cz[ch] - zi;
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        // (one compare per ballot: the molecule itself, r2 = 0, is let in here and dropped when its row is sorted)
                        const bool hit = r2 < rn2_hi;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                        int p = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                        if (hit && p < rowcap) W.rows[il * rowcap + p] = ck[ch];
                        cnt += __popcll(m);
                        inner_l += r2 < (float)kRcSq ? 1 : 0;
                        amb_l = __builtin_fabsf(r2 - (float)kRnSq) < eps_ ? 1 : amb_l;  // inside the band: a hit, and ambiguous
                    }
                    const int ninner = __builtin_amdgcn_readlane(dpp_wave_sum_i32(inner_l), 63);
                    const unsigned long long amb = __builtin_amdgcn_ballot_w64(amb_l != 0);
                    if (lane == 0) {
                        W.cnt[il] = (cnt & 0x3fffffff) | (amb != 0ull ? 0x40000000 : (W.cnt[il] & 0x40000000));
                        W.ncin[il] += ninner;                              // (counts the molecule itself once: taken off below)
                    }
                }
            };
            if (nch <= 4) run_batch(std::integral_constant<int, 4>{}); else run_batch(std::integral_constant<int, kPairChunks>{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- sort the rows and write them: LPR lanes per row (two rows per pass unless a row is long) ----------------
        int longest = lane < nib ? (W.cnt[lane] & 0x3fffffff) : 0;
        longest = __builtin_amdgcn_readfirstlane(wave_max_i(longest));
        const int lpr = longest > 32 ? 64 : 32, rpp = 64 / lpr;                       // lanes per row, rows per pass
        for (int r0 = 0; r0 < nib; r0 += rpp) {
            const int sub = lane / lpr, sl = lane - sub * lpr;
            const int il = r0 + sub;
            const bool rowok = il < nib;
            const int craw = rowok ? W.cnt[il] : 0;
            const int nraw = craw & 0x3fffffff;
            const int nrow = nraw < rowcap ? nraw : rowcap;
            const bool active = sl < nrow;
            uint32_t key = active ? W.rows[il * rowcap + sl] : kNoEntry;
            int i = 0;
            if (rowok) i = __builtin_bit_cast(int, W.own[il].w);
            if (key == ((uint32_t)i << 10)) key = kNoEntry;                   // (i, central image) is not an entry (molint.F90:532)
            if (__builtin_amdgcn_ballot_w64((craw & 0x40000000) != 0) != 0ull) {
                // a row with a hit too close to call in single precision: every entry of it by the reference's
                // expression, unfused (:529-537); what fails goes (about one row in 2000)
                if (active && key != kNoEntry && (craw & 0x40000000)) {
#pragma clang fp contract(off)
                    const int j = (int)(key >> 10), k = (int)(key & 1023u);
                    const double vx = P[3 * j] - P[3 * i], vy = P[3 * j + 1] - P[3 * i + 1], vz = P[3 * j + 2] - P[3 * i + 2];
                    const double tx = vx + IV[3 * k], ty = vy + IV[3 * k + 1], tz = vz + IV[3 * k + 2];
                    const double r2 = tx * tx + ty * ty + tz * tz;
                    if (!(r2 < kRnSq)) key = kNoEntry;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (rowok && sl < rowcap) W.rows[il * rowcap + sl] = key;                                   // the rank loop reads the rows from LDS: dropped entries and the
                                                                               // lanes past the row's end as kNoEntry (never smaller than a key)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool valid = key != kNoEntry;
            const int nloop = __builtin_amdgcn_readfirstlane(wave_max_i(nrow));
            int rank = 0;                                                     // keys are unique; dropped ones sort last
            const uint4* k4 = reinterpret_cast<const uint4*>(&W.rows[(rowok ? il : 0) * rowcap]);
            for (int e = 0; e < nloop; e += 4) {
                const uint4 kk = k4[e >> 2];
                rank += (kk.x < key ? 1 : 0) + (kk.y < key ? 1 : 0) + (kk.z < key ? 1 : 0) + (kk.w < key ? 1 : 0);
            }
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid), bm = __builtin_amdgcn_ballot_w64(valid && (key & 1023u) != 0u);
            const unsigned long long rowmask = lpr == 64 ? ~0ull : (0xffffffffull << (32 * sub));
            const int nvalid = __popcll(vm & rowmask), nbnd = __popcll(bm & rowmask);
            const int ncin = rowok ? max(W.ncin[il] - 1, 0) : 0;
            if (valid && rank < S) LM[(size_t)i * kRow + rank] = pack_entry((int)(key >> 10), (int)(key & 1023u));
            if (rowok && sl == 0) {
                const int total = nraw > rowcap ? nraw : nvalid;        // more single-precision hits than a row holds: reported as an overflow
                nn[(size_t)b * N + i] = nvalid < S ? nvalid : S;
                cin[(size_t)b * N + i] = (unsigned char)((ncin < 127 ? ncin : 127) | (nbnd ? 0x80 : 0));
                wmin = total < wmin ? total : wmin;
                wmax = total > wmax ? total : wmax;
            }
        }
        __builtin_amdgcn_wave_barrier();                                      // rows are reused by the next block of molecules
    }
    wmin = wave_min_i(wmin); wmax = wave_max_i(wmax);
    if (lane == 0) {
        // thousands of wavefronts of a box report to the same two words: look first (the values only ever move one way,
        // so a stale read can only cause a superfluous atomic, never a missed one) and touch them only to change them
        if (wmin < __hip_atomic_load(&stats[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&stats[2 * b], wmin);
        if (wmax > __hip_atomic_load(&stats[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[2 * b + 1], wmax);
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
//   grid = (segments, boxes), block = min(1024, segment length): a 64-molecule segment of a large box is one wavefront's work
//   (with 1024 threads fifteen of sixteen wavefronts idled through every barrier: 583 us instead of ~150 for 64 x 32768)
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
    const int NT = (int)blockDim.x;                                   // 1024 for a whole LDS-sized box, 64 for the 64-molecule segments of large boxes
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
        for (int i = i0 + tid; i < i1; i += NT) { ORD[i] = i; NNS[i] = min(NN[i], S); }
    } else {
        // hist[group][key]: the lanes of a wavefront share the group and differ in key, i.e. in LDS bank (the scan below
        // walks it in (key, group) order: element e is key e / ngroups, group e % ngroups)
        const int K = 1 << kbits, M = K * ngroups;
        auto hslot = [&](int key, int grp) { return grp * K + key; };
        for (int e = tid; e < M; e += NT) hist[e] = 0;
        __syncthreads();
        for (int i = i0 + tid; i < i1; i += NT) atomicAdd(&hist[hslot(keyof(i, NN[i]), (i - i0) >> 6)], 1);
        __syncthreads();
        // exclusive scan of hist[0..M): `per` consecutive elements per thread
        const int per = (M + NT - 1) / NT;
        const int e0 = min(M, tid * per), e1 = min(M, e0 + per);
        int local = 0;
        for (int e = e0; e < e1; ++e) local += hist[hslot(e / ngroups, e % ngroups)];
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int run = incl - local;
        for (int w = 0; w < wid; ++w) run += wsum[w];
        for (int e = e0; e < e1; ++e) { const int sl = hslot(e / ngroups, e % ngroups), v = hist[sl]; hist[sl] = run; run += v; }
        __syncthreads();
        for (int base = i0; base < i1; base += NT) {
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
                const int dst = i0 + hist[hslot(key, (i - i0) >> 6)] + rank;
                ORD[dst] = i; NNS[dst] = min(n, S);
            }
        }
    }
    __syncthreads();   // ORD / NNS of this segment are read back below by other threads of this workgroup
    // Column t <- row ORD[t], entries of the central image first: the full-box kernel then skips the image-vector
    // gather for the slots every lane of a wavefront knows to be central (c0min).  NNS[t] = n | n0 << 8 (n0 = central
    // entries), CM[group] = longest row | smallest n0 << 8.
    for (int base = i0; base < i1; base += NT) {
        const int t = base + tid;
        const bool valid = t < i1;
        const int i = valid ? ORD[t] : 0;
        const int n = valid ? NNS[t] : 0;
        const int nmax = __builtin_amdgcn_readfirstlane(wave_max_i(n));
        const uint4* row = reinterpret_cast<const uint4*>(LM + (size_t)i * kRow);
        int w = 0;                                            // next slot of the column
        // rows of up to 32 entries (the usual case) are fetched whole before anything is stored: eight independent
        // 16-byte loads in flight instead of a load -> store chain per four entries
        const bool fastrow = nmax <= 32;
        uint4 rv[8];
        if (fastrow) {
#pragma unroll
            for (int u = 0; u < 8; ++u) rv[u] = (4 * u < n) ? row[u] : make_uint4(0u, 0u, 0u, 0u);
        }
        // a wavefront of interior molecules (the sort key groups them): every entry is of the central image, the column
        // is the row as it stands -- one coalesced store per slot, no second pass
        const bool interior = __ballot(valid && (CI[i] >> 7) != 0) == 0ull;
        if (fastrow && interior) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (4 * u < nmax) {
                    const uint32_t e[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (4 * u + v < nmax && valid) L[(size_t)(4 * u + v) * N + t] = 4 * u + v < n ? e[v] : 0u;
                }
            }
            if (valid) NNS[t] = n | (n << 8);
            const int c0min = __builtin_amdgcn_readfirstlane(wave_min_i(valid ? n : 0x7fff));
            if (lane == 0 && valid) CM[t >> 6] = nmax | ((c0min > 255 ? 255 : c0min) << 8);
            continue;
        }
        if (fastrow) {
            // one pass: count the central-image entries first (registers only), then every entry goes straight to its
            // slot -- central ones from 0 up, the others from n0 up -- with ONE store per entry instead of two predicated ones
            int n0 = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t e[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
                for (int v = 0; v < 4; ++v) n0 += (4 * u + v < n && (e[v] >> kJBits) == 0u) ? 1 : 0;
            }
            int wc = 0, wb = n0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (4 * u < nmax) {
                    const uint32_t e[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        if (4 * u + v < n) {
                            const bool central = (e[v] >> kJBits) == 0u;
                            const int dst = central ? wc : wb;
                            L[(size_t)dst * N + t] = e[v];
                            wc += central ? 1 : 0; wb += central ? 0 : 1;
                        }
                    }
                }
            }
            if (valid) {
                for (int z = n; z < nmax; ++z) L[(size_t)z * N + t] = 0u;          // zero-padded to the group's longest row
                NNS[t] = n | (n0 << 8);
            }
            const int c0min = __builtin_amdgcn_readfirstlane(wave_min_i(valid ? n0 : 0x7fff));
            if (lane == 0 && valid) CM[t >> 6] = nmax | ((c0min > 255 ? 255 : c0min) << 8);
            continue;
        }
        for (int pass = 0; pass < 2; ++pass) {                // rows longer than 32 entries: two passes over the row in memory
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
