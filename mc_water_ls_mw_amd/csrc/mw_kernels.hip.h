// mw_kernels.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine.
//
// What each kernel stands behind in the reference (keb721/mc_water_ls_mw):
//   k_build_neighbours  compute_neighbours       molint.F90:501-559
//   k_model_energy      compute_model_energy     molint.F90:407-499
//   k_move_energy / k_local_energy_single   compute_local_real_energy molint.F90:220-404
// None of it is a translation: the list is slot-major and packed for coalesced
// reads, positions of a whole box are staged in LDS, the three-body sum is
// evaluated from per-atom moments in O(neighbours), and the single-move path
// maps one request onto one 64-wide wavefront.  Double precision throughout;
// this is gather + transcendental work, so no MFMA.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mw {

// ---- model constants (molint.F90:63-74, constants.f90:42-43) -----------------------
constexpr double kAngToBohr = 1.0 / 0.5291772108;
constexpr double kSigma     = 2.3925 * kAngToBohr;       // bohr
constexpr double kEpsilon   = 6.189 / 627.509469;        // Hartree
constexpr double kLambda    = 23.15;
constexpr double kBigA      = 7.049556277;
constexpr double kBigB      = 0.6022245584;
constexpr double kGamma     = 1.2;
constexpr double kSmallA    = 1.8;
// molint.F90:74 has no _dp suffix: the reference holds float32(-0.33331324756) widened
constexpr double kCos0      = (double)(-0.33331324756f);
constexpr double kSigA      = kSigma * kSmallA;                       // rc = a*sigma
constexpr double kRcSq      = kSigma * kSmallA * kSigma * kSmallA;    // molint.F90:255,432 order
constexpr double kRn        = kSmallA * kSigma * 1.18;                // molint.F90:516
constexpr double kRnSq      = kRn * kRn;                              // molint.F90:537
constexpr double kAeps      = kBigA * kEpsilon;
constexpr double kLamEps    = kLambda * kEpsilon;
constexpr double kGamSig    = kGamma * kSigma;
constexpr double kSigSq     = kSigma * kSigma;

// ---- packed list entry: (jmol-1) in the low 22 bits, (image-1) in the next 10 -------
constexpr int      kJBits = 22;
constexpr uint32_t kJMask = (1u << kJBits) - 1u;

__device__ __forceinline__ uint32_t pack_entry(int j0, int k0) { return (uint32_t)j0 | ((uint32_t)k0 << kJBits); }

// The list is kept in two layouts, each coalesced for its consumer:
//   list  [box][S][N]   slot-major     -- full-box kernel: thread = molecule, loop over slots
//   listm [box][N][64]  molecule-major -- single-move kernels: lane = slot of one molecule's row
// (a row is 256 B = two 128-B lines; S <= 64)
constexpr int kRow = 64;

// Wave-uniform broadcast of a double from lane `l` (l must be uniform: v_readlane, no LDS traffic).
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// ---- double-precision primitives sized for this tolerance ---------------------------------
// The parity bar is 1e-10 relative on energies; these keep every factor below 1e-14 relative
// while costing a fraction of the IEEE-exact sqrt / divide / libm exp sequences (which spend
// most of their instructions on the last ulp and on special cases that cannot occur here:
// the arguments are finite, positive (r^2), nonzero (r - a sigma < 0) or <= 0 (exponent)).

// The gfx950 v_rsq_f64 / v_rcp_f64 estimates are good to ~5e-8 relative (measured, tools/hwprec.hip);
// one third-order correction brings both to double rounding (1.4e-16 / <1e-16 measured).

// 1/sqrt(x), x > 0 finite and normal.
__device__ __forceinline__ double fast_rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * y, y, 1.0);                 // 1 - x y^2
    return __builtin_fma(y, e * __builtin_fma(e, 0.375, 0.5), y);   // y (1 + e/2 + 3e^2/8)
}

// 1/x, x finite, normal, nonzero.
__device__ __forceinline__ double fast_rcp(double x)
{
    const double y = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, y, 1.0);                     // 1 - x y
    return __builtin_fma(y, __builtin_fma(e, e, e), y);             // y (1 + e + e^2)
}

// d = a*b + c with c in a scalar register pair: the three-address v_fma_f64.  (Left to itself the
// compiler keeps the Horner coefficients in VGPRs and emits v_mov_b64 + v_fmac_f64 per step.)
__device__ __forceinline__ double fma_sc(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

// exp(x) for x <= 0 (any magnitude; underflows smoothly to 0).  n = round(x log2 e),
// r = x - n ln2 in two pieces, degree-11 Taylor on |r| <= 0.347 (remainder < 7e-15), 2^n by ldexp.
__device__ __forceinline__ double fast_exp_neg(double x)
{
    x = __builtin_fmax(x, -800.0);                                  // exp(-800) == 0 in double anyway
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);    // ln2 high part (fdlibm split)
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);           // ln2 low part
    double p = 2.50521083854417187751e-08;                           // 1/11!
    p = fma_sc(p, r, 2.75573192239858906526e-07);                    // 1/10!
    p = fma_sc(p, r, 2.75573192239858906526e-06);                    // 1/9!
    p = fma_sc(p, r, 2.48015873015873015873e-05);                    // 1/8!
    p = fma_sc(p, r, 1.98412698412698412698e-04);                    // 1/7!
    p = fma_sc(p, r, 1.38888888888888888889e-03);                    // 1/6!
    p = fma_sc(p, r, 8.33333333333333333333e-03);                    // 1/5!
    p = fma_sc(p, r, 4.16666666666666666667e-02);                    // 1/4!
    p = fma_sc(p, r, 1.66666666666666666667e-01);                    // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);
}

// ---- wave / block reductions ---------------------------------------------------------
// Inclusive prefix sums over the 64 lanes through the DPP network: four shifts inside each row of 16
// lanes, then lane 15 of a row into the next row and lane 31 into the upper half.  Lane 63 ends up with
// the wave's total (summation order: a fixed tree, the same for every call).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_wave_sum(double v)
{
    v += dpp_mov_f64<0x111, 0xf>(v);      // row_shr:1
    v += dpp_mov_f64<0x112, 0xf>(v);      // row_shr:2
    v += dpp_mov_f64<0x114, 0xf>(v);      // row_shr:4
    v += dpp_mov_f64<0x118, 0xf>(v);      // row_shr:8
    v += dpp_mov_f64<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_mov_f64<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int dpp_wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // valid in lane 0
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_down(v, off, 64));
    return v;
}

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
                        const int* __restrict__ nivect, uint32_t* __restrict__ list,
                        uint32_t* __restrict__ listm, int* __restrict__ nn, int* __restrict__ stats,
                        const int* __restrict__ use_grid, int N, int S, int ivcap, int box0)
{
#pragma clang fp contract(off)
    const int b = box0 + blockIdx.y;
    if (use_grid[b]) return;              // this box goes through k_cell_search
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double* P  = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    const int niv = nivect[b];
    uint32_t* L = list + (size_t)b * S * N;
    uint32_t* LM = listm + ((size_t)b * N + (i < N ? i : 0)) * kRow;
    const bool active = i < N;
    const int ii = active ? i : 0;
    const double xi = P[3 * ii], yi = P[3 * ii + 1], zi = P[3 * ii + 2];   // molint.F90:522
    int cnt = 0;

    for (int j = 0; j < N; ++j) {                                           // :525
        const double vx = P[3 * j] - xi, vy = P[3 * j + 1] - yi, vz = P[3 * j + 2] - zi;   // :529
        for (int k = 0; k < niv; ++k) {                                     // :531
            const double tx = vx + IV[3 * k], ty = vy + IV[3 * k + 1], tz = vz + IV[3 * k + 2];   // :534
            const double r2 = tx * tx + ty * ty + tz * tz;                  // :535
            if (r2 < kRnSq && !(k == 0 && j == i)) {                        // :532,537
                if (active && cnt < S) { const uint32_t e = pack_entry(j, k); L[(size_t)cnt * N + i] = e; LM[cnt] = e; }
                ++cnt;
            }
        }
    }
    if (active) nn[(size_t)b * N + i] = cnt < S ? cnt : S;

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
//     enumeration order, before they are written in both list layouts.
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
                   uint32_t* __restrict__ list, uint32_t* __restrict__ listm,
                   int* __restrict__ nn, int* __restrict__ stats,
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
    int cnt = 0;

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
                        }
                    }
                }
            }
        }
        // rank sort (keys are unique): entry a goes to slot #{keys smaller than key a}
        const int n = cnt < S ? cnt : S;
        uint32_t* L = list + (size_t)b * S * N;
        uint32_t* LM = listm + ((size_t)b * N + i) * kRow;
        for (int a = 0; a < n; ++a) {
            const uint32_t ka = buf[a * 256];
            int r = 0;
            for (int c = 0; c < n; ++c) r += (buf[c * 256] < ka) ? 1 : 0;
            const uint32_t e = pack_entry((int)(ka >> 10), (int)(ka & 1023u));
            L[(size_t)r * N + i] = e;
            LM[r] = e;
        }
        nn[(size_t)b * N + i] = n;
    }
    int mn = wave_min_i(active ? cnt : 0x7fffffff);
    int mx = wave_max_i(active ? cnt : 0);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&stats[2 * b], mn);
        atomicMax(&stats[2 * b + 1], mx);
    }
}

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

__device__ __forceinline__ void load_pos(const double* __restrict__ P, int j, const Override& o1, const Override& o2,
                                         double& x, double& y, double& z)
{
    const double* p = P + 3 * (size_t)j;
    x = p[0]; y = p[1]; z = p[2];
    if (j == o1.idx) { x = o1.x; y = o1.y; z = o1.z; }
    if (j == o2.idx) { x = o2.x; y = o2.y; z = o2.z; }
}

__device__ __forceinline__ void pair_terms(double r2, double& rinv, double& e1, double& g)
{
    rinv = fast_rsqrt(r2);                       // molint.F90:278
    const double den = fma_sc(r2, rinv, -kSigA); // r - a sigma, r = r2 / sqrt(r2)        :286
    // r2 < rc^2 but r rounded onto rc: the term is exactly 0 in that limit          :288
    const double w = fast_rcp(__builtin_fmin(den, -1.0e-300));
    const double t = fast_exp_neg(0.2 * kSigma * w);
    const double t2 = t * t, t4 = t2 * t2;
    e1 = t4 * t;                                 // :291
    g  = t4 * t2;                                // :292
}

// Returns the local energy in every lane.  `ninter` / `nslots` (wave-uniform) receive the number
// of in-range interactions as the reference enumerates them (pairs + triplet slots with
// cos(theta) < 0.99) and the number of list slots visited (n_i + sum of n_j over in-range j),
// which prices the call's algorithmic bytes.
__device__ __forceinline__ double local_energy_wave(const double* __restrict__ P, const double* __restrict__ IV,
                                                    const uint32_t* __restrict__ LM, const int* __restrict__ NN,
                                                    int i, const Override& o1, const Override& o2, int lane,
                                                    unsigned int& ninter, unsigned int& nslots)
{
    double xi, yi, zi;
    load_pos(P, i, o1, o2, xi, yi, zi);                                   // molint.F90:258
    const int n_i = NN[i];

    // pass 0: imol's own list, one slot per lane
    const bool has = lane < n_i;
    const uint32_t e = has ? LM[(size_t)i * kRow + lane] : 0u;
    const int j = (int)(e & kJMask), kimg = (int)(e >> kJBits);
    double xj, yj, zj;
    load_pos(P, j, o1, o2, xj, yj, zj);
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
        const double ajx = __shfl(dx, jl, 64), ajy = __shfl(dy, jl, 64), ajz = __shfl(dz, jl, 64);
        const double rinv_j = __shfl(rinv, jl, 64), g_j = __shfl(g, jl, 64);

        // j--i--k: later in-range slots of imol's own list                 :302-318
        if (inr && lane > jl) {
            const double ct = ((ajx * dx + ajy * dy + ajz * dz) * rinv_j) * rinv;     // :316,365
            if (ct < 0.99) { const double d = ct - kCos0; acc3 += g_j * (g * (d * d)); ++ntl; }   // :367-368,385-387
        }

        // i--j--k: jmol's list, translated by j's image                    :324-343
        const int jj = __shfl(j, jl, 64);
        const double sjx = __shfl(jvx, jl, 64), sjy = __shfl(jvy, jl, 64), sjz = __shfl(jvz, jl, 64);
        const double pjx = __shfl(qx, jl, 64), pjy = __shfl(qy, jl, 64), pjz = __shfl(qz, jl, 64);
        const int n_j = NN[jj];
        nslots += (unsigned int)n_j;
        if (lane < n_j) {
            const uint32_t e2 = LM[(size_t)jj * kRow + lane];
            const int kk = (int)(e2 & kJMask), k2 = (int)(e2 >> kJBits);
            double xk, yk, zk;
            load_pos(P, kk, o1, o2, xk, yk, zk);
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
    double tot = acc2 + kLamEps * acc3;                                    // :397
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tot += __shfl_xor(tot, off, 64);
        ntl += (unsigned int)__shfl_xor((int)ntl, off, 64);
    }
    ninter += ntl;
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
    double rinvo[kCap], rinvn[kCap];           // 1/r_ij at the old / trial position
    double go[kCap], gn[kCap];                 // exp(gamma sigma/(r_ij - a sigma)) old / trial
    int flag[kCap];                            // bit0 = in range of the old position, bit1 = of the trial position
    unsigned long long cm[kCap];               // bit p of the end-to-end slot numbering set: a row ends at slot p
    uint32_t qe[64];                           // queue of in-range third bodies: packed list entry ...
    int qown[64];                              // ... and rank | (image, inverse image, flags of that rank) << 5 of the j whose row it came from
};
static_assert(sizeof(WaveScratch) % 8 == 0, "scratch records must keep 8-byte alignment");

// Returns false (nothing written) when the request needs the plain routine.
// `row(j, s)` returns list entry s of molecule j and `nnof(j)` its row length: global memory (molecule-major
// list) or, for small systems in the sweep driver, LDS copies.
template <typename PosFn, typename IvFn, typename RowFn, typename NnFn>
__device__ __forceinline__ bool move_energy_wave(PosFn getpos, IvFn getiv, RowFn row, NnFn nnof,
                                                 WaveScratch* __restrict__ ws, int niv,
                                                 int i, int n_i, uint32_t e,
                                                 double xo, double yo, double zo,
                                                 double xn, double yn, double zn, int lane, MoveRes& res)
{
    // ---- pass 0: imol's own row; lanes 0..31 take slot l against the OLD position, lanes 32..63 the same
    // slot against the TRIAL position, so that one rsqrt/reciprocal/exp sequence serves both evaluations.
    // `e` arrives as entry (lane & 31) of imol's row, fetched by the caller ahead of time (whatever the row
    // length: rows are padded).  Rows longer than 32 entries take the plain routine, and so does a molecule
    // that neighbours one of its own periodic images.
    if (n_i > 32) return false;
    const int half = lane >> 5, sl = lane & 31;
    const bool has = sl < n_i;
    const int j = has ? (int)(e & kJMask) : 0, kimg = has ? (int)(e >> kJBits) : 0;
    if (__ballot(has && j == i) != 0ull) return false;
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

    double rinv = 0.0, e1 = 0.0, g = 0.0;
    if (in) pair_terms(r2, rinv, e1, g);
    const double qq = kSigSq * rinv * rinv;
    const double accp = in ? (kAeps * (kBigB * (qq * qq) - 1.0)) * e1 : 0.0;  // :294-297 (old in lanes 0..31, trial in 32..63)
    double t3o = 0.0, t3n = 0.0;
    unsigned int nto = 0, ntn = 0;

    // ---- compact the in-range neighbours (of either position) into the wave's scratch ------------
    const bool inu = (U >> sl) & 1u;
    const int rank = __popc(U & ((1u << sl) - 1u));
    // The rows of the in-range j are laid end to end (slots 0..T-1).  An inclusive prefix sum over the 32
    // slot lanes of each half gives every j its first slot, and in its upper 16 bits the list slots each
    // evaluation visits (half 0: old position, half 1: trial position).
    const int mine = (inu ? nnj : 0) | ((in ? nnj : 0) << 16);
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
    const int lin = kimg <= cc ? kimg - 1 : kimg, linv = niv - 1 - lin;
    const int kinv = kimg == 0 ? 0 : (linv < cc ? linv + 1 : linv);
    // image (10 bits) | inverse image (10 bits) | in range of old, trial position (2 bits), by rank like jv
    const int flg = (int)((mo_ >> sl) & 1u) | (int)(((mn_ >> sl) & 1u) << 1);
    const int wv = __builtin_amdgcn_ds_permute(dstl << 2, kimg | (kinv << 10) | (flg << 20));
    // row-end marks: chunk c of the scan reads mask cm[c]; a slot's owner is the number of marks before it
    if (lane < kCap) ws->cm[lane] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (inu && half == 0 && rank > 0 && start > 0)      // (rows are never empty: j lists i back)
        __hip_atomic_fetch_or(&ws->cm[(start - 1) >> 6], 1ull << ((start - 1) & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (inu) {
        if (half == 0) {
            ws->q[0][rank] = qx; ws->q[1][rank] = qy; ws->q[2][rank] = qz;
            ws->rinvo[rank] = rinv; ws->go[rank] = g;
            ws->flag[rank] = flg;
        } else {
            ws->rinvn[rank] = rinv; ws->gn[rank] = g;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- rows of the in-range j: fetched ahead ----------------------------------------------------
    // The i--j--k stage below walks the rows of all in-range j laid end to end, 64 slots per chunk.  A
    // chunk's slot -> (owner rank, owner's packed word, row entry) fetch is issued TWO CHUNKS AHEAD of its
    // evaluation -- the first two right here, before the j--i--k stage -- so the row fetch (global memory for
    // the big boxes) is never waited for.
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

    // ---- j--i--k triplets: pairs (a < b) of in-range neighbours, one pair per lane ------------
    // (molint.F90:302-318; a is the earlier list slot, so cos is formed in the reference's order)
    const int npairs = cntU * (cntU - 1) / 2;
    for (int p0 = 0; p0 < npairs; p0 += 64) {
        const int p = p0 + lane;
        int b = (int)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        if (b * (b - 1) / 2 > p) --b;
        if ((b + 1) * b / 2 <= p) ++b;
        const int a = p - b * (b - 1) / 2;
        if (p < npairs) {
            // every operand in one batch of LDS reads (one round trip), whichever positions are in range
            const int fa = ws->flag[a], fb = ws->flag[b];
            const double qax = ws->q[0][a], qay = ws->q[1][a], qaz = ws->q[2][a];
            const double qbx = ws->q[0][b], qby = ws->q[1][b], qbz = ws->q[2][b];
            const double roa = ws->rinvo[a], rob = ws->rinvo[b], goa = ws->go[a], gob = ws->go[b];
            const double rna = ws->rinvn[a], rnb = ws->rinvn[b], gna = ws->gn[a], gnb = ws->gn[b];
            if (fa & fb & 1) {
                const double ct = (((qax - xo) * (qbx - xo) + (qay - yo) * (qby - yo) + (qaz - zo) * (qbz - zo))
                                   * roa) * rob;                                                // :316,365
                if (ct < 0.99) { const double d = ct - kCos0; t3o += goa * (gob * (d * d)); ++nto; }
            }
            if (fa & fb & 2) {
                const double ct = (((qax - xn) * (qbx - xn) + (qay - yn) * (qby - yn) + (qaz - zn) * (qbz - zn))
                                   * rna) * rnb;
                if (ct < 0.99) { const double d = ct - kCos0; t3n += gna * (gnb * (d * d)); ++ntn; }
            }
        }
    }

    // ---- i--j--k triplets (molint.F90:324-343): the rows of all in-range j, end to end --------
    // Two stages.  SCAN: every slot gets the cheap part (gather, distance test); the ~1/3 that are in
    // range are queued (entry + owner rank, 8 bytes) in the wave's scratch.  FLUSH: whenever 64 are
    // queued (and at the end) one full pass does the expensive part -- rsqrt, reciprocal, exp and the two
    // cosines -- with every lane busy, instead of three passes at one third occupancy.
    int nq = 0;                                              // queued entries (wave-uniform)
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < nq) {
            const uint32_t e2 = ws->qe[lane];
            const int qw = ws->qown[lane];
            const int own = qw & 31, kj = (qw >> 5) & 1023, fl = qw >> 25;
            const int kk = (int)(e2 & kJMask), k2 = (int)(e2 >> kJBits);
            double xk, yk, zk, kvx, kvy, kvz, sjx, sjy, sjz;
            getpos(kk, xk, yk, zk);
            getiv(k2, kvx, kvy, kvz);
            getiv(kj, sjx, sjy, sjz);
            const double pjx = ws->q[0][own], pjy = ws->q[1][own], pjz = ws->q[2][own];
            const double ro = ws->rinvo[own], rn = ws->rinvn[own], go_ = ws->go[own], gn_ = ws->gn[own];
            const double bx = ((xk + kvx) + sjx) - pjx;                          // :332,334
            const double by = ((yk + kvy) + sjy) - pjy;
            const double bz = ((zk + kvz) + sjz) - pjz;
            const double s2 = bx * bx + by * by + bz * bz;                       // :335 (in range: tested at scan)
            double rk, gk, e1k;
            pair_terms(s2, rk, e1k, gk);
            if (fl & 1) {
                const double ct = (-((pjx - xo) * bx + (pjy - yo) * by + (pjz - zo) * bz) * ro) * rk;   // :320,341,365
                if (ct < 0.99) { const double d = ct - kCos0; t3o += go_ * (gk * (d * d)); ++nto; }
            }
            if (fl & 2) {
                const double ct = (-((pjx - xn) * bx + (pjy - yn) * by + (pjz - zn) * bz) * rn) * rk;
                if (ct < 0.99) { const double d = ct - kCos0; t3n += gn_ * (gk * (d * d)); ++ntn; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        nq = 0;
    };
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < T;
        const int own = own_a, wj = w_a;
        const uint32_t e2 = ent_a;
        own_a = own_b; w_a = w_b; ent_a = ent_b;
        if (t0 + 128 < T) fetch(t + 128, own_b, w_b, ent_b);
        const int kj = wj & 1023;
        const int kk = (int)(e2 & kJMask), k2 = (int)(e2 >> kJBits);
        double xk, yk, zk, kvx, kvy, kvz, sjx, sjy, sjz;
        getpos(kk, xk, yk, zk);
        getiv(k2, kvx, kvy, kvz);
        getiv(kj, sjx, sjy, sjz);
        const double pjx = ws->q[0][own], pjy = ws->q[1][own], pjz = ws->q[2][own];
        const bool self = valid && (kk == i);
        const bool selfimg = self && (k2 == ((wj >> 10) & 1023));   // the molecule itself, not an image: k's shift undoes j's
        const bool selfmove = self && !selfimg;
        const double box_ = ((xk + kvx) + sjx) - pjx;                            // :332,334
        const double boy_ = ((yk + kvy) + sjy) - pjy;
        const double boz_ = ((zk + kvz) + sjz) - pjz;
        const double s2o = box_ * box_ + boy_ * boy_ + boz_ * boz_;              // :335
        if (__ballot(selfmove) != 0ull) {
            // an image of the molecule itself as third body moves with it: both geometries, in line (rare)
            if (selfmove) {
                const int fl = wj >> 20;
                const double bnx = ((xn + kvx) + sjx) - pjx, bny = ((yn + kvy) + sjy) - pjy, bnz = ((zn + kvz) + sjz) - pjz;
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
        if (nq + c > 64) flush();
        if (inq) {
            const int slot = nq + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mq, 0u));
            ws->qe[slot] = e2; ws->qown[slot] = own | (wj << 5);
        }
        nq += c;
    }
    if (nq > 0) flush();
    __builtin_amdgcn_wave_barrier();                          // scratch is reused by the wave's next request

    // Wave sums on the DPP network (no LDS round trips): afterwards lane 63 holds the totals.
    const double eo = readlane_f64(dpp_wave_sum(kLamEps * t3o + (half == 0 ? accp : 0.0)), 63);   // :397
    const double en = readlane_f64(dpp_wave_sum(kLamEps * t3n + (half == 1 ? accp : 0.0)), 63);
    const unsigned int cs = (unsigned int)__builtin_amdgcn_readlane(dpp_wave_sum_i32((int)(nto | (ntn << 16))), 63);
    nto = cs & 0xffffu; ntn = cs >> 16;
    res.eo = eo; res.en = en;
    res.io = (unsigned int)__popc(mo_) + nto; res.in_ = (unsigned int)__popc(mn_) + ntn;
    res.so = so; res.sn = sn;
    return true;
}

// One workgroup per work item {box, first request, last request+1}: the requests are
// sorted by box on upload, so the workgroup stages that box's positions in LDS once
// (LDSPOS) and its 16 wavefronts then serve the item's requests from LDS gathers.
//   mode bit 0: write e_old (mirrored positions), bit 1: write e_new (trial position)
template <bool LDSPOS>
__global__ __launch_bounds__(1024)
void k_move_energy(const double* __restrict__ pos, const double* __restrict__ ivect,
                   const int* __restrict__ nivect, const uint32_t* __restrict__ listm,
                   const int* __restrict__ nn, const int4* __restrict__ work,
                   const int* __restrict__ req_imol, const double* __restrict__ req_trial,
                   const int* __restrict__ perm,
                   double* __restrict__ e_old, double* __restrict__ e_new,
                   unsigned int* __restrict__ counts,   // [nreq][4]: inter_old, slots_old, inter_new, slots_new
                   int N, int ivcap, int mode)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int4 w = work[blockIdx.x];
    const int b = w.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* P  = pos + (size_t)b * N * 3;
    const double* IV = ivect + (size_t)b * ivcap * 3;
    const uint32_t* LM = listm + (size_t)b * N * kRow;
    const int* NN = nn + (size_t)b * N;
    const int niv = nivect[b];

    // dynamic LDS: [positions when LDSPOS][image vectors][16 wave scratches][row lengths, one byte each, when
    // LDSPOS] (positions at offset 0: a gather's address is one multiply and the ds_read offsets are immediates)
    double* spos = smem;
    double* siv = smem + (LDSPOS ? 3 * (size_t)N : 0);
    WaveScratch* ws = reinterpret_cast<WaveScratch*>(siv + (size_t)ivcap * 3) + wave;
    unsigned char* snn = reinterpret_cast<unsigned char*>(reinterpret_cast<WaveScratch*>(siv + (size_t)ivcap * 3) + 16);
    for (int t = tid; t < niv * 3; t += 1024) siv[t] = IV[t];
    if (LDSPOS) {
        for (int t = tid; t < 3 * N; t += 1024) spos[t] = P[t];
        for (int t = tid; t < N; t += 1024) snn[t] = (unsigned char)NN[t];      // maxneigh <= 64
    }
    __syncthreads();

    auto getiv = [&](int k, double& x, double& y, double& z) { x = siv[3 * k]; y = siv[3 * k + 1]; z = siv[3 * k + 2]; };
    auto getpos = [&](int jx, double& x, double& y, double& z) {
        const double* p = LDSPOS ? (spos + 3 * (size_t)jx) : (P + 3 * (size_t)jx);
        x = p[0]; y = p[1]; z = p[2];
    };
    auto row = [&](int jx, int sl) { return LM[(size_t)jx * kRow + sl]; };
    auto nnof = [&](int jx) { return LDSPOS ? (int)snn[jx] : NN[jx]; };

    // The wave's requests are m = w.y + wave + 16 k.  Lane k fetches request k's molecule (and trial position)
    // up front; entry (lane & 31) of the molecule's own row is then fetched one request ahead of the one being
    // evaluated, so no request starts by waiting on memory.
    const int nmine = (w.z - w.y - wave + 15) / 16;                          // <= 64 (work items hold <= 1024 requests)
    const int mk = w.y + wave + 16 * lane;
    const int iall = lane < nmine ? req_imol[mk] : 0;
    double tx = 0.0, ty = 0.0, tz = 0.0;
    if ((mode & 2) && lane < nmine) { tx = req_trial[3 * (size_t)mk]; ty = req_trial[3 * (size_t)mk + 1]; tz = req_trial[3 * (size_t)mk + 2]; }
    uint32_t e_nx = nmine > 0 ? row(__builtin_amdgcn_readfirstlane(iall), lane & 31) : 0u;

    for (int k = 0; k < nmine; ++k) {
        const int m = w.y + wave + 16 * k;
        const int i = __builtin_amdgcn_readlane(iall, k);
        const uint32_t e = e_nx;
        if (k + 1 < nmine) e_nx = row(__builtin_amdgcn_readlane(iall, k + 1), lane & 31);
        double xo, yo, zo;
        getpos(i, xo, yo, zo);
        double xn = xo, yn = yo, zn = zo;
        if (mode & 2) { xn = readlane_f64(tx, k); yn = readlane_f64(ty, k); zn = readlane_f64(tz, k); }

        MoveRes r;
        const bool fast = move_energy_wave(getpos, getiv, row, nnof, ws, niv, i, nnof(i), e, xo, yo, zo, xn, yn, zn, lane, r);
        if (!fast) {
            Override none; none.idx = -1; none.x = none.y = none.z = 0.0;
            Override tr; tr.idx = i; tr.x = xn; tr.y = yn; tr.z = zn;
            r.eo = local_energy_wave(P, IV, LM, NN, i, none, none, lane, r.io, r.so);
            r.en = local_energy_wave(P, IV, LM, NN, i, tr, none, lane, r.in_, r.sn);
        }
        if (lane == 0) {
            const size_t o = (size_t)perm[m];
            if (mode & 1) { e_old[o] = r.eo; counts[4 * o] = r.io; counts[4 * o + 1] = r.so; }
            if (mode & 2) { e_new[o] = r.en; counts[4 * o + 2] = r.in_; counts[4 * o + 3] = r.sn; }
        }
    }
}

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
// Volume move of one walker by its wavefront: mc_volume (mc_moves.F90:1216-1534; MINU/leshift off; ref_ljr,
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
    const uint32_t* list_g;   // slot-major list, walker's first box
    const int* nn_g;
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
    const int* NN = c.nn_g + (size_t)l * c.N;
    auto getiv = [&](int k, double& x, double& y, double& z) { x = IVl[3 * k]; y = IVl[3 * k + 1]; z = IVl[3 * k + 2]; };
    auto getpos = [&](int j, double& x, double& y, double& z) {
        const double* p = Ps ? Ps + 3 * (size_t)j : Pg + 3 * (size_t)j;
        x = p[0]; y = p[1]; z = p[2];
    };
    double esum = 0.0;
    int i = lane;
    uint32_t cur[8];
    int n_cur = 0;
    if (i < c.N) {
        n_cur = NN[i];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = u < c.S ? Lg[(size_t)u * c.N + i] : 0u;
    }
    for (; i < c.N; i += 64) {
        const int inext = i + 64 < c.N ? i + 64 : -1;
        const int n_next = inext >= 0 ? NN[inext] : 0;
        AtomSum a = atom_energy<64>(i, n_cur, Lg, c.N, c.S, c.queue, getpos, getiv, cur, inext);
        esum += a.e;
        n_cur = n_next;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off, 64);
    return esum;
}

__device__ __forceinline__
int volume_move_wave(const VolCtx& c, const SweepParams& sp, const double* weight, const double* __restrict__ mu_bin,
                     const double* __restrict__ binwidth, double u0, double u1, double u2, double u3,
                     int ls, double& ls_mu, double men[2], int lane)
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
    int bad = 0;
    for (int l = 0; l < L; ++l) {                                                               // :1285-1358
        dev_rescale(c, l, recip_used[l], c.shmat + 9 * l, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int niv = dev_compute_ivects(c.shmat + 9 * l, c.siv + (size_t)l * c.ivcap * 3,
                                           c.ivect_g + (size_t)l * c.ivcap * 3, c.ivcap, lane);
        if (niv < 0) { bad = 1; break; }
        if (lane == 0) {
            c.svol[l] = fabs(dev_det3(c.shmat + 9 * l));
            double rcp[9];
            dev_recipmatrix(c.shmat + 9 * l, rcp);
#pragma unroll
            for (int t = 0; t < 9; ++t) c.srecip[l * 9 + t] = rcp[t];
            c.sniv[l] = niv; c.nivect_g[l] = niv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
            mu = mu * sp.beta - (double)N * log(c.svol[0] / c.svol[1]);
            ls_mu = mu;
            new_eta = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);
        }
        const double diffkT = sp.beta * dE + new_eta - old_eta + sp.beta * sp.pressure * (Vls - Vold)
                              - (double)N * log(Vls / Vold);                                     // :1381-1382
        double cmp = exp(-diffkT);
        cmp = cmp > 1.0 ? 1.0 : cmp;
        ok = u3 < cmp ? 1 : 0;                                                                   // :1410
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
        for (int l = 0; l < L; ++l) {
            dev_rescale(c, l, recip_new[l], c.shmat + 9 * l, lane);                              // back through the NEW recip
            const int niv = dev_compute_ivects(c.shmat + 9 * l, c.siv + (size_t)l * c.ivcap * 3,
                                               c.ivect_g + (size_t)l * c.ivcap * 3, c.ivcap, lane);   // :1510-1512
            if (lane == 0 && niv > 0) { c.sniv[l] = niv; c.nivect_g[l] = niv; }
        }
        men[0] = backup_e[0]; men[1] = backup_e[1];                                              // :1514
        if (L == 2) {                                                                            // :1516-1520
            double mu = (men[0] + sp.pressure * c.svol[0]) - (men[1] + sp.pressure * c.svol[1]);
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
template <bool LDSPOS, bool LDSLIST, bool WITHVOL>
__global__ __launch_bounds__(64)
void k_sweep_translation(double* pos, double* hmat, double* ivect,
                         int* nivect, const uint32_t* __restrict__ listm, const uint32_t* __restrict__ list,
                         const int* __restrict__ nn, double* __restrict__ energy,
                         int* __restrict__ wls, double* __restrict__ wmu, unsigned long long* __restrict__ wacc,
                         unsigned long long* __restrict__ wswitch, double* __restrict__ wshift,
                         SweepParams sp, double* wweight, double* whist, double* wuhist,
                         const double* __restrict__ mu_bin, const double* __restrict__ binwidth,
                         double* volume, unsigned long long* __restrict__ wvol, int* __restrict__ wflag,
                         int N, int S, int ivcap, int nmoves, unsigned long long seed, unsigned long long move0,
                         int walker0, double* __restrict__ mvlog)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ WaveScratch ws;
    __shared__ uint32_t squeue[kQCap * 64];      // in-range queue of the volume move's full-box energy
    __shared__ double shmat[2][9], svol[2];      // the walker's cells: volume moves change them in place
    __shared__ int sniv[2];
    const int lane = threadIdx.x;
    const int wlk = walker0 + blockIdx.x;
    const int L = sp.nlat;
    const int box0 = wlk * L;
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
    // LDSLIST (the reference's own system sizes, ~48 molecules): list rows (32 entries each) and row lengths too,
    // so that nothing in the move loop waits on global memory.  Rows longer than 32 keep the global list.
    uint32_t* srow = reinterpret_cast<uint32_t*>(spos + (LDSPOS ? (size_t)L * N * 3 : 0));
    int* snn = reinterpret_cast<int*>(srow + (LDSLIST ? (size_t)L * N * 32 : 0));
    if (LDSLIST) {
        for (int l = 0; l < L; ++l) {
            const uint32_t* LMg = listm + (size_t)(box0 + l) * N * kRow;
            for (int t = lane; t < N * 32; t += 64) srow[(size_t)l * N * 32 + t] = LMg[(size_t)(t >> 5) * kRow + (t & 31)];
            for (int t = lane; t < N; t += 64) snn[l * N + t] = nn[(size_t)(box0 + l) * N + t];
        }
    }
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
    vc.nivect_g = nivect + box0; vc.list_g = list + (size_t)box0 * S * N; vc.nn_g = nn + (size_t)box0 * N;
    vc.queue = squeue + lane; vc.N = N; vc.S = S; vc.ivcap = ivcap; vc.L = L;
    unsigned long long nvol_try = 0, nvol_acc = 0;
    int flag = 0;

    // this walker's weight table (read by eta_weight, updated by mc_update_wl_bins) and histograms
    double* weight = wweight + (size_t)wlk * sp.nbins;
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
        const bool is_volume = WITHVOL && !(u7 < sp.transP);    // WITHVOL = false: translation-only build, no call, lean registers
        bool ok = false;
        double eo[2] = {0.0, 0.0}, en[2] = {0.0, 0.0}, diffkT = 0.0;
        int imol = 0;
        if (is_volume) {                                                          // mc_moves.F90:232-235
            int rv = 0;
            if constexpr (WITHVOL) rv = volume_move_wave(vc, sp, weight, mu_bin, binwidth, u0, u1, u2, u3, ls, ls_mu, men, lane);
            ++nvol_try;
            if (rv == 1) ++nvol_acc;
            if (rv < 0) flag = 1;
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
            const uint32_t* SR = srow + (size_t)l * N * 32;
            const int* SN = snn + l * N;
            auto row = [&](int jx, int sl) { return LDSLIST ? SR[jx * 32 + sl] : LM[(size_t)jx * kRow + sl]; };
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
        if (L == 1) {
            diffkT = sp.beta * dE0;                                               // :1106
        } else {
            const double eta_old = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);   // :1112-1116
            ls_mu = ls_mu + (dE0 - dE1) * sp.beta;
            const double eta_new = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);
            diffkT = (ls == 1 ? dE0 : dE1) * sp.beta + eta_new - eta_old;
        }
        double pacc = exp(-diffkT);
        pacc = pacc > 1.0 ? 1.0 : pacc;
        ok = u5 < pacc;                                                           // :1145-1146 (false for NaN)
        if (ok) {
            ++acc;
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
                    // weight(k) += av_binwidth*wl_factor/binwidth(k); then subtract the minimum over the window (:1680-1685)
                    double mn = 1.7976931348623157e308;
                    for (int b = sp.start_bin - 1 + lane; b < sp.end_bin; b += 64) {
                        double w = weight[b];
                        if (b == k - 1) w = w + sp.av_binwidth * sp.wl_factor / bwk;
                        mn = w < mn ? w : mn;
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mn, off, 64); mn = o < mn ? o : mn; }
                    for (int b = sp.start_bin - 1 + lane; b < sp.end_bin; b += 64) {
                        double w = weight[b];
                        if (b == k - 1) w = w + sp.av_binwidth * sp.wl_factor / bwk;
                        weight[b] = w - mn;
                    }
                    gauge += mn;
                    if (lane == 0) hist[k - 1] = hist[k - 1] + sp.av_binwidth / bwk;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
        if (L == 2 && sp.always_switch) {                                         // mc_lattice_switch, :1536-1594
            const int lsw = 3 - ls;
            const double eta_w = dev_eta_weight(sp, weight, mu_bin, binwidth, ls_mu);
            const double deta = eta_w - eta_w;                                    // new_eta - old_eta, :1557-1558
            const double Els = ls == 1 ? men[0] : men[1], Elsn = ls == 1 ? men[1] : men[0];
            const double V1 = svol[0], V2 = svol[1];
            const double Vls = ls == 1 ? V1 : V2, Vlsn = ls == 1 ? V2 : V1;
            double dk;
            if (sp.npt) dk = sp.beta * Elsn - sp.beta * Els + sp.beta * sp.pressure * (Vlsn - Vls) - (double)N * log(Vlsn / Vls) + deta;
            else        dk = sp.beta * Elsn - sp.beta * Els + deta;
            double cmp = exp(-dk);
            cmp = cmp > 1.0 ? 1.0 : cmp;
            if (u6 < cmp) {
                double mu = (men[0] + sp.pressure * V1) - (men[1] + sp.pressure * V2);          // :1581-1583
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
    if (lane == 0) {
        wls[wlk] = ls; wmu[wlk] = ls_mu; wacc[wlk] += acc; wswitch[wlk] += nsw; wshift[wlk] += gauge;
        wvol[2 * wlk] += nvol_try; wvol[2 * wlk + 1] += nvol_acc;
        if (flag) wflag[wlk] = 1;
        energy[box0] = men[0];
        if (L == 2) energy[box0 + 1] = men[1];
    }
}

// Single request with by-value overrides (the drop-in compute_local_real_energy call):
// one wave, result written straight to host-visible memory.
__global__ __launch_bounds__(64)
void k_local_energy_single(double* __restrict__ pos, const double* __restrict__ ivect,
                           const uint32_t* __restrict__ listm, const int* __restrict__ nn,
                           int b, int i, Override o1, Override o2, int commit,
                           double* __restrict__ e_out, int N, int ivcap)
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
    }
}

}  // namespace mw
