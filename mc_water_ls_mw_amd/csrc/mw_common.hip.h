// mw_common.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine:
// model constants, the packed list entry, double-precision primitives (rsqrt / reciprocal / exp sized
// for the 1e-10 parity bar), wave and DPP reductions.
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

// exp(x) for -800 <= x <= 0.  n = round(x log2 e), r = x - n ln2 in two pieces, degree-10 minimax polynomial
// on |r| <= ln2/2 (relative error 4.5e-16 over the interval, evaluated in double: tighter than the degree-11
// Taylor sum, one multiply-add fewer), 2^n by ldexp.
__device__ __forceinline__ double fast_exp_neg_bounded(double x)
{
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);    // ln2 high part (fdlibm split)
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);           // ln2 low part
    double p = 2.76263932715658755431e-07;
    p = fma_sc(p, r, 2.76401815000467573734e-06);
    p = fma_sc(p, r, 2.48015042451010998305e-05);
    p = fma_sc(p, r, 1.98411702687116374552e-04);
    p = fma_sc(p, r, 1.38888889325820222981e-03);
    p = fma_sc(p, r, 8.33333338566921363877e-03);
    p = fma_sc(p, r, 4.16666666665728713248e-02);
    p = fma_sc(p, r, 1.66666666665543999892e-01);
    p = fma_sc(p, r, 5.00000000000000555112e-01);
    p = fma_sc(p, r, 1.00000000000000666134e+00);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);
}
// exp(x) for x <= 0 (any magnitude; underflows smoothly to 0)
__device__ __forceinline__ double fast_exp_neg(double x)
{
    return fast_exp_neg_bounded(__builtin_fmax(x, -800.0));         // exp(-800) == 0 in double anyway
}

// exp(x) for any x (the Monte Carlo driver's acceptance tests: arguments of either sign and any size): the same reduction
// and polynomial as above -- 4.5e-16 relative -- with the result's exponent applied by ldexp, which overflows to +inf and
// underflows through the denormals to 0 as exp itself does; NaN stays NaN.  (The library exp costs the driver ~20 vector
// registers at its call site -- the difference between three and four wavefronts per SIMD.)
__device__ __forceinline__ double exp_any(double x)
{
    const double xc = __builtin_fmin(__builtin_fmax(x, -746.0), 710.0);
    const double r = fast_exp_neg_bounded(xc);
    return x != x ? x : r;
}

// log(x) for finite x > 0 (NaN otherwise), ~1.5 ulp, 30 instructions where the library's double-double log takes 150 (the Monte Carlo
// driver's bin search, a scalar computation a wavefront has to run on its vector unit): x = m 2^e with m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s) = 2 s (1 + z/3 + z^2/5 + ... + z^10/21), s = (m - 1)/(m + 1), z = s^2 <= 0.0295 (the series' next term < 1e-18).
__device__ __forceinline__ double fast_log_pos(double x)
{
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);                     // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double s = (m - 1.0) * fast_rcp(m + 1.0);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = fma_sc(p, z, 1.0 / 19.0);
    p = fma_sc(p, z, 1.0 / 17.0);
    p = fma_sc(p, z, 1.0 / 15.0);
    p = fma_sc(p, z, 1.0 / 13.0);
    p = fma_sc(p, z, 1.0 / 11.0);
    p = fma_sc(p, z, 1.0 / 9.0);
    p = fma_sc(p, z, 1.0 / 7.0);
    p = fma_sc(p, z, 1.0 / 5.0);
    p = fma_sc(p, z, 1.0 / 3.0);
    p = __builtin_fma(p, z, 1.0);
    const double ef = (double)e;
    const double r = __builtin_fma(ef, 6.93147180369123816490e-01, __builtin_fma(ef, 1.90821492927058770002e-10, (s + s) * p));
    return x > 0.0 ? r : __builtin_nan("");
}

// The two exponentials of an in-range pair from one: with t = exp(0.2 sigma/(r - a sigma)),
// exp(sigma/(r - a sigma)) = t^5 (molint.F90:291,459) and g = exp(gamma sigma/(r - a sigma)) = t^6 (:292,462).
// r2 < rc^2 but r rounded onto (or within 1.2e-3 bohr of) rc: both are exactly 0 in double (t < 1e-300), which
// the clamp reproduces while keeping the exponent inside fast_exp_neg_bounded's range.
constexpr double kDenClamp = -1.2e-3;
__device__ __forceinline__ void pair_terms(double r2, double& rinv, double& e1, double& g)
{
    rinv = fast_rsqrt(r2);                       // molint.F90:278
    const double den = fma_sc(r2, rinv, -kSigA); // r - a sigma, r = r2 / sqrt(r2)        :286
    const double w = fast_rcp(__builtin_fmin(den, kDenClamp));
    const double t = fast_exp_neg_bounded(0.2 * kSigma * w);
    const double t2 = t * t, t4 = t2 * t2;
    e1 = t4 * t;
    g  = t4 * t2;
}

// Per-molecule MOMENTS of the in-range neighbourhood, the by-product of the full-box pass that the single-move kernel's moment
// path consumes (mw_move_energy.hip.h, move_energy_mom_wave): with g_k = exp(gamma sigma/(r_jk - a sigma)) and u_k the unit vector
// from j to its in-range neighbour k,
//   [0] S0 = sum g_k   [1..3] S1 = sum g_k u_k   [4..8] S2 = sum g_k u_k u_k^T (xx, yy, xy, xz, yz; zz = S0 - xx - yy: the u_k are unit
//   vectors)   [9] the number of in-range neighbours
// -- what the i--j--k triplet sum of a molecule i next to j needs of j's other neighbours:
//   sum_k g_k (u_i . u_k - c0)^2 = u_i^T S2 u_i - 2 c0 u_i . S1 + c0^2 S0.
constexpr int kMomStride = 10;   // doubles per molecule (80 bytes: five 16-byte stores / loads -- the full-box pass that writes them is bound by those bytes)

// ---- staged vectors in LDS ------------------------------------------------------------
// LDS layout of the staged positions and image vectors (what a random gather costs the LDS: MI355X_MICROARCH.md,
// LDS table -- the plain [N][3] layout makes the compiler fuse x,y into ds_read2_b64, which runs at HALF the rate of
// ds_read_b64 / ds_read_b128: 10 LDS cycles per gathered position instead of 6):
//   kLayoutAoS  : r[j][3]                      (ds_read2_b64 + ds_read_b64)
//   kLayoutPair : xy[j] 16-byte pairs, z[j]    (ds_read_b128 + ds_read_b64)
//   kLayoutSoA  : x[j], y[j], z[j]             (3 x ds_read_b64)
constexpr int kLayoutAoS = 0, kLayoutPair = 1, kLayoutSoA = 2;

__host__ __device__ constexpr size_t lds_vec_bytes(size_t n) { return (n * 24 + 15) & ~(size_t)15; }   // n vectors, 16-byte granules

template <int LAYOUT>
struct LdsVecs {
    const double* base;
    int n;
    __device__ __forceinline__ void get(int j, double& x, double& y, double& z) const
    {
        if constexpr (LAYOUT == kLayoutAoS) {
            const double* p = base + 3 * (size_t)j; x = p[0]; y = p[1]; z = p[2];
        } else if constexpr (LAYOUT == kLayoutPair) {
            const double2 xy = reinterpret_cast<const double2*>(base)[j]; x = xy.x; y = xy.y; z = base[2 * (size_t)n + j];
        } else {
            x = base[j]; y = base[(size_t)n + j]; z = base[2 * (size_t)n + j];
        }
    }
    // element t of the flat [n][3] source goes here
    __device__ __forceinline__ static size_t slot(int t, int n)
    {
        if constexpr (LAYOUT == kLayoutAoS) return (size_t)t;
        const int j = t / 3, c = t - 3 * j;
        if constexpr (LAYOUT == kLayoutPair) return c < 2 ? 2 * (size_t)j + c : 2 * (size_t)n + j;
        else return (size_t)c * n + j;
    }
};

// Stage the flat [n][3] array `src` into LDS in layout LAYOUT with THREADS threads.  The loads of a batch are ALL
// issued before the first store: the plain loop `dst[slot(t)] = src[t]` compiles to load, wait, store per iteration
// -- twelve HBM round trips in a row for a 4096-molecule box (~10 us of a workgroup that lives ~48 us).
template <int LAYOUT, int THREADS>
__device__ __forceinline__ void stage_vecs(double* __restrict__ dst, const double* __restrict__ src, int count, int n, int tid)
{
    constexpr int kBatch = 12;                           // (`count` vectors are copied; `n` is the layout's capacity)
    const int total = 3 * count;
    if (total <= 0) return;
    for (int base = 0; base < total; base += kBatch * THREADS) {
        double v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) { const int t = base + tid + k * THREADS; v[k] = src[t < total ? t : total - 1]; }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) { const int t = base + tid + k * THREADS; if (t < total) dst[LdsVecs<LAYOUT>::slot(t, n)] = v[k]; }
    }
}

// The image vectors of a box (a few dozen, 3 x niv doubles): the first THREADS elements are requested with `stage_iv_begin`
// BEFORE the positions are staged and stored with `stage_iv_end` after, so their round trip overlaps the box's.
template <int THREADS>
__device__ __forceinline__ double stage_iv_begin(const double* __restrict__ IV, int niv, int tid)
{
    return niv > 0 ? IV[tid < 3 * niv ? tid : 0] : 0.0;
}
template <int LAYOUT, int THREADS>
__device__ __forceinline__ void stage_iv_end(double* __restrict__ siv, const double* __restrict__ IV, int niv, int ivcap, int tid, double first)
{
    if (tid < 3 * niv) siv[LdsVecs<LAYOUT>::slot(tid, ivcap)] = first;
    for (int t = tid + THREADS; t < 3 * niv; t += THREADS) siv[LdsVecs<LAYOUT>::slot(t, ivcap)] = IV[t];   // (more than THREADS / 3 images: small sheared cells)
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
// Two wave sums for the price of one: v_permlane32_swap (gfx950) exchanges the upper 32 lanes of `a` with the lower 32 of
// `b`, so one add folds both sums to 32 lanes each -- a's in lanes 0..31, b's in lanes 32..63 -- and a five-step DPP
// reduction inside the halves finishes both.  Returns the sums through lanes 31 and 63.
__device__ __forceinline__ void dpp_wave_sum2(double a, double b, double& sum_a, double& sum_b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    double v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    v += dpp_mov_f64<0x111, 0xf>(v);      // row_shr:1
    v += dpp_mov_f64<0x112, 0xf>(v);      // row_shr:2
    v += dpp_mov_f64<0x114, 0xf>(v);      // row_shr:4
    v += dpp_mov_f64<0x118, 0xf>(v);      // row_shr:8
    v += dpp_mov_f64<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    sum_a = readlane_f64(v, 31);
    sum_b = readlane_f64(v, 63);
}
// Minimum over the 64 lanes, same network; lanes a step does not reach keep their own value.  Lane 63 ends up with it.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_keep_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_wave_min(double v)
{
    double o;
    o = dpp_keep_f64<0x111, 0xf>(v); v = o < v ? o : v;
    o = dpp_keep_f64<0x112, 0xf>(v); v = o < v ? o : v;
    o = dpp_keep_f64<0x114, 0xf>(v); v = o < v ? o : v;
    o = dpp_keep_f64<0x118, 0xf>(v); v = o < v ? o : v;
    o = dpp_keep_f64<0x142, 0xa>(v); v = o < v ? o : v;
    o = dpp_keep_f64<0x143, 0xc>(v); v = o < v ? o : v;
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

}  // namespace mw
