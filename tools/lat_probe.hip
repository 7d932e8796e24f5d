// lat_probe.hip -- what ONE wavefront pays per instruction on gfx950 (the latency model behind the single-chain work of round 4):
// dependent and independent FP64 chains, cross-lane reads, LDS round trips, ballots, the cycle counter itself.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lat_probe tools/lat_probe.hip && tools/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP 256
// the counter read is tied to the value under test on both sides: the chain cannot move across it
#define TICK(t) do { asm volatile("" : "+v"(a) :: "memory"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); \
                     asm volatile("" : "+v"(a) :: "memory"); } while (0)
__device__ __forceinline__ double rl(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__global__ __launch_bounds__(64) void k_probe(double* out, unsigned long long* cyc, double seed)
{
    __shared__ double lds[1024];
    __shared__ int chase[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) { lds[i] = seed + i; chase[i] = (i * 37 + 11) & 1023; }
    __syncthreads();
    double a = seed + lane, b = seed * 0.5, c = 1.0 - seed;
    unsigned long long t0, t1;
    int n = 0;
    // 0: the counter itself
    unsigned long long tacc = 0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(a)); { unsigned long long tt; TICK(tt); tacc ^= tt; } }
    TICK(t1); if (lane == 0) { cyc[n] = (t1 - t0) / 17; cyc[30] = tacc; } ++n;
    // 1: dependent v_fma_f64
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP; ++i) a = __builtin_fma(a, b, c);
    TICK(t1); if (lane == 0) cyc[n] = t1 - t0; ++n;
    // 2: two independent chains
    double a2 = a + 1.0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 2; ++i) { a = __builtin_fma(a, b, c); a2 = __builtin_fma(a2, b, c); }
    TICK(t1); if (lane == 0) cyc[n] = t1 - t0; ++n;
    a += a2;
    // 3: four independent chains
    double a3 = a + 2.0, a4 = a + 3.0; a2 = a + 1.0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { a = __builtin_fma(a, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); a4 = __builtin_fma(a4, b, c); }
    TICK(t1); if (lane == 0) cyc[n] = t1 - t0; ++n;
    a += a2 + a3 + a4;
    // 4: dependent v_add_f32
    float f = (float)a;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP; ++i) { f = f * 1.0001f + 0.5f; }
    TICK(t1); if (lane == 0) cyc[n] = t1 - t0; ++n;
    a += f;
    // 5: fma -> readlane (uniform) -> fma with the scalar
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { const double s = rl(a, 3); a = __builtin_fma(a, b, s); }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 6: dependent LDS round trips (pointer chase, b32)
    int p = lane;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) p = chase[p];
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    a += p;
    // 7: dependent LDS b64 read + add
    p &= 1023;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { a += lds[p]; p = (p + (int)a) & 1023; }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 8: ds_bpermute chain
    int q = lane;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) q = __builtin_amdgcn_ds_bpermute(((q + 1) & 63) << 2, q);
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    a += q;
    // 9: DPP chain (row_shr:1 add, f64 = two movs + add)
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x111, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x111, 0xf, 0xf, true);
        a += __hiloint2double(hi, lo);
    }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 10: compare -> ballot -> scalar branch (uniform), loop-carried
    int cnt = 0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { const unsigned long long m = __ballot(a > (double)i); if (m & 2ull) cnt += 1; else a += 1.0; }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    a += cnt;
    // 11: v_rcp_f64 chain
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) a = __builtin_amdgcn_rcp(a) + 1.5;
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 12: lane-0-only LDS write followed by an all-lane read of it (exec mask switch + in-order LDS)
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { if (lane == 0) lds[5] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); a += lds[5]; }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 13: the same with workgroup-scope fences (s_waitcnt vmcnt(0) lgkmcnt(0))
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { if (lane == 0) lds[5] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); a += lds[5]; }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 14: independent integer VALU + SALU mix (s_add, v_add alternating, no dependence between them)
    int iv = lane, sv = (int)seed;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 2; ++i) { iv = iv * 3 + 1; sv = __builtin_amdgcn_readfirstlane(sv * 5 + 7); }
    TICK(t1); if (lane == 0) cyc[n] = t1 - t0; ++n;
    a += iv + sv;
    // 15: compare -> select, all on the vector unit (no scalar round trip)
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { a = a > (double)i ? a * 0.5 : a + 1.0; }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 16: compare -> wave mask -> scalar select -> vector use, no branch
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { const unsigned long long m = __ballot(a > (double)i); const double k = (m & 2ull) ? 0.5 : 1.5; a = a * k + 1.0; }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 17: scalar-only loop-carried branch (condition from the scalar unit), vector work on both sides
    int sc = __builtin_amdgcn_readfirstlane((int)seed + 3);
    TICK(t0);
#pragma unroll 1
    for (int i = 0; i < REP; ++i) { sc = sc * 5 + 1; if (sc & 8) a += 1.0; else a *= 0.5; }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0); ++n;
    // 18: exec-masked single-lane work (lane 0 only) without a fence, then everybody continues
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { if (lane == 0) a += 1.0; a *= 0.999; }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    // 19: f64 readlane with a constant lane, result used by the scalar unit only (no vector use)
    int sacc = 0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { a = a * 1.0001; sacc += __builtin_amdgcn_readlane(__double2loint(a), 5); }
    TICK(t1); if (lane == 0) cyc[n] = (t1 - t0) * 4; ++n;
    a += sacc;
    out[lane] = a;
    if (lane == 0) cyc[31] = n;
}

int main()
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 32 * 8);
    hipMemset(cyc, 0, 32 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, cyc, 0.999);
    hipDeviceSynchronize();
    unsigned long long h[32];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"clock64 itself (cycles per stamp)", "dependent v_fma_f64", "2 independent f64 chains", "4 independent f64 chains", "dependent f32 mul-add",
                           "f64 fma -> readlane -> fma(sgpr)", "dependent LDS b32 pointer chase", "dependent LDS b64 read + add + index", "ds_bpermute chain",
                           "DPP row_shr f64 add", "cmp -> ballot -> uniform branch", "v_rcp_f64 + add chain", "lane-0 LDS write, wavefront fence, all-lane read",
                           "same with workgroup-scope fences", "independent VALU int + SALU mix (per pair)", "cmp -> v_cndmask select (vector only)",
                           "cmp -> ballot -> scalar select -> vector use", "scalar-condition branch, vector work both sides", "lane-0-only add then all-lane mul",
                           "mul -> readlane_b32 -> scalar add"};
    for (int i = 0; i < (int)h[31]; ++i)
        printf("%-52s %8.2f cycles per step\n", names[i], i == 0 ? (double)h[i] : (double)h[i] / (i == 14 ? REP / 2 : REP));
    return 0;
}
