// kbench.hip -- side-by-side timing of kernel variants on one MI355X (development tool, not product code).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -o tools/kbench tools/kbench.hip
//   python tools/dump_workload.py /tmp/w.bin && tools/kbench /tmp/w.bin [rounds] [only-variant]
//
// The whole engine is compiled into this program (the include below), so the variants run on the engine's own
// device state: lists built by the product builder, energies checked against the product kernel.
#include "../mc_water_ls_mw_amd/csrc/mw_api.hip"

#include <algorithm>
#include <chrono>
#include <functional>
#include <map>
#include <thread>

#define CK(x) do { if ((x)) { fprintf(stderr, "FAIL %s: %s\n", #x, mw_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Variant { std::string name; std::function<void()> launch; std::vector<float> us; };

template <int LAYOUT, bool BATCH4>
static void launch_me(int nboxes)
{
    const Geo ge = model_geo(nboxes);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_model_energy<true, 1024, LAYOUT, BATCH4>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget); attr = true; }
    hipLaunchKernelGGL((mw::k_model_energy<true, 1024, LAYOUT, BATCH4>), dim3(ge.nsplit, ge.nsplit == 1 ? std::min(nboxes, g.cu) : nboxes), dim3(1024), ge.shmem, g.stream, g.d_pos, g.d_ivect,
                       g.d_nivect, g.d_list, g.d_order, g.d_nns, g.d_cmax, g.d_partial, g.d_cpartial, g.d_energy, g.d_counts, g.N, g.S, g.ivcap, 0, ge.nsplit, ge.chunk, nboxes);
}

template <int LAYOUT>
static void launch_mv()
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_move_energy<true, LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget); attr = true; }
    const size_t iv_bytes = kMoveScratch + mw::lds_vec_bytes((size_t)g.ivcap);
    hipLaunchKernelGGL((mw::k_move_energy<true, LAYOUT>), dim3(g.mwork_n), dim3(1024),
                       iv_bytes + mw::lds_vec_bytes((size_t)g.N) + (((size_t)g.N + 7) & ~(size_t)7) + (size_t)g.mchunk * sizeof(int), g.stream,
                       g.d_pos, g.d_ivect, g.d_nivect, g.d_listm, g.d_nn, g.d_mwork, g.d_mimol, g.d_mtrial, g.d_mperm,
                       g.d_meold, g.d_menew, g.d_mcnt, g.d_mdecl, g.N, g.ivcap, 3);
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: kbench workload.bin [rounds] [only]\n"); return 2; }
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const char* only = argc > 3 ? argv[3] : nullptr;
    FILE* fh = fopen(argv[1], "rb");
    if (!fh) { perror(argv[1]); return 2; }
    int hdr[3];
    double h[9];
    if (fread(hdr, 4, 3, fh) != 3 || fread(h, 8, 9, fh) != 9) return 2;
    const int W = hdr[0], N = hdr[1], M = hdr[2];
    std::vector<double> pos((size_t)W * N * 3);
    if (fread(pos.data(), 8, pos.size(), fh) != pos.size()) return 2;
    std::vector<int> imol((size_t)W * M);
    std::vector<double> trial((size_t)W * M * 3);
    if (fread(imol.data(), 4, imol.size(), fh) != imol.size() || fread(trial.data(), 8, trial.size(), fh) != trial.size()) return 2;
    fclose(fh);
    printf("workload: %d walkers x %d molecules, %d moves/walker\n", W, N, M);

    CK(mw_init(0, N, W, 50));
    for (int b = 1; b <= W; ++b) { int niv; CK(mw_set_cell(b, h, &niv)); }
    CK(mw_upload_positions_range(1, W, pos.data()));
    int mn, mx;
    CK(mw_build_neighbours_batch(1, W, &mn, &mx));
    printf("lists: nn %d..%d   order_seg %d kbits %d\n", mn, mx, g.order_seg, g.order_kbits);
    {
        hipEvent_t a, b;
        HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
        HK(hipEventRecord(a, g.stream));
        for (int k = 0; k < 3; ++k) CK(mw_build_neighbours_launch(1, W));
        HK(hipEventRecord(b, g.stream));
        HK(hipEventSynchronize(b));
        float ms = 0.f;
        HK(hipEventElapsedTime(&ms, a, b));
        printf("time  list rebuild (all kernels)                     %8.1f us per build of %d boxes\n", ms * 1e3f / 3, W);
    }
    std::vector<double> eref(W);
    CK(mw_model_energy_batch(1, W, eref.data()));
    long long np_ref, nt_ref;
    CK(mw_model_energy_counts_total(1, W, &np_ref, &nt_ref));

    std::vector<Variant> vs;
    vs.push_back({"product", [&] { (void)launch_model_energy(1, W); }, {}});
    if (!lds_fits(g.N, g.ivcap)) {
        auto big = [&](bool batch) {
            const Geo ge = model_geo(W);
            if (batch) hipLaunchKernelGGL((mw::k_model_energy<false, 256, mw::kLayoutPair, true>), dim3(ge.nsplit, W), dim3(256), ge.shmem, g.stream, g.d_pos, g.d_ivect,
                       g.d_nivect, g.d_list, g.d_order, g.d_nns, g.d_cmax, g.d_partial, g.d_cpartial, g.d_energy, g.d_counts, g.N, g.S, g.ivcap, 0, ge.nsplit, ge.chunk, W);
            else hipLaunchKernelGGL((mw::k_model_energy<false, 256, mw::kLayoutPair, false>), dim3(ge.nsplit, W), dim3(256), ge.shmem, g.stream, g.d_pos, g.d_ivect,
                       g.d_nivect, g.d_list, g.d_order, g.d_nns, g.d_cmax, g.d_partial, g.d_cpartial, g.d_energy, g.d_counts, g.N, g.S, g.ivcap, 0, ge.nsplit, ge.chunk, W);
        };
        vs.push_back({"k_model_energy<global, slot at a time>", [=] { big(false); }, {}});
        vs.push_back({"k_model_energy<global, BATCH4>", [=] { big(true); }, {}});
    }
    if (lds_fits(g.N, g.ivcap)) {
    vs.push_back({"k_model_energy<AoS>", [&] { launch_me<mw::kLayoutAoS, false>(W); }, {}});
    vs.push_back({"k_model_energy<Pair>", [&] { launch_me<mw::kLayoutPair, false>(W); }, {}});
    vs.push_back({"k_model_energy<Pair,BATCH4>", [&] { launch_me<mw::kLayoutPair, true>(W); }, {}});
    vs.push_back({"k_model_energy<SoA>", [&] { launch_me<mw::kLayoutSoA, false>(W); }, {}});
    vs.push_back({"k_model_energy<SoA,BATCH4>", [&] { launch_me<mw::kLayoutSoA, true>(W); }, {}});
    }

    // ---- the drop-in single call: host wall time per synchronous mw_local_energy_patched through the C ABI ------------
    if (!only || std::string(only) == "latency") {
        std::vector<double> p1(pos.begin(), pos.begin() + (size_t)N * 3);
        auto run = [&](int ncalls) {
            double acc = 0.0, e = 0.0;
            int prev = 0;
            unsigned s = 12345u;
            for (int k = 0; k < ncalls; ++k) {
                s = s * 1664525u + 1013904223u;
                const int im = (int)(s % (unsigned)N) + 1;
                if (mw_local_energy_patched(1, im, &p1[3 * (size_t)(im - 1)], prev, prev ? &p1[3 * (size_t)(prev - 1)] : nullptr, &e)) { fprintf(stderr, "%s\n", mw_last_error()); break; }
                acc += e; prev = im;
            }
            return acc;
        };
        run(2000);
        {   // device-side stamps of one call each (100 MHz clock): poll -> decoded, evaluation
            double tp = 0.0, te = 0.0; double e;
            for (int k = 0; k < 1000; ++k) {
                const int im = 1 + (k * 37) % N;
                mw_local_energy_patched(1, im, &p1[3 * (size_t)(im - 1)], 0, nullptr, &e);
                tp += (double)g.h_slots[0].pad_c[0] * 0.01; te += (double)g.h_slots[0].pad_c[1] * 0.01;
            }
            printf("time  server: poll read %.2f us, evaluation %.2f us (device clock, mean of 1000)\n", tp / 1000, te / 1000);
#ifdef MW_LAT_STAMPS      // (build with -DMW_LAT_STAMPS: the stamps themselves cost a few tenths of a microsecond)
            double acc8[8] = {0};
            for (int k = 0; k < 200; ++k) {
                const int im = 1 + (k * 53) % N;
                mw_local_energy_patched(1, im, &p1[3 * (size_t)(im - 1)], 0, nullptr, &e);
                unsigned long long st[16];
                (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(mw::g_lat_stamps), sizeof st);
                for (int q = 1; q < 8; ++q) acc8[q] += (double)(st[q] - st[q - 1]) * 0.01;
            }
            printf("time  inside one evaluation (us): own row %.2f | compaction %.2f | row fetch issue %.2f | j-i-k pairs %.2f | scan %.2f | last flush %.2f | sums %.2f\n",
                   acc8[1] / 200, acc8[2] / 200, acc8[3] / 200, acc8[4] / 200, acc8[5] / 200, acc8[6] / 200, acc8[7] / 200);
#endif
        }
        const auto t0 = std::chrono::steady_clock::now();
        const double acc = run(20000);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20000.0;
        printf("time  mw_local_energy_patched, one call (server %s)   %8.2f us   (sum %.6f)\n", g.srv_enabled ? "resident" : "off: one launch per call", us, acc);
        if (g.srv_enabled && W >= 2) {   // two host threads, one lattice each (the reference's dormant OpenMP sections): each has its own slot
            std::vector<double> p2(pos.begin() + (size_t)N * 3, pos.begin() + (size_t)N * 6);
            auto one = [&](int box, const std::vector<double>& pp, int ncalls, double* out) {
                double a = 0.0, e = 0.0; int prev = 0; unsigned s = 777u + box;
                for (int k = 0; k < ncalls; ++k) {
                    s = s * 1664525u + 1013904223u;
                    const int im = (int)(s % (unsigned)N) + 1;
                    if (mw_local_energy_patched(box, im, &pp[3 * (size_t)(im - 1)], prev, prev ? &pp[3 * (size_t)(prev - 1)] : nullptr, &e)) break;
                    a += e; prev = im;
                }
                *out = a;
            };
            double a1 = 0.0, a2 = 0.0;
            const auto tt = std::chrono::steady_clock::now();
            std::thread th(one, 2, std::cref(p2), 20000, &a2);
            one(1, p1, 20000, &a1);
            th.join();
            printf("time  two threads, one lattice each, per PAIR of calls  %8.2f us   (sums %.6f %.6f)\n",
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tt).count() / 20000.0, a1, a2);
        }
        double em = 0.0;
        const auto t1 = std::chrono::steady_clock::now();
        for (int k = 0; k < 200; ++k) CK(mw_model_energy(1, &em));
        printf("time  mw_model_energy, one box, one call               %8.2f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / 200.0);
        // alternating: the server is stopped by every exclusive entry point and restarted by the next single call
        const auto t2 = std::chrono::steady_clock::now();
        for (int k = 0; k < 200; ++k) { double e; CK(mw_local_energy_patched(1, 1 + k, &p1[3 * (size_t)k], 0, nullptr, &e)); CK(mw_sync()); }
        printf("time  single call + mw_sync (server stop/start)         %8.2f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t2).count() / 200.0);
    }

    // ---- single-move kernel: layouts of the staged vectors ------------------------------------------------------
    std::vector<Variant> mv;
    if (M > 0 && lds_fits_move(g.N, g.ivcap)) {
        std::vector<int> ils((size_t)W * M);
        for (int w = 0; w < W; ++w) for (int m = 0; m < M; ++m) ils[(size_t)w * M + m] = w + 1;
        CK(mw_moves_upload(W * M, ils.data(), imol.data(), trial.data()));
        if (g.mlds) {
            CK(mw_moves_launch());
            std::vector<double> eo((size_t)W * M), en((size_t)W * M), eo2((size_t)W * M), en2((size_t)W * M);
            CK(mw_moves_fetch(eo.data(), en.data()));
            mv.push_back({"k_move_energy product", [&] { (void)launch_moves(3); }, {}});
            mv.push_back({"k_move_energy<AoS>", [&] { launch_mv<mw::kLayoutAoS>(); }, {}});
            mv.push_back({"k_move_energy<Pair>", [&] { launch_mv<mw::kLayoutPair>(); }, {}});
            mv.push_back({"k_move_energy<SoA>", [&] { launch_mv<mw::kLayoutSoA>(); }, {}});
            for (auto& v : mv) {
                if (only && v.name.find(only) == std::string::npos) continue;
                v.launch();
                HK(hipGetLastError());
                CK(mw_moves_fetch(eo2.data(), en2.data()));
                double worst = 0.0;
                for (size_t k = 0; k < eo.size(); ++k) worst = std::max(worst, std::max(fabs(eo2[k] - eo[k]), fabs(en2[k] - en[k])));
                printf("check %-44s max |diff| %.2e Ha\n", v.name.c_str(), worst);
            }
        }
    }

    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    // correctness of every variant first (energies of all boxes against the product kernel)
    for (auto& v : vs) {
        if (only && v.name.find(only) == std::string::npos) continue;
        HK(hipMemsetAsync(g.d_energy, 0, sizeof(double) * W, g.stream));
        v.launch();
        HK(hipGetLastError());
        std::vector<double> e(W);
        std::vector<unsigned long long> c((size_t)2 * W);
        HK(hipMemcpyAsync(e.data(), g.d_energy, sizeof(double) * W, hipMemcpyDeviceToHost, g.stream));
        HK(hipMemcpyAsync(c.data(), g.d_counts, sizeof(unsigned long long) * 2 * W, hipMemcpyDeviceToHost, g.stream));
        HK(hipStreamSynchronize(g.stream));
        double worst = 0.0;
        long long np = 0, nt = 0;
        const Geo gchk = model_geo(W);
        for (int b = 0; b < W; ++b) { worst = std::max(worst, fabs(e[b] - eref[b]) / fabs(eref[b])); np += (long long)c[2 * b]; nt += (long long)c[2 * b + 1]; }
        printf("check %-44s max rel diff %.2e  counts %s\n", v.name.c_str(), worst, (np == np_ref && nt == nt_ref) ? "equal" : "DIFFER");
    }
    // timing: interleaved rounds, 5 launches per variant per round
    for (int r = 0; r < rounds; ++r) {
        for (auto& v : vs) {
            if (only && v.name.find(only) == std::string::npos) continue;
            for (int k = 0; k < 5; ++k) {
                HK(hipEventRecord(e0, g.stream));
                v.launch();
                HK(hipEventRecord(e1, g.stream));
                HK(hipEventSynchronize(e1));
                float ms = 0.f;
                HK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0 || k > 1) v.us.push_back(ms * 1e3f);
            }
        }
    }
    for (int r = 0; r < rounds; ++r) {
        for (auto& v : mv) {
            if (only && v.name.find(only) == std::string::npos) continue;
            for (int k = 0; k < 3; ++k) {
                HK(hipEventRecord(e0, g.stream));
                v.launch();
                HK(hipEventRecord(e1, g.stream));
                HK(hipEventSynchronize(e1));
                float ms = 0.f;
                HK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0 || k > 0) v.us.push_back(ms * 1e3f);
            }
        }
    }
    for (auto* grp : {&vs, &mv})
    for (auto& v : *grp) {
        if (v.us.empty()) continue;
        std::sort(v.us.begin(), v.us.end());
        printf("time  %-44s median %8.1f us   min %8.1f us   (n=%zu)\n", v.name.c_str(), v.us[v.us.size() / 2], v.us[0], v.us.size());
    }
    CK(mw_finalize());
    return 0;
}
