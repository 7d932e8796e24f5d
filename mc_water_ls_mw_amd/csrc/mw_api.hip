// mw_api.hip -- the C ABI of libmw_hip.so (include/mw_energy.h): context, device
// mirrors of the host's model::ljr / model::hmatrix, launch logic.
//
// Host-side state mirrors what the reference's `module energy` keeps
// (molint.F90:41-45,79-81): nivect/ivect per box, the neighbour list, and the
// energies of the last evaluation.  There is no CPU compute path here: every
// energy and every list comes from the gfx950 kernels in mw_kernels.hip.h.
#include "mw_kernels.hip.h"
#include "../../include/mw_energy.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <mutex>
#include <shared_mutex>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

namespace {

thread_local std::string g_err;
// Every entry point takes this lock: the engine is one context per process, calls from several host threads (the
// reference's dormant OpenMP would evaluate both lattices concurrently, mc_moves.F90:1006-1018) are serialised.
std::recursive_mutex g_mu;
// The single local-energy call (the drop-in compute_local_real_energy) does not take g_mu: it holds g_gate shared and
// its lattice's mail slot, so two host threads can evaluate the two lattices of a move at the same time
// (mc_moves.F90:1006-1018).  Every other entry point holds g_gate exclusively (outermost level only:
// entry points call each other) and first stops the resident server those calls talk to.
std::shared_mutex g_gate;
int g_depth = 0;                        // nesting of exclusive entry points on the thread that holds g_mu
struct DeviceGuard;
struct ExclusiveGuard;
#define MW_LOCK ExclusiveGuard mw_lock_; DeviceGuard mw_dev_; if (mw_lock_.rc) return 1

int fail(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIPCHK(call)                                                                             \
    do {                                                                                         \
        hipError_t err__ = (call);                                                               \
        if (err__ != hipSuccess)                                                                 \
            return fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
    } while (0)

constexpr int kTimerSlots = 4096;
constexpr int kLdsBudget = 160 * 1024 - 2048;   // leave room for the static reduction arrays

struct Ctx {
    bool live = false;
    int device = 0, N = 0, nbox = 0, S = 0, ivcap = 0;
    int cu = 0, nsplit_max = 0;
    hipStream_t stream = nullptr;
    double* d_pos = nullptr;
    double* d_mom = nullptr;           // [box][N][kMomStride]: per-molecule moments (k_model_energy's by-product) for the single-move kernel's moment path
    int mom_first = 0, mom_count = 0;  // the boxes whose moments the LAST full-box launch left valid (cleared by everything that may move a molecule)
    int swm_first = 0, swm_count = 0;  // the boxes (1-based first) whose moments in d_mom the Monte Carlo driver keeps current from launch to launch
                                       // (walkers in global memory): cleared by every entry point that writes positions or cells behind the driver's back
    int m_boxlo = 0, m_boxhi = -1, m_minreq = 0;   // the uploaded requests: their boxes (0-based range) and the fewest requests any of them has
    double* d_ivect = nullptr;
    int* d_nivect = nullptr;
    uint32_t* d_list = nullptr;    // slot-major   [box][S][N]
    uint32_t* d_listm = nullptr;   // molecule-major [box][N][64]
    int* d_nn = nullptr;
    int* d_stats = nullptr;
    // sorted slot-major layout (k_list_order): column t of d_list belongs to molecule d_order[t]
    int* d_order = nullptr;        // [box][N]
    int* d_nns = nullptr;          // [box][N]   row length of column t
    int* d_cmax = nullptr;         // [box][ceil(N/64)] longest row of each group of 64 columns
    unsigned char* d_cin = nullptr;   // [box][N]   neighbours inside the energy cutoff when the list was built
    int order_kbits = -1, order_seg = 0;   // sort-key bits and segment length of k_list_order
    // cell-grid neighbour builder
    mw::GridDesc* d_grid = nullptr;
    int* d_usegrid = nullptr;
    int *d_cellid = nullptr, *d_shift = nullptr, *d_sorted = nullptr;
    float4 *d_wrel = nullptr, *d_wpos = nullptr;   // wrapped cell-relative positions (single precision) by molecule / by cell-sorted slot
    int* d_wsh = nullptr;                          // packed shifts by cell-sorted slot
    bool legacy_search = false;                    // MW_CELL_SEARCH=legacy: the one-thread-per-molecule search (cross-check)
    bool sort_in_lds = false;                      // bin + scan + scatter of a box in one workgroup (k_cell_sort_box); MW_CELL_SORT=global: the three kernels
    int *d_ccount = nullptr, *d_cstart = nullptr, *d_ccursor = nullptr;
    int cstride = 0;
    std::vector<mw::GridDesc> h_grid;
    std::vector<int> h_usegrid;
    std::vector<char> h_listbuilt;     // per box: a neighbour list has been built (the full-box kernel may be run over it)
    int last_sweep[6] = {0, 0, 0, 0, 0, 0};   // what the last launch of the driver was: lattices, look-ahead, residency, volume moves, LDS bytes, row stride
    bool grid_on_device = false;   // some box of this context has (had) a cell grid: descriptors travel with mw_sweep_sync_cells
    bool force_brute = false;
    double* d_partial = nullptr;
    unsigned long long* d_cpartial = nullptr;
    double* d_energy = nullptr;
    unsigned long long* d_counts = nullptr;
    // device-resident translation driver (walker = nlat consecutive boxes)
    double* d_hmat = nullptr;                    // [box][9] hmatrix(:,:,ils), column-major
    bool sweep_ready = false;
    mw::SweepParams sp;
    int nwalkers = 0;
    double *d_sw_mubin = nullptr, *d_sw_binwidth = nullptr;
    double *d_wweight = nullptr, *d_whist = nullptr, *d_wuhist = nullptr;   // [walker][nbins]
    unsigned long long* d_wswitch = nullptr;
    int mchunk = 16;                 // requests per work item of the uploaded batch
    bool m_noself = false;           // every box of the uploaded batch went through the cell grid: no molecule meets an image of itself
    double* d_wshift = nullptr;
    double* d_tabscratch = nullptr;  // [3 nbins last | 3 nbins out | chunks x nbins partial] for mw_sweep_reduce_tables
    size_t tabscratch_n = 0;      // per walker: sum of the minima mc_update_wl_bins subtracted since the last read-out
    unsigned long long* d_wvol = nullptr;        // [walker][2] volume moves attempted / accepted
    int* d_wflag = nullptr;                      // [walker] bit 0: a volume move needed more image vectors than ivcap; bit 1: 'dd' walker outside its window at eq_mc_cycles
    double* d_wwin = nullptr;                    // [walker][4] start_bin, end_bin, mu_lo, mu_hi ('dd' windows); used when has_windows
    double *d_wfac = nullptr, *d_wsum = nullptr; // [walker] Wang-Landau increment, Swetnam's visit total
    int* d_winflag = nullptr;                    // [walker] walker_in_window
    double* d_wmom = nullptr; size_t wmom_cap = 0;   // the driver's moment scratch (doubles), grown on demand
    hipEvent_t ev_srv = nullptr;                     // the server's stream waits on it for the moments made on the main stream
    double* d_pm = nullptr; int* d_srvmomok = nullptr;   // the resident server's moment path: the positions its moments were made from [nbox][N][3]; per box, still in step
    double* d_wstep = nullptr;                   // [walker][2] max_trans, dv_max (bohr) when the walkers' step sizes differ (mw_sweep_steps)
    bool has_steps = false;
    int sweep_log_ahead = 8;                     // look-ahead allowed when the move log is on (tests pin it to compare builds)
    bool has_windows = false;
    double* d_volume = nullptr;                  // [box] |det hmatrix|
    int* d_wls = nullptr;
    double* d_wmu = nullptr;
    unsigned long long* d_wacc = nullptr;
    double* d_swlog = nullptr;
    unsigned long long list_version = 1, nnmax_version = 0;   // lists rebuilt <-> cached max row length
    int nnmax_cached = 0;
    size_t swlog_cap = 0;
    // staged moves
    int mcap = 0, mn = 0;
    int* d_mimol = nullptr;
    double *d_mtrial = nullptr, *d_meold = nullptr, *d_menew = nullptr;
    unsigned int* d_mcnt = nullptr;
    unsigned int* d_mtot = nullptr; int mtot_cap = 0, mtot_n = 0;   // [work item][4]: the counts of the requests the move kernel's moment path served (d_mcnt holds 0 for those)
    int* d_mperm = nullptr;        // sorted request -> caller's index
    int* d_mdecl = nullptr;        // [0], [1] counts (alternate launches), then {request, box} of the requests k_move_energy left to k_move_fallback
    int mdecl_par = 0;             // which count word the next launch uses (the fallback kernel zeroes the other)
    int4* d_mwork = nullptr;       // work items {box, begin, end, 0}
    int mwork_cap = 0, mwork_n = 0;
    bool mlds = false;
    int mmode = 0;
    // pinned, device-visible scratch for single results
    double* h_pin = nullptr;
    char* h_stage = nullptr; char* d_stage = nullptr; size_t stage_bytes = 0;   // pinned + mapped: one box's cell record / positions on their way in
    double* d_pin = nullptr;
    unsigned long long pin_seq = 0;   // completion word of the single-call kernel (h_pin + 8 doubles)
    // resident server of the single local-energy call (k_local_server): mail slots in host-mapped memory
    hipStream_t sstream = nullptr;
    mw::MailHead* h_head = nullptr;  mw::MailHead* d_head = nullptr;
    mw::MailSlot* h_slots = nullptr; mw::MailSlot* d_slots = nullptr;
    mw::MailSlot* req_slots = nullptr;  // where requests are posted: h_slots, or device memory the host writes through the BAR
    mw::MailSlot* d_req = nullptr;      // the same lines as the server kernel addresses them
    void* req_dev_alloc = nullptr;
    int nslots = 0;
    bool srv_running = false, srv_enabled = true;
    unsigned long long sseq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long spend[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // per slot: sequence number of a posted, not yet collected request (0: none)
    unsigned long long spend_epoch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // host mirrors
    std::vector<double> h_ivect;   // nbox * ivcap * 3
    std::vector<int> h_nivect;     // nbox
    hipEvent_t ev[kTimerSlots][2] = {};   // created on first use
};

Ctx g;
std::mutex g_slot_mu[8];                // one per mail slot
std::mutex g_srv_mu;                    // start / stop of the server
std::atomic<unsigned long long> g_epoch{0};   // bumped by every exclusive entry point: a reply posted before, collected after, is stale
std::atomic<bool> g_srv_enabled{true};  // MW_LOCAL_SERVER != 0 (read by the single call before it holds any lock)

int server_stop();                      // defined below (needs the context)

struct ExclusiveGuard {
    int rc = 0;                         // a fault of the resident server surfaces HERE, at the entry point that stopped it
    ExclusiveGuard()
    {
        g_mu.lock();
        if (g_depth++ == 0) {
            g_gate.lock(); g_epoch.fetch_add(1, std::memory_order_relaxed);
            g.mom_count = 0;            // moments of an earlier entry point's full-box pass: positions may have moved since (only a pass
                                        // inside THIS entry point -- mw_step_launch -- makes them valid for its move kernel)
            if (g.srv_running) rc = server_stop();
            // A request posted ahead (mw_local_energy_post) that the server never got to -- it left between the post and this
            // entry point -- is CANCELLED: marked as answered, so that the server started by the next single call does not
            // replay it and commit its stale override positions over what this entry point is about to upload.  (Its collect
            // returns 2, "ask again", because of the epoch.)  The server is stopped: nobody else writes the reply lines.
            if (g.live && g.h_slots)
                for (int sl = 0; sl < g.nslots && sl < 8; ++sl)
                    if (g.spend[sl] && reinterpret_cast<volatile unsigned long long*>(&g.h_slots[sl].rep_seq)[0] != g.spend[sl]) {
                        reinterpret_cast<volatile unsigned long long*>(&g.h_slots[sl].rep_seq)[0] = g.spend[sl];
                        std::atomic_thread_fence(std::memory_order_seq_cst);
                    }
        }
    }
    ~ExclusiveGuard()
    {
        if (--g_depth == 0) g_gate.unlock();
        g_mu.unlock();
    }
};

// The current HIP device is per host thread: an entry point called from a thread other than the one that ran
// mw_init (the reference's OpenMP sections, a Python worker thread) would otherwise allocate and launch on
// device 0.  Every entry point makes the engine's device current and restores the caller's on return.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    DeviceGuard()
    {
        if (g.live && hipGetDevice(&prev) == hipSuccess && prev != g.device)
            switched = hipSetDevice(g.device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

constexpr size_t kQueue1024 = (size_t)(mw::kQCap + 1) * 1024 * sizeof(uint32_t);
constexpr size_t kQueue256 = (size_t)(mw::kQCap + 1) * 256 * sizeof(uint32_t);
constexpr int kFullLayout = mw::kLayoutPair;      // LDS layout of the full-box kernel's staged vectors (mw_full_energy.hip.h)
bool lds_fits(int N, int ivcap)
{
    return kQueue1024 + mw::lds_vec_bytes((size_t)N) + mw::lds_vec_bytes((size_t)ivcap) <= (size_t)kLdsBudget;
}
constexpr size_t kMoveScratch = 16 * sizeof(mw::WaveScratch);
constexpr size_t kMoveStage = (size_t)mw::kMoveChunk * sizeof(int);   // the molecules of an item's requests in LDS (at most)
bool lds_fits_move(int N, int ivcap)
{
    return kMoveScratch + mw::lds_vec_bytes((size_t)N) + mw::lds_vec_bytes((size_t)ivcap) + (((size_t)N + 7) & ~(size_t)7) + kMoveStage <= (size_t)kLdsBudget;
}

int check_live() { return g.live ? 0 : fail("mw: engine not initialised (call mw_init / energy_init first)"); }
int check_box(int ils) { return (ils >= 1 && ils <= g.nbox) ? 0 : fail("mw: box index %d outside 1..%d", ils, g.nbox); }
int check_range(int first, int count)
{
    return (first >= 1 && count >= 1 && first + count - 1 <= g.nbox)
               ? 0 : fail("mw: box range %d..%d outside 1..%d", first, first + count - 1, g.nbox);
}
int check_mol(int imol) { return (imol >= 1 && imol <= g.N) ? 0 : fail("mw: molecule index %d outside 1..%d", imol, g.N); }

// Image vectors exactly as compute_ivects builds them (molint.F90:174-217):
// central cell first, then icell, jcell, kcell loops (kcell fastest), (sx+sy)+sz.
int host_ivects(const double h[9], std::vector<double>& out, int imv[3])
{
    const double* h1 = h; const double* h2 = h + 3; const double* h3 = h + 6;
    const double rc = mw::kSmallA * mw::kSigma;
    const int im = (int)std::floor(rc / std::sqrt(h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2])) + 1;   // :189
    const int jm = (int)std::floor(rc / std::sqrt(h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2])) + 1;
    const int km = (int)std::floor(rc / std::sqrt(h3[0] * h3[0] + h3[1] * h3[1] + h3[2] * h3[2])) + 1;
    const long long n = (long long)(2 * im + 1) * (2 * jm + 1) * (2 * km + 1);                            // :193
    imv[0] = im; imv[1] = jm; imv[2] = km;
    if (n > MW_MAX_IVECT) return -1;
    out.assign((size_t)n * 3, 0.0);                                                                       // :197
    size_t k = 1;
    for (int ic = -im; ic <= im; ++ic) {
        const double sx[3] = {(double)ic * h1[0], (double)ic * h1[1], (double)ic * h1[2]};               // :201
        for (int jc = -jm; jc <= jm; ++jc) {
            const double sy[3] = {(double)jc * h2[0], (double)jc * h2[1], (double)jc * h2[2]};           // :203
            for (int kc = -km; kc <= km; ++kc) {
                if (ic == 0 && jc == 0 && kc == 0) continue;                                             // :207
                const double sz[3] = {(double)kc * h3[0], (double)kc * h3[1], (double)kc * h3[2]};       // :205
                for (int d = 0; d < 3; ++d) {
                    volatile double s = sx[d] + sy[d];   // keep (sx+sy)+sz unfused and in this order     :208
                    out[3 * k + d] = s + sz[d];
                }
                ++k;
            }
        }
    }
    return (int)n;
}

// Grid for the cell-list neighbour builder: spacing >= list radius along every cell vector.
// nc = 0 means "fewer than 3 cells somewhere": that box keeps the brute-force kernel.
mw::GridDesc make_grid(const double h[9], const int imv[3], int max_cells)
{
    mw::GridDesc G;
    std::memset(&G, 0, sizeof G);
    const double* a = h; const double* b = h + 3; const double* c = h + 6;     // cell vectors
    const double bc[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
    const double ca[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
    const double ab[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const double det = a[0] * bc[0] + a[1] * bc[1] + a[2] * bc[2];
    G.im[0] = imv[0]; G.im[1] = imv[1]; G.im[2] = imv[2];
    if (!(std::fabs(det) > 0.0)) return G;
    // r = s1 a + s2 b + s3 c  =>  s1 = (b x c).r / det, ...
    for (int d = 0; d < 3; ++d) { G.hinv[d] = bc[d] / det; G.hinv[3 + d] = ca[d] / det; G.hinv[6 + d] = ab[d] / det; }
    const double rn = mw::kRn * (1.0 + 1.0e-9);
    const double* cr[3] = {bc, ca, ab};
    int nc[3];
    for (int d = 0; d < 3; ++d) {
        const double width = std::fabs(det) / std::sqrt(cr[d][0] * cr[d][0] + cr[d][1] * cr[d][1] + cr[d][2] * cr[d][2]);
        const double q = std::floor(width / rn);
        nc[d] = q > 1024.0 ? 1024 : (int)q;
        if (nc[d] < 3) return G;                          // nc stays 0: brute force for this box
        if (imv[d] > 500) return G;
    }
    while ((long long)nc[0] * nc[1] * nc[2] > max_cells) {   // coarser cells are still valid cells
        int big = 0;
        if (nc[1] > nc[big]) big = 1;
        if (nc[2] > nc[big]) big = 2;
        if (nc[big] <= 3) return G;
        --nc[big];
    }
    G.nc[0] = nc[0]; G.nc[1] = nc[1]; G.nc[2] = nc[2];
    G.ncell = nc[0] * nc[1] * nc[2];
    for (int d = 0; d < 9; ++d) G.h[d] = h[d];
    // Error bound of k_cell_pairs' single-precision squared distance.  Coordinates there are relative to a grid
    // cell's origin: |.| <= 2 D for a candidate, D for the molecule itself, D = the grid cell's longest diagonal.
    // Each coordinate difference carries at most 8 ulp(D) of rounding (conversions, the piece offset, the
    // subtraction), the squared sum 2 sqrt(3) r delta + 4 ulp(r^2) at r ~ rn.  Doubled for safety; a pair whose
    // single-precision r^2 lies within eps of rn^2 is re-decided in double precision by the reference's expression.
    double D = 0.0;
    for (int sg = 0; sg < 4; ++sg) {
        const double s1 = (sg & 1) ? -1.0 : 1.0, s2 = (sg & 2) ? -1.0 : 1.0;
        double v[3];
        for (int d = 0; d < 3; ++d) v[d] = a[d] / nc[0] + s1 * b[d] / nc[1] + s2 * c[d] / nc[2];
        D = std::max(D, std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
    }
    const double ulp = 5.9604644775390625e-08, r = mw::kRn + 1.0;      // 2^-24
    G.eps = (float)(2.0 * (2.0 * std::sqrt(3.0) * r * 8.0 * ulp * D + 4.0 * ulp * r * r));
    return G;
}

int grow_ivcap(int need)
{
    int cap = std::max(g.ivcap, (need + 3) & ~3);       // (as much as asked for: the Monte Carlo driver keeps every box's table in LDS,
    if (cap > MW_MAX_IVECT) cap = MW_MAX_IVECT;         //  where a doubled capacity cost the 48-molecule walkers their eighth place per CU)
    std::vector<double> nh((size_t)g.nbox * cap * 3, 0.0);
    for (int b = 0; b < g.nbox; ++b)
        std::memcpy(&nh[(size_t)b * cap * 3], &g.h_ivect[(size_t)b * g.ivcap * 3], sizeof(double) * 3 * g.ivcap);
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipFree(g.d_ivect));
    HIPCHK(hipMalloc(&g.d_ivect, nh.size() * sizeof(double)));
    HIPCHK(hipMemcpy(g.d_ivect, nh.data(), nh.size() * sizeof(double), hipMemcpyHostToDevice));
    g.h_ivect.swap(nh);
    g.ivcap = cap;
    return 0;
}

int ensure_moves(int n)
{
    if (n <= g.mcap) return 0;
    HIPCHK(hipStreamSynchronize(g.stream));
    if (g.d_mimol) {
        HIPCHK(hipFree(g.d_mimol)); HIPCHK(hipFree(g.d_mtrial)); HIPCHK(hipFree(g.d_meold));
        HIPCHK(hipFree(g.d_menew)); HIPCHK(hipFree(g.d_mcnt)); HIPCHK(hipFree(g.d_mperm)); HIPCHK(hipFree(g.d_mdecl));
        g.d_mimol = nullptr; g.mcap = 0;
    }
    int cap = 1024;
    while (cap < n) cap *= 2;
    HIPCHK(hipMalloc(&g.d_mimol, sizeof(int) * cap));
    HIPCHK(hipMalloc(&g.d_mtrial, sizeof(double) * 3 * cap));
    HIPCHK(hipMalloc(&g.d_meold, sizeof(double) * cap));
    HIPCHK(hipMalloc(&g.d_menew, sizeof(double) * cap));
    HIPCHK(hipMalloc(&g.d_mcnt, sizeof(unsigned int) * 4 * cap));
    HIPCHK(hipMalloc(&g.d_mperm, sizeof(int) * cap));
    HIPCHK(hipMalloc(&g.d_mdecl, sizeof(int) * (2 * (size_t)cap + 2)));
    HIPCHK(hipMemset(g.d_mdecl, 0, 2 * sizeof(int)));
    g.mdecl_par = 0;
    g.mcap = cap;
    return 0;
}

// Launch geometry of the full-box kernel for `count` boxes.
struct Geo { bool lds; int block, nsplit, chunk; size_t shmem; };
Geo model_geo(int count)
{
    Geo ge;
    ge.lds = lds_fits(g.N, g.ivcap);
    if (ge.lds) {
        // one workgroup stages the whole box; split a box over several workgroups only
        // when there are too few boxes to occupy the 256 CUs
        ge.block = 1024;
        int want = (2 * g.cu + count - 1) / count;
        int maxsplit = (g.N + ge.block - 1) / ge.block;
        ge.nsplit = want < 1 ? 1 : (want > maxsplit ? maxsplit : want);
        ge.shmem = kQueue1024 + mw::lds_vec_bytes((size_t)g.N) + mw::lds_vec_bytes((size_t)g.ivcap);
    } else {
        ge.block = 256;
        ge.nsplit = (g.N + ge.block - 1) / ge.block;
        ge.shmem = kQueue256 + mw::lds_vec_bytes((size_t)g.ivcap);
    }
    if (ge.nsplit > g.nsplit_max) ge.nsplit = g.nsplit_max;
    ge.chunk = (((g.N + ge.nsplit - 1) / ge.nsplit) + 63) & ~63;   // whole groups of 64 list columns (cmax is per group)
    return ge;
}

int launch_model_energy(int first, int count, bool with_mom = false, bool write_energy = true)
{
    const Geo ge = model_geo(count);
    double* mom = nullptr;
    if (with_mom && ge.lds) {
        if (!g.d_mom) HIPCHK(hipMalloc(&g.d_mom, (size_t)g.nbox * g.N * mw::kMomStride * sizeof(double)));
        mom = g.d_mom;
    }
    const int wen = write_energy ? 1 : 0;
    // whole boxes staged in LDS, one workgroup per box: the workgroups are persistent, one per compute unit (its LDS holds
    // one), each taking every g.cu-th box and reading its next box while the current one's tail drains
    static const bool persist = !(std::getenv("MW_MODEL_PERSIST") && std::getenv("MW_MODEL_PERSIST")[0] == '0');   // 0: one workgroup per box (A/B only)
    dim3 grid(ge.nsplit, persist && ge.lds && ge.nsplit == 1 ? std::min(count, g.cu) : count);
    const int box0 = first - 1;
    if (ge.lds && mom)
        hipLaunchKernelGGL((mw::k_model_energy<true, 1024, kFullLayout, false, true>), grid, dim3(1024), ge.shmem, g.stream, g.d_pos, g.d_ivect,
                           g.d_nivect, g.d_list, g.d_order, g.d_nns, g.d_cmax, g.d_partial, g.d_cpartial, g.d_energy, g.d_counts, g.N, g.S, g.ivcap, box0, ge.nsplit, ge.chunk, count,
                           mom, wen);
    else if (ge.lds)
        hipLaunchKernelGGL((mw::k_model_energy<true, 1024, kFullLayout>), grid, dim3(1024), ge.shmem, g.stream, g.d_pos, g.d_ivect,
                           g.d_nivect, g.d_list, g.d_order, g.d_nns, g.d_cmax, g.d_partial, g.d_cpartial, g.d_energy, g.d_counts, g.N, g.S, g.ivcap, box0, ge.nsplit, ge.chunk, count,
                           (double*)nullptr, wen);
    else
        hipLaunchKernelGGL((mw::k_model_energy<false, 256, kFullLayout, true>), grid, dim3(256), ge.shmem, g.stream, g.d_pos, g.d_ivect,
                           g.d_nivect, g.d_list, g.d_order, g.d_nns, g.d_cmax, g.d_partial, g.d_cpartial, g.d_energy, g.d_counts, g.N, g.S, g.ivcap, box0, ge.nsplit, ge.chunk, count,
                           mom, wen);
    HIPCHK(hipGetLastError());
    if (mom) { g.mom_first = first; g.mom_count = count; g.swm_count = 0; }   // (d_mom rewritten for these boxes: the driver's claim on it ends -- its launch renews it)
    if (ge.nsplit > 1 && write_energy) {           // split boxes: the partials of box b live at [b*nsplit .. b*nsplit+nsplit); unsplit boxes wrote their energy themselves
        hipLaunchKernelGGL(mw::k_sum_partials, dim3(count), dim3(64), 0, g.stream, g.d_partial, g.d_cpartial,
                           g.d_energy, g.d_counts, box0, count, ge.nsplit);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

size_t sort_box_lds_bytes() { return (size_t)g.N * 24 + ((size_t)g.cstride + 1) * 4; }

int launch_build(int first, int count)
{
    const int box0 = first - 1;
    ++g.list_version;
    for (int b = box0; b < box0 + count; ++b) g.h_listbuilt[(size_t)b] = 1;
    g.swm_count = 0;                     // (the driver's moments of walkers in global memory are made afresh after every list build: what
                                         //  the accepted moves' updates add in rounding stays bounded by a list interval, as the walkers in LDS
                                         //  have it per launch)
    int ngrid = 0;
    for (int b = box0; b < box0 + count; ++b) ngrid += g.h_usegrid[b] ? 1 : 0;
    const bool fused_sort = ngrid > 0 && g.sort_in_lds && !g.legacy_search;       // k_cell_sort_box resets the statistics itself
    if (!fused_sort) {
        hipLaunchKernelGGL(mw::k_init_stats, dim3((count + 255) / 256), dim3(256), 0, g.stream, g.d_stats, box0, count);   // {min, max} per box
        HIPCHK(hipGetLastError());
    }
    dim3 grid((g.N + 255) / 256, count);
    if (ngrid > 0) {
        if (fused_sort) {
            // boxes whose cell-ordered records fit LDS: bin + scan + scatter in one workgroup per box
            hipLaunchKernelGGL(mw::k_cell_sort_box, dim3(count), dim3(1024), sort_box_lds_bytes(), g.stream, g.d_pos, g.d_grid,
                               g.d_cstart, g.d_wpos, g.d_wsh, g.d_stats, g.N, g.cstride, box0);
            HIPCHK(hipGetLastError());
        } else {
        HIPCHK(hipMemsetAsync(g.d_ccount + (size_t)box0 * g.cstride, 0, sizeof(int) * (size_t)count * g.cstride, g.stream));
        hipLaunchKernelGGL(mw::k_cell_bin, grid, dim3(256), 0, g.stream, g.d_pos, g.d_grid, g.d_cellid, g.d_shift, g.d_wrel, g.d_ccount,
                           g.N, g.cstride, box0);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(mw::k_cell_scan, dim3(count), dim3(1024), 0, g.stream, g.d_grid, g.d_ccount, g.d_cstart, g.d_ccursor,
                           g.cstride, box0);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(mw::k_cell_scatter, grid, dim3(256), 0, g.stream, g.d_grid, g.d_cellid, g.d_shift, g.d_wrel, g.d_ccursor,
                           g.d_sorted, g.d_wpos, g.d_wsh, g.N, g.cstride, box0);
        HIPCHK(hipGetLastError());
        }
        if (g.legacy_search) {
            hipLaunchKernelGGL(mw::k_cell_search, grid, dim3(256), (size_t)g.S * 256 * sizeof(uint32_t), g.stream, g.d_pos, g.d_ivect,
                               g.d_grid, g.d_cellid, g.d_shift, g.d_cstart, g.d_sorted, g.d_listm, g.d_nn, g.d_cin, g.d_stats,
                               g.N, g.S, g.ivcap, g.cstride, box0);
        } else {
            // one wavefront per block of grid cells along the third axis, four per workgroup; cells per block: enough
            // for ~17 molecules per wavefront -- one block of kPairIB rows, and the block's candidates still fit one
            // register batch (measured on 512 x 4096 ice: 3 cells 1.35 ms, 4 cells 1.18 ms, 5 cells 1.19 ms;
            // MW_PAIR_BCELLS overrides)
            int bcells = 1, maxblocks = 0;
            for (int b = box0; b < box0 + count; ++b)
                if (g.h_usegrid[b]) { bcells = std::max(bcells, (int)(17.5 * g.h_grid[(size_t)b].ncell / g.N + 0.5)); }
            if (const char* ev = std::getenv("MW_PAIR_BCELLS")) bcells = std::atoi(ev);
            bcells = std::max(1, std::min(bcells, mw::kPairMaxB));
            for (int b = box0; b < box0 + count; ++b) {
                if (!g.h_usegrid[b]) continue;
                const mw::GridDesc& G = g.h_grid[(size_t)b];
                const int B = std::min(bcells, G.nc[2]);
                maxblocks = std::max(maxblocks, G.nc[0] * G.nc[1] * ((G.nc[2] + B - 1) / B));
            }
            const int nwg = (maxblocks + 3) / 4, count8 = (count + 7) & ~7;       // a 1-D grid: the kernel maps workgroup -> (box, cell blocks) by XCD
            hipLaunchKernelGGL(mw::k_cell_pairs, dim3((unsigned)nwg * (unsigned)count8), dim3(256), 0, g.stream, g.d_pos, g.d_ivect, g.d_grid,
                               g.d_cstart, g.d_wpos, g.d_wsh, g.d_listm, g.d_nn, g.d_cin, g.d_stats, g.N, g.S, g.ivcap, g.cstride, box0, bcells,
                               nwg, count);
        }
        HIPCHK(hipGetLastError());
    }
    if (ngrid < count) {
        const int bt = std::min(256, (g.N + 63) & ~63);                           // a block no larger than the box needs
        hipLaunchKernelGGL(mw::k_build_neighbours, dim3((g.N + bt - 1) / bt, count), dim3(bt), 0, g.stream, g.d_pos, g.d_ivect, g.d_nivect,
                           g.d_listm, g.d_nn, g.d_cin, g.d_stats, g.d_usegrid, g.N, g.S, g.ivcap, box0);
        HIPCHK(hipGetLastError());
    }
    // the slot-major layout of the full-box kernel, columns sorted by work (mw_neighbours.hip.h, k_list_order)
    {
        const int nseg = (g.N + g.order_seg - 1) / g.order_seg;
        const int ngroups = (std::min(g.N, g.order_seg) + 63) / 64;
        const size_t shmem = g.order_kbits < 0 ? 0 : sizeof(int) * ((size_t)ngroups << g.order_kbits);
        const int nthreads = std::min(1024, std::max(64, (std::min(g.N, g.order_seg) + 63) & ~63));
        hipLaunchKernelGGL(mw::k_list_order, dim3(nseg, count), dim3(nthreads), shmem, g.stream, g.d_listm, g.d_nn, g.d_cin, g.d_stats,
                           g.d_list, g.d_order, g.d_nns, g.d_cmax, g.N, g.S, box0, g.order_kbits, g.order_seg);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

int finish_build(int first, int count, int* min_nn, int* max_nn)
{
    std::vector<int> st((size_t)count * 2);
    HIPCHK(hipMemcpyAsync(st.data(), g.d_stats + 2 * (first - 1), sizeof(int) * 2 * count, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    int mn = 0x7fffffff, mx = 0, worst = -1;
    for (int b = 0; b < count; ++b) {
        if (st[2 * b] < mn) mn = st[2 * b];
        if (st[2 * b + 1] > mx) { mx = st[2 * b + 1]; worst = first + b; }
    }
    if (min_nn) *min_nn = mn;
    if (max_nn) *max_nn = mx;
    if (first == 1 && count == g.nbox) { g.nnmax_cached = mx; g.nnmax_version = g.list_version; }   // (the driver's launch asks for the longest row
                                                                                                   //  of any box: no second read-back of the same words)
    if (mx > g.S)
        return fail("mw: neighbour list overflow in box %d: a molecule has %d entries, maxneigh = %d", worst, mx, g.S);
    return 0;
}

// Free everything the context holds (any subset may be allocated: mw_init's failure path comes here too).
void release_all()
{
    if (g.srv_running) (void)server_stop();
    if (g.stream) { (void)hipSetDevice(g.device); (void)hipStreamSynchronize(g.stream); }
    if (g.sstream) { (void)hipStreamSynchronize(g.sstream); (void)hipStreamDestroy(g.sstream); }
    if (g.h_head) (void)hipHostFree(g.h_head);
    if (g.h_slots) (void)hipHostFree(g.h_slots);
    if (g.req_dev_alloc) { (void)hipFree(g.req_dev_alloc); g.req_dev_alloc = nullptr; }
    void* ptrs[] = {g.d_hmat, g.d_sw_mubin, g.d_sw_binwidth, g.d_wweight, g.d_whist, g.d_wuhist, g.d_wls, g.d_wmu, g.d_wacc,
                    g.d_wswitch, g.d_wshift, g.d_wvol, g.d_wflag, g.d_wwin, g.d_wfac, g.d_wsum, g.d_winflag, g.d_wstep, g.d_volume, g.d_swlog, g.d_tabscratch, g.d_pos, g.d_ivect,
                    g.d_nivect, g.d_list, g.d_listm, g.d_nn, g.d_stats, g.d_order, g.d_nns, g.d_cmax, g.d_cin, g.d_grid,
                    g.d_usegrid, g.d_cellid, g.d_shift, g.d_sorted, g.d_wrel, g.d_wpos, g.d_wsh, g.d_ccount, g.d_cstart, g.d_ccursor, g.d_partial,
                    g.d_cpartial, g.d_energy, g.d_counts, g.d_mimol, g.d_mtrial, g.d_meold, g.d_menew, g.d_mcnt, g.d_mperm, g.d_mdecl,
                    g.d_mwork, g.d_mom, g.d_mtot, g.d_wmom, g.d_pm, g.d_srvmomok};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (g.ev_srv) { (void)hipEventDestroy(g.ev_srv); g.ev_srv = nullptr; }
    if (g.h_pin) (void)hipHostFree(g.h_pin);
    if (g.h_stage) (void)hipHostFree(g.h_stage);
    for (int s = 0; s < kTimerSlots; ++s) {
        if (g.ev[s][0]) (void)hipEventDestroy(g.ev[s][0]);
        if (g.ev[s][1]) (void)hipEventDestroy(g.ev[s][1]);
    }
    if (g.stream) (void)hipStreamDestroy(g.stream);
    g = Ctx();
}

}  // namespace

extern "C" {

const char* mw_last_error(void) { return g_err.c_str(); }

int mw_is_initialised(void) { return g.live ? 1 : 0; }

int mw_constants(double out[8])
{
    MW_LOCK;
    out[0] = mw::kSigma; out[1] = mw::kEpsilon; out[2] = mw::kLambda; out[3] = mw::kBigA;
    out[4] = mw::kBigB;  out[5] = mw::kGamma;   out[6] = mw::kSmallA; out[7] = mw::kCos0;
    return 0;
}

// The instantiations of the Monte Carlo driver: lattices per walker x residency (0: positions and rows from global memory /
// L2, 1: positions in LDS, 2: positions and list rows in LDS) x with / without volume moves; and, for walkers whose data
// stay in global memory, look-ahead over 2 or 4 moves (wavefronts per workgroup = lattices x look-ahead).
static const void* sweep_kernel(int nlat, int residency, bool withvol, int spec)
{
#define MW_SWEEP_K(L, SP, P, R, V) reinterpret_cast<const void*>(&mw::k_sweep<L, SP, P, R, V>)
    static const void* const tab[2][3][2] = {
        {{MW_SWEEP_K(1, 1, false, false, false), MW_SWEEP_K(1, 1, false, false, true)},
         {MW_SWEEP_K(1, 1, true, false, false),  MW_SWEEP_K(1, 1, true, false, true)},
         {MW_SWEEP_K(1, 1, true, true, false),   MW_SWEEP_K(1, 1, true, true, true)}},
        {{MW_SWEEP_K(2, 1, false, false, false), MW_SWEEP_K(2, 1, false, false, true)},
         {MW_SWEEP_K(2, 1, true, false, false),  MW_SWEEP_K(2, 1, true, false, true)},
         {MW_SWEEP_K(2, 1, true, true, false),   MW_SWEEP_K(2, 1, true, true, true)}}};
    static const void* const ahead[2][2][2] = {       // [lattices][look-ahead 2 / 4][volume moves], residency 0
        {{MW_SWEEP_K(1, 2, false, false, false), MW_SWEEP_K(1, 2, false, false, true)},
         {MW_SWEEP_K(1, 4, false, false, false), MW_SWEEP_K(1, 4, false, false, true)}},
        {{MW_SWEEP_K(2, 2, false, false, false), MW_SWEEP_K(2, 2, false, false, true)},
         {MW_SWEEP_K(2, 4, false, false, false), MW_SWEEP_K(2, 4, false, false, true)}}};
    static const void* const ahead_lds[2][2][2] = {   // the same for walkers entirely in LDS (residency 2): the reference's own handful
        {{MW_SWEEP_K(1, 2, true, true, false), MW_SWEEP_K(1, 2, true, true, true)},          // of 48-molecule walkers is a handful of
         {MW_SWEEP_K(1, 4, true, true, false), MW_SWEEP_K(1, 4, true, true, true)}},         // chains, and their speed is the chain's
        {{MW_SWEEP_K(2, 2, true, true, false), MW_SWEEP_K(2, 2, true, true, true)},
         {MW_SWEEP_K(2, 4, true, true, false), MW_SWEEP_K(2, 4, true, true, true)}}};
    static const void* const ahead_pos[2][2][2] = {   // ... and for the sizes in between (positions in LDS, rows in global memory)
        {{MW_SWEEP_K(1, 2, true, false, false), MW_SWEEP_K(1, 2, true, false, true)},
         {MW_SWEEP_K(1, 4, true, false, false), MW_SWEEP_K(1, 4, true, false, true)}},
        {{MW_SWEEP_K(2, 2, true, false, false), MW_SWEEP_K(2, 2, true, false, true)},
         {MW_SWEEP_K(2, 4, true, false, false), MW_SWEEP_K(2, 4, true, false, true)}}};
    static const void* const ahead8[2] = {            // eight moves in flight: ONE lattice, walkers in global memory (a 4096-molecule box or a few of them)
        MW_SWEEP_K(1, 8, false, false, false), MW_SWEEP_K(1, 8, false, false, true)};
    static const void* const ahead6[2] = {            // six moves in flight: TWO lattices entirely in LDS (twelve wavefronts of the 168-register builds fill a CU:
        MW_SWEEP_K(2, 6, true, true, false), MW_SWEEP_K(2, 6, true, true, true)};          // one walker per CU -- the reference's handful of 48-molecule walkers)
#undef MW_SWEEP_K
    if (spec == 8) return (nlat == 1 && residency == 0) ? ahead8[withvol ? 1 : 0] : nullptr;
    if (spec == 6) return (nlat == 2 && residency == 2) ? ahead6[withvol ? 1 : 0] : nullptr;
    if (spec > 1 && residency == 0) return ahead[nlat - 1][spec == 4 ? 1 : 0][withvol ? 1 : 0];
    if (spec > 1 && residency == 1) return ahead_pos[nlat - 1][spec == 4 ? 1 : 0][withvol ? 1 : 0];
    if (spec > 1 && residency == 2) return ahead_lds[nlat - 1][spec == 4 ? 1 : 0][withvol ? 1 : 0];
    return tab[nlat - 1][residency][withvol ? 1 : 0];
}

static int init_impl(int device, int nwater, int nboxes, int maxneigh);

int mw_init(int device, int nwater, int nboxes, int maxneigh)
{
    MW_LOCK;
    if (g.live) return fail("mw_init: already initialised (call mw_finalize first)");
    const int rc = init_impl(device, nwater, nboxes, maxneigh);
    if (rc != 0) {                       // a failed allocation half way: give back what was taken, keep the message
        const std::string msg = g_err;
        release_all();
        g_err = msg;
    }
    return rc;
}

static int init_impl(int device, int nwater, int nboxes, int maxneigh)
{
    if (nwater < 1 || nboxes < 1) return fail("mw_init: nwater = %d, nboxes = %d must be positive", nwater, nboxes);
    if (nwater > (1 << mw::kJBits)) return fail("mw_init: nwater = %d exceeds the %d-bit packed index", nwater, mw::kJBits);
    if (maxneigh < 1 || maxneigh > MW_MAXNEIGH_LIMIT)
        return fail("mw_init: maxneigh = %d outside 1..%d", maxneigh, MW_MAXNEIGH_LIMIT);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail("mw_init: no HIP device available (%s); this engine has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0) {
        // one process per GPU: take the local rank from the launcher's environment
        device = 0;
        const char* vars[] = {"MW_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK",
                              "MPI_LOCALRANKID", "SLURM_LOCALID"};
        for (const char* v : vars) {
            const char* s = std::getenv(v);
            if (s && *s) { device = std::atoi(s) % ndev; if (device < 0) device = 0; break; }
        }
    }
    if (device >= ndev) return fail("mw_init: device %d outside 0..%d", device, ndev - 1);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("mw_init: device %d is %s; libmw_hip.so carries gfx950 code only", device, prop.gcnArchName);

    g = Ctx();
    g.device = device; g.N = nwater; g.nbox = nboxes; g.S = maxneigh; g.ivcap = 32;
    g.cu = prop.multiProcessorCount;
    g.nsplit_max = (nwater + 255) / 256;
    HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    const size_t nb = (size_t)nboxes, N = (size_t)nwater;
    // (+ one staging ticket: k_model_energy reads whole tickets of the next box, the last of which may run past its end)
    HIPCHK(hipMalloc(&g.d_pos, (nb * N * 3 + mw::kStageTicket) * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_ivect, nb * g.ivcap * 3 * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_nivect, nb * sizeof(int)));
    HIPCHK(hipMalloc(&g.d_hmat, nb * 9 * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_volume, nb * sizeof(double)));
    HIPCHK(hipMemset(g.d_volume, 0, nb * sizeof(double)));
    HIPCHK(hipMemset(g.d_hmat, 0, nb * 9 * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_list, nb * N * (size_t)maxneigh * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.d_listm, nb * N * (size_t)mw::kRow * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.d_nn, nb * N * sizeof(int)));
    HIPCHK(hipMalloc(&g.d_stats, nb * 2 * sizeof(int)));
    {
        const size_t ngroups = (N + 63) / 64;
        HIPCHK(hipMalloc(&g.d_order, nb * N * sizeof(int)));
        HIPCHK(hipMalloc(&g.d_nns, nb * N * sizeof(int)));
        HIPCHK(hipMalloc(&g.d_cmax, nb * ngroups * sizeof(int)));
        HIPCHK(hipMalloc(&g.d_cin, nb * N));
        HIPCHK(hipMemset(g.d_nns, 0, nb * N * sizeof(int)));
        HIPCHK(hipMemset(g.d_cmax, 0, nb * ngroups * sizeof(int)));
        HIPCHK(hipMemset(g.d_cin, 0, nb * N));
        // identity order until the first list build (an energy call before any build sees empty rows anyway)
        std::vector<int> ident(nb * N);
        for (size_t b = 0; b < nb; ++b) for (size_t i = 0; i < N; ++i) ident[b * N + i] = (int)i;
        HIPCHK(hipMemcpy(g.d_order, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
        // Segments of k_list_order: the whole box when the full-box kernel stages its positions in LDS; when it gathers
        // them through the caches a wavefront keeps its 64 consecutive molecules (neighbours in index are neighbours in
        // space: measured on 64 x 32768 molecules, sorting over 256 / 1024 / 32768 molecules costs 16 / 80 / 95 % in
        // cache misses, more than the balance gains).  MW_ORDER_SEG overrides (a multiple of 64).
        // Sort key bits: the (key, group) table must fit kOrderSlots.
        g.order_seg = lds_fits(nwater, 32) ? ((nwater + 63) & ~63) : 64;
        if (const char* sg = std::getenv("MW_ORDER_SEG")) { const int v = std::atoi(sg); if (v >= 64) g.order_seg = (v + 63) & ~63; }
        const size_t seg_groups = ((size_t)std::min(nwater, g.order_seg) + 63) / 64;
        g.order_kbits = -1;
        for (int kb = 8; kb >= 0; --kb)
            if ((seg_groups << kb) <= (size_t)mw::kOrderSlots) { g.order_kbits = kb; break; }
    }
    g.cstride = nwater + 64;
    HIPCHK(hipMalloc(&g.d_grid, nb * sizeof(mw::GridDesc)));
    HIPCHK(hipMalloc(&g.d_usegrid, nb * sizeof(int)));
    HIPCHK(hipMalloc(&g.d_cellid, nb * N * sizeof(int)));
    HIPCHK(hipMalloc(&g.d_shift, nb * N * sizeof(int)));
    HIPCHK(hipMalloc(&g.d_sorted, nb * N * sizeof(int)));
    HIPCHK(hipMalloc(&g.d_wrel, nb * N * sizeof(float4)));
    HIPCHK(hipMalloc(&g.d_wpos, nb * N * sizeof(float4)));
    HIPCHK(hipMalloc(&g.d_wsh, nb * N * sizeof(int)));
    { const char* cs = std::getenv("MW_CELL_SEARCH"); g.legacy_search = cs && std::strcmp(cs, "legacy") == 0; }
    {
        const char* cs = std::getenv("MW_CELL_SORT");
        g.sort_in_lds = nwater <= mw::kSortBoxMax && !(cs && std::strcmp(cs, "global") == 0);
        if (g.sort_in_lds)
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_cell_sort_box), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sort_box_lds_bytes()));
    }
    HIPCHK(hipMalloc(&g.d_ccount, nb * (size_t)g.cstride * sizeof(int)));
    HIPCHK(hipMalloc(&g.d_cstart, nb * ((size_t)g.cstride + 1) * sizeof(int)));
    HIPCHK(hipMalloc(&g.d_ccursor, nb * (size_t)g.cstride * sizeof(int)));
    HIPCHK(hipMemset(g.d_grid, 0, nb * sizeof(mw::GridDesc)));
    HIPCHK(hipMemset(g.d_usegrid, 0, nb * sizeof(int)));
    g.h_grid.assign(nb, mw::GridDesc());
    for (auto& G : g.h_grid) std::memset(&G, 0, sizeof G);
    g.h_usegrid.assign(nb, 0);
    g.h_listbuilt.assign(nb, 0);
    { const char* fb = std::getenv("MW_FORCE_BRUTE_NEIGHBOURS"); g.force_brute = fb && *fb && *fb != '0'; }
    HIPCHK(hipMalloc(&g.d_partial, nb * g.nsplit_max * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_cpartial, nb * g.nsplit_max * 2 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&g.d_energy, nb * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_counts, nb * 2 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(g.d_pos, 0, (nb * N * 3 + mw::kStageTicket) * sizeof(double)));
    HIPCHK(hipMemset(g.d_ivect, 0, nb * g.ivcap * 3 * sizeof(double)));
    HIPCHK(hipMemset(g.d_nivect, 0, nb * sizeof(int)));
    HIPCHK(hipMemset(g.d_nn, 0, nb * N * sizeof(int)));
    HIPCHK(hipMemset(g.d_list, 0, nb * N * (size_t)maxneigh * sizeof(uint32_t)));
    HIPCHK(hipMemset(g.d_listm, 0, nb * N * (size_t)mw::kRow * sizeof(uint32_t)));
    HIPCHK(hipMemset(g.d_energy, 0, nb * sizeof(double)));
    HIPCHK(hipMemset(g.d_counts, 0, nb * 2 * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc(&g.h_pin, 4096, hipHostMallocMapped));
    std::memset(g.h_pin, 0, 4096);
    HIPCHK(hipHostGetDevicePointer((void**)&g.d_pin, g.h_pin, 0));
    g.stage_bytes = std::max((size_t)nwater * 3 * sizeof(double), sizeof(mw::CellRecord) + (size_t)MW_MAX_IVECT * 3 * sizeof(double));
    HIPCHK(hipHostMalloc((void**)&g.h_stage, g.stage_bytes, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void**)&g.d_stage, g.h_stage, 0));
    {   // mail slots of the resident local-energy server: one per lattice, at most 8
        g.nslots = std::min(nboxes, 8);
        HIPCHK(hipStreamCreateWithFlags(&g.sstream, hipStreamNonBlocking));
        HIPCHK(hipHostMalloc((void**)&g.h_head, sizeof(mw::MailHead), hipHostMallocMapped));
        HIPCHK(hipHostMalloc((void**)&g.h_slots, sizeof(mw::MailSlot) * 8, hipHostMallocMapped));
        std::memset(g.h_head, 0, sizeof(mw::MailHead));
        std::memset(g.h_slots, 0, sizeof(mw::MailSlot) * 8);
        HIPCHK(hipHostGetDevicePointer((void**)&g.d_head, g.h_head, 0));
        HIPCHK(hipHostGetDevicePointer((void**)&g.d_slots, g.h_slots, 0));
        const char* ev = std::getenv("MW_LOCAL_SERVER");
        g.srv_enabled = !(ev && *ev == '0');
        g_srv_enabled.store(g.srv_enabled, std::memory_order_release);
        g.req_slots = g.h_slots; g.d_req = g.d_slots;
        const char* rq = std::getenv("MW_SERVER_REQ");
        if (!(rq && std::strcmp(rq, "host") == 0)) {
            // Request lines in fine-grained DEVICE memory, written by the host through the PCIe BAR: the server polls
            // local memory (0.45 us a poll instead of a 1.3 us PCIe read, and an idle server puts no traffic on the
            // bus) and a request reaches it as one posted write.  Only where the host can address device memory (large
            // BAR): probed with a system call that reports EFAULT instead of faulting; otherwise, or with
            // MW_SERVER_REQ=host, the request lines stay in host-mapped memory next to the reply line (which the host
            // polls, so it stays there either way).
            void* p = nullptr;
            if (hipExtMallocWithFlags(&p, sizeof(mw::MailSlot) * 8, hipDeviceMallocFinegrained) == hipSuccess && p) {
                bool ok = hipMemset(p, 0, sizeof(mw::MailSlot) * 8) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
                int fds[2];
                if (ok && pipe(fds) == 0) {
                    ok = write(fds[1], p, 8) == 8;
                    close(fds[0]); close(fds[1]);
                } else ok = false;
                if (ok) { g.req_dev_alloc = p; g.req_slots = static_cast<mw::MailSlot*>(p); g.d_req = g.req_slots; }
                else { (void)hipGetLastError(); (void)hipFree(p); }
            } else (void)hipGetLastError();
            if (!g.req_dev_alloc && rq && std::strcmp(rq, "device") == 0)
                std::fprintf(stderr, "mw: MW_SERVER_REQ=device: device memory is not host-addressable here, requests stay in host memory\n");
        }
    }
    g.h_ivect.assign(nb * g.ivcap * 3, 0.0);
    g.h_nivect.assign(nb, 0);
    // the LDS-staged kernel asks for more than the default 64 KiB of dynamic LDS
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_model_energy<true, 1024, kFullLayout, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_model_energy<true, 1024, kFullLayout>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_list_order),
                               hipFuncAttributeMaxDynamicSharedMemorySize, mw::kOrderSlots * (int)sizeof(int)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_cell_search),
                               hipFuncAttributeMaxDynamicSharedMemorySize, MW_MAXNEIGH_LIMIT * 256 * (int)sizeof(uint32_t)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_move_energy<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_move_energy<true, mw::kLayoutSoA, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mw::k_move_energy<true, mw::kLayoutSoA, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    for (int v = 0; v < 12; ++v)   // the sweep driver's dynamic LDS (image vectors of small or sheared cells, staged positions and rows) can pass 64 KiB
        HIPCHK(hipFuncSetAttribute(sweep_kernel(1 + (v & 1), (v >> 1) % 3, v >= 6, 1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
    for (int v = 0; v < 24; ++v)
        HIPCHK(hipFuncSetAttribute(sweep_kernel(1 + (v & 1), v >> 3, (v & 2) != 0, (v & 4) ? 4 : 2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
    for (int v = 0; v < 2; ++v) {
        HIPCHK(hipFuncSetAttribute(sweep_kernel(1, 0, v != 0, 8), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
        HIPCHK(hipFuncSetAttribute(sweep_kernel(2, 2, v != 0, 6), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
    }
    g.live = true;
    return 0;
}

int mw_finalize(void)
{
    MW_LOCK;
    if (!g.live) return 0;
    release_all();
    return 0;
}

int mw_device_info(char* name, int name_len, int* compute_units, long long* global_mem)
{
    MW_LOCK;
    if (check_live()) return 1;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, g.device));
    if (name && name_len > 0) { std::strncpy(name, prop.name, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (global_mem) *global_mem = (long long)prop.totalGlobalMem;
    return 0;
}

// Cells of `count` consecutive boxes in one go: image vectors (compute_ivects, molint.F90:174-217), volume, grid
// descriptors on the host, then ONE copy per device array for the whole range (a farm of thousands of walkers sets up
// in a handful of transfers instead of seven per box).
static int set_cells_impl(int first_ils, int count, const double* h, int* nivect_out)
{
    std::vector<std::vector<double>> ivs((size_t)count);
    std::vector<int> ns((size_t)count), imvs((size_t)count * 3);
    int need = 0;
    for (int k = 0; k < count; ++k) {
        int imv[3] = {1, 1, 1};
        const int n = host_ivects(h + 9 * (size_t)k, ivs[k], imv);
        if (n < 0) return fail("mw_set_cell: cell of box %d is so small that it needs more than %d image vectors", first_ils + k, MW_MAX_IVECT);
        ns[k] = n; imvs[3 * k] = imv[0]; imvs[3 * k + 1] = imv[1]; imvs[3 * k + 2] = imv[2];
        need = std::max(need, n);
    }
    if (need > g.ivcap && grow_ivcap(need)) return 1;
    std::vector<double> vol((size_t)count);
    const size_t b0 = (size_t)(first_ils - 1);
    for (int k = 0; k < count; ++k) {
        const size_t off = (b0 + k) * g.ivcap * 3;
        std::memcpy(&g.h_ivect[off], ivs[k].data(), ivs[k].size() * sizeof(double));
        g.h_nivect[b0 + k] = ns[k];
        // volume(ils) = |det hmatrix(:,:,ils)| as util_determinant expands it (util.f90:16-41; molint.F90:125)
        const double* m = h + 9 * (size_t)k;   // m[(c-1)*3 + (r-1)] = hmatrix(r,c)
        double det = m[0] * (m[4] * m[8] - m[7] * m[5]);
        det = det - m[3] * (m[1] * m[8] - m[7] * m[2]);
        det = det + m[6] * (m[1] * m[5] - m[4] * m[2]);
        vol[k] = std::fabs(det);
        g.h_grid[b0 + k] = make_grid(m, &imvs[3 * k], g.cstride);
        g.h_usegrid[b0 + k] = (!g.force_brute && g.h_grid[b0 + k].nc[0] > 0) ? 1 : 0;
        if (!g.h_usegrid[b0 + k]) g.h_grid[b0 + k].nc[0] = 0;
        else g.grid_on_device = true;
        if (nivect_out) nivect_out[k] = ns[k];
    }
    if (count == 1) {
        // One box (the host's volume move calls compute_ivects four times per attempt, mc_moves.F90:1285-1358,1510-1512): the
        // record goes into pinned memory the device reads in place, and one small kernel files it -- a launch and a
        // synchronisation instead of six transfers.
        mw::CellRecord* rec = reinterpret_cast<mw::CellRecord*>(g.h_stage);
        rec->niv = ns[0]; rec->usegrid = g.h_usegrid[b0]; rec->vol = vol[0];
        std::memcpy(rec->h, h, 9 * sizeof(double));
        rec->grid = g.h_grid[b0];
        std::memcpy(g.h_stage + sizeof(mw::CellRecord), ivs[0].data(), ivs[0].size() * sizeof(double));
        hipLaunchKernelGGL(mw::k_set_cell, dim3(1), dim3(256), 0, g.stream, reinterpret_cast<const mw::CellRecord*>(g.d_stage),
                           reinterpret_cast<const double*>(g.d_stage + sizeof(mw::CellRecord)), g.d_ivect + b0 * g.ivcap * 3, g.d_nivect + b0,
                           g.d_hmat + 9 * b0, g.d_volume + b0, g.d_grid + b0, g.d_usegrid + b0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(g.stream));
        return 0;
    }
    HIPCHK(hipMemcpyAsync(g.d_ivect + b0 * g.ivcap * 3, &g.h_ivect[b0 * g.ivcap * 3], (size_t)count * g.ivcap * 3 * sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_nivect + b0, &g.h_nivect[b0], (size_t)count * sizeof(int), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_hmat + 9 * b0, h, (size_t)count * 9 * sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_volume + b0, vol.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_grid + b0, &g.h_grid[b0], (size_t)count * sizeof(mw::GridDesc), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_usegrid + b0, &g.h_usegrid[b0], (size_t)count * sizeof(int), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));      // `h` and `vol` are the caller's / this frame's
    return 0;
}

int mw_set_cell(int ils, const double h[9], int* nivect_out)
{
    g.swm_count = 0;
    MW_LOCK;
    if (check_live() || check_box(ils)) return 1;
    if (!h) return fail("mw_set_cell: null pointer");
    return set_cells_impl(ils, 1, h, nivect_out);
}

int mw_set_cells_range(int first_ils, int count, const double* h, int* nivect_out)
{
    g.swm_count = 0;
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    if (!h) return fail("mw_set_cells_range: null pointer");
    return set_cells_impl(first_ils, count, h, nivect_out);
}

int mw_get_ivects(int ils, double* out, int max_vectors, int* nivect_out)
{
    MW_LOCK;
    if (check_live() || check_box(ils)) return 1;
    const int n = g.h_nivect[ils - 1];
    if (nivect_out) *nivect_out = n;
    if (out) {
        if (max_vectors < n) return fail("mw_get_ivects: buffer holds %d vectors, box %d has %d", max_vectors, ils, n);
        std::memcpy(out, &g.h_ivect[(size_t)(ils - 1) * g.ivcap * 3], sizeof(double) * 3 * n);
    }
    return 0;
}

int mw_upload_positions(int ils, const double* xyz)
{
    g.swm_count = 0;
    MW_LOCK;
    if (check_live() || check_box(ils)) return 1;
    if (!xyz) return fail("mw_upload_positions: null pointer");
    const size_t bytes = (size_t)g.N * 3 * sizeof(double);
    HIPCHK(hipMemcpyAsync(g.d_pos + (size_t)(ils - 1) * g.N * 3, xyz, bytes, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));   // the caller may overwrite ljr right after we return
    return 0;
}

int mw_download_positions(int ils, double* xyz)
{
    MW_LOCK;
    if (check_live() || check_box(ils)) return 1;
    const size_t bytes = (size_t)g.N * 3 * sizeof(double);
    HIPCHK(hipMemcpyAsync(xyz, g.d_pos + (size_t)(ils - 1) * g.N * 3, bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_upload_positions_range(int first_ils, int count, const double* xyz)
{
    g.swm_count = 0;
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    if (!xyz) return fail("mw_upload_positions_range: null pointer");
    const size_t per = (size_t)g.N * 3;
    HIPCHK(hipMemcpyAsync(g.d_pos + (size_t)(first_ils - 1) * per, xyz, per * count * sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_download_positions_range(int first_ils, int count, double* xyz)
{
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    const size_t per = (size_t)g.N * 3;
    HIPCHK(hipMemcpyAsync(xyz, g.d_pos + (size_t)(first_ils - 1) * per, per * count * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_patch_position(int ils, int imol, const double r[3])
{
    g.swm_count = 0;
    MW_LOCK;
    if (check_live() || check_box(ils) || check_mol(imol)) return 1;
    HIPCHK(hipMemcpyAsync(g.d_pos + ((size_t)(ils - 1) * g.N + (imol - 1)) * 3, r, 3 * sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_build_neighbours_launch(int first_ils, int count)
{
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    for (int b = first_ils; b < first_ils + count; ++b)
        if (g.h_nivect[b - 1] < 1) return fail("mw_build_neighbours: box %d has no cell yet (call mw_set_cell / compute_ivects)", b);
    return launch_build(first_ils, count);
}

int mw_build_neighbours_batch(int first_ils, int count, int* min_nn, int* max_nn)
{
    MW_LOCK;
    if (mw_build_neighbours_launch(first_ils, count)) return 1;
    return finish_build(first_ils, count, min_nn, max_nn);
}

int mw_build_neighbours(int ils, int* min_nn, int* max_nn) { return mw_build_neighbours_batch(ils, 1, min_nn, max_nn); }

int mw_get_neighbours(int ils, int* nn, int* jn, int* vn)
{
    MW_LOCK;
    if (check_live() || check_box(ils)) return 1;
    const size_t N = (size_t)g.N, S = (size_t)g.S, R = (size_t)mw::kRow;
    std::vector<int> hnn(N);
    std::vector<uint32_t> hl(N * R);
    HIPCHK(hipMemcpyAsync(hnn.data(), g.d_nn + (size_t)(ils - 1) * N, N * sizeof(int), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(hl.data(), g.d_listm + (size_t)(ils - 1) * N * R, N * R * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (size_t i = 0; i < N; ++i) {
        if (nn) nn[i] = hnn[i];
        for (size_t s = 0; s < S; ++s) {
            const bool used = (int)s < hnn[i];
            const uint32_t e = used ? hl[i * R + s] : 0u;           // molecule-major rows
            if (jn) jn[i * S + s] = used ? (int)(e & mw::kJMask) + 1 : 0;   // reference layout jn(slot, imol)
            if (vn) vn[i * S + s] = used ? (int)(e >> mw::kJBits) + 1 : 0;
        }
    }
    return 0;
}

int mw_neighbour_total(int first_ils, int count, long long* total_entries)
{
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    std::vector<int> hnn((size_t)count * g.N);
    HIPCHK(hipMemcpyAsync(hnn.data(), g.d_nn + (size_t)(first_ils - 1) * g.N, hnn.size() * sizeof(int), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    long long t = 0;
    for (int v : hnn) t += v;
    *total_entries = t;
    return 0;
}

int mw_model_energy_counts_total(int first_ils, int count, long long* npairs, long long* ntriplets)
{
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    std::vector<unsigned long long> c((size_t)count * 2);
    HIPCHK(hipMemcpyAsync(c.data(), g.d_counts + 2 * (size_t)(first_ils - 1), c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    long long p = 0, t = 0;
    for (int b = 0; b < count; ++b) { p += (long long)c[2 * (size_t)b]; t += (long long)c[2 * (size_t)b + 1]; }
    if (npairs) *npairs = p;
    if (ntriplets) *ntriplets = t;
    return 0;
}

int mw_model_energy_launch(int first_ils, int count)
{
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    return launch_model_energy(first_ils, count);
}

int mw_model_energy_fetch(int first_ils, int count, double* e_out)
{
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    HIPCHK(hipMemcpyAsync(e_out, g.d_energy + (first_ils - 1), sizeof(double) * count, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_model_energy_batch(int first_ils, int count, double* e_out)
{
    MW_LOCK;
    if (mw_model_energy_launch(first_ils, count)) return 1;
    return mw_model_energy_fetch(first_ils, count, e_out);
}

int mw_model_energy(int ils, double* e) { return mw_model_energy_batch(ils, 1, e); }

// compute_model_energy(ils) as the host calls it (molint.F90:407-499; after every volume move, mc_moves.F90:1340): mirror the
// lattice's positions and evaluate, ONE call -- the positions travel through pinned memory, the energy comes back into pinned
// memory, one synchronisation at the end.
int mw_model_energy_of(int ils, const double* xyz, double* e)
{
    g.swm_count = 0;
    MW_LOCK;
    if (check_live() || check_box(ils)) return 1;
    if (!xyz || !e) return fail("mw_model_energy_of: null pointer");
    const size_t bytes = (size_t)g.N * 3 * sizeof(double);
    std::memcpy(g.h_stage, xyz, bytes);
    HIPCHK(hipMemcpyAsync(g.d_pos + (size_t)(ils - 1) * g.N * 3, g.h_stage, bytes, hipMemcpyHostToDevice, g.stream));
    if (launch_model_energy(ils, 1)) return 1;
    HIPCHK(hipMemcpyAsync(g.h_pin + 16, g.d_energy + (ils - 1), sizeof(double), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    *e = g.h_pin[16];
    return 0;
}

int mw_model_energy_counts(int ils, long long* npairs, long long* ntriplets)
{
    MW_LOCK;
    if (check_live() || check_box(ils)) return 1;
    unsigned long long c[2];
    HIPCHK(hipMemcpyAsync(c, g.d_counts + 2 * (size_t)(ils - 1), sizeof c, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    if (npairs) *npairs = (long long)c[0];
    if (ntriplets) *ntriplets = (long long)c[1];
    return 0;
}

// Launch the resident server if it is not running (g_srv_mu held by the caller).
static int server_start_locked()
{
    if (g.srv_running) return 0;
    int prev = -1;
    const bool sw = hipGetDevice(&prev) == hipSuccess && prev != g.device && hipSetDevice(g.device) == hipSuccess;
    g.h_head->quit = 0; g.h_head->exited = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    // a few tenths of a second of empty polls (one poll is a PCIe round trip, ~1 us) and the server leaves by itself
    static const bool stamps = std::getenv("MW_SERVER_STAMPS") != nullptr;
    static const bool plain = std::getenv("MW_SERVER_PLAIN_LOADS") != nullptr;      // experiment only: L1-cached position loads
    // The server's moment path (k_local_server): every molecule's moments of every box, from the full-box kernel, and the positions
    // they belong to -- made HERE, each time the server starts (every entry point that may move a molecule stops it first).  For the
    // drop-in's handful of boxes (a farm's thousands are not served one call at a time); MW_SERVER_MOMENTS=0: off.
    g.swm_count = 0;                     // (single calls patch positions)
    static const bool srvmom = !(std::getenv("MW_SERVER_MOMENTS") && std::getenv("MW_SERVER_MOMENTS")[0] == '0');
    double* mom = nullptr;
    bool allbuilt = true;                // (the full-box kernel over a box that never had a list would follow whatever its arrays hold)
    for (char c : g.h_listbuilt) allbuilt = allbuilt && c != 0;
    if (srvmom && g.nbox <= 64 && model_geo(g.nbox).lds && allbuilt) {
        bool okm = launch_model_energy(1, g.nbox, true, false) == 0 && g.d_mom != nullptr;
        if (okm && !g.d_pm) okm = hipMalloc(&g.d_pm, (size_t)g.nbox * g.N * 3 * sizeof(double)) == hipSuccess;
        if (okm && !g.d_srvmomok) okm = hipMalloc(&g.d_srvmomok, (size_t)g.nbox * sizeof(int)) == hipSuccess;
        if (okm && !g.ev_srv) okm = hipEventCreateWithFlags(&g.ev_srv, hipEventDisableTiming) == hipSuccess;
        if (okm) {
            // (no host wait: the server's stream waits for the moments on the device -- 52 -> 20-odd us per server start, which a host
            //  with volume moves pays every few dozen calls; "still in step" = any non-zero word)
            okm = hipMemcpyAsync(g.d_pm, g.d_pos, (size_t)g.nbox * g.N * 3 * sizeof(double), hipMemcpyDeviceToDevice, g.stream) == hipSuccess
               && hipMemsetAsync(g.d_srvmomok, 1, (size_t)g.nbox * sizeof(int), g.stream) == hipSuccess
               && hipEventRecord(g.ev_srv, g.stream) == hipSuccess
               && hipStreamWaitEvent(g.sstream, g.ev_srv, 0) == hipSuccess;
        }
        if (okm) mom = g.d_mom;
        else (void)hipGetLastError();
        g.mom_count = 0;                 // (the server will change them under the batch kernels' feet: not theirs to reuse)
    }
    if (plain)
        hipLaunchKernelGGL(mw::k_local_server<false>, dim3(g.nslots), dim3(64), 0, g.sstream, g.d_head, g.d_slots, g.d_req, g.d_pos, g.d_ivect,
                           g.d_nivect, g.d_listm, g.d_nn, g.N, g.ivcap, 300000LL, stamps ? 1 : 0, mom, g.d_pm, g.d_srvmomok);
    else
        hipLaunchKernelGGL(mw::k_local_server<true>, dim3(g.nslots), dim3(64), 0, g.sstream, g.d_head, g.d_slots, g.d_req, g.d_pos, g.d_ivect,
                           g.d_nivect, g.d_listm, g.d_nn, g.N, g.ivcap, 300000LL, stamps ? 1 : 0, mom, g.d_pm, g.d_srvmomok);
    const hipError_t err = hipGetLastError();
    if (sw) (void)hipSetDevice(prev);
    if (err != hipSuccess) return fail("mw: launching the local-energy server failed: %s", hipGetErrorString(err));
    g.srv_running = true;
    return 0;
}

namespace {
// Stop the server and wait for it (called with g_gate held exclusively: no request is in flight).
int server_stop()
{
    std::lock_guard<std::mutex> lk(g_srv_mu);
    if (!g.srv_running) return 0;
    reinterpret_cast<volatile int*>(&g.h_head->quit)[0] = 1;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    int prev = -1;
    const bool sw = hipGetDevice(&prev) == hipSuccess && prev != g.device && hipSetDevice(g.device) == hipSuccess;
    const hipError_t err = hipStreamSynchronize(g.sstream);
    if (sw) (void)hipSetDevice(prev);
    g.srv_running = false;
    if (err != hipSuccess) return fail("mw: the local-energy server ended with %s", hipGetErrorString(err));
    return 0;
}
}  // namespace

// Wait for the reply to request `seq` of mail slot `sl` (the slot's mutex and g_gate shared are held by the caller).
static int server_wait(int sl, unsigned long long seq, double* e)
{
    volatile mw::MailSlot* m = g.h_slots + sl;
    // The reply normally shows within microseconds.  Every few thousand polls (a read of host memory that only changes when
    // a wavefront leaves): is the server still there?  It retires by itself after its idle limit, and a request posted just
    // then would otherwise wait for a restart nobody triggers.  The no-reply limit is wall-clock (MW_SERVER_TIMEOUT seconds,
    // default 20), not a poll count.
    static const double timeout_s = [] { const char* ev = std::getenv("MW_SERVER_TIMEOUT"); const double v = ev ? atof(ev) : 0.0; return v > 0.0 ? v : 20.0; }();
    std::chrono::steady_clock::time_point t_post{};
    for (long spin = 1;; ++spin) {
        if (m->rep_seq == seq) break;
        __builtin_ia32_pause();
        if ((spin & 0xfff) == 0 && reinterpret_cast<volatile int*>(&g.h_head->exited)[0] != 0) {
            std::lock_guard<std::mutex> lk(g_srv_mu);
            if (reinterpret_cast<volatile int*>(&g.h_head->exited)[0] != 0) {
                // it left (idle limit, racing with this request) -- or it faulted: the stream tells.  The slots' wavefronts
                // leave one by one: the others are told to go too (each finishes the request it has; a request posted
                // meanwhile is picked up by the server started below), or a slot kept busy by another thread would
                // hold this one up for as long as it stays busy.
                reinterpret_cast<volatile int*>(&g.h_head->quit)[0] = 1;
                std::atomic_thread_fence(std::memory_order_seq_cst);
                int prev = -1;
                const bool sw = hipGetDevice(&prev) == hipSuccess && prev != g.device && hipSetDevice(g.device) == hipSuccess;
                const hipError_t err = hipStreamSynchronize(g.sstream);
                if (sw) (void)hipSetDevice(prev);
                g.srv_running = false;
                if (err != hipSuccess) return fail("mw: the local-energy server ended with %s", hipGetErrorString(err));
                if (m->rep_seq == seq) break;
                if (server_start_locked()) return 1;          // it picks the pending request up: rep_seq != req_seq
            }
        }
        if ((spin & 0xffff) == 0) {
            const auto now = std::chrono::steady_clock::now();
            if (t_post == std::chrono::steady_clock::time_point{}) t_post = now;
            else if (std::chrono::duration<double>(now - t_post).count() > timeout_s)
                return fail("mw: no reply from the local-energy server within %.0f s", timeout_s);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (e) *e = m->energy;
    return 0;
}

// Post one request to the resident server (g_gate shared + the slot's mutex held by the caller); returns its sequence number.
static int server_post(int sl, int ils, int imol, const mw::Override& o1, const mw::Override& o2, unsigned long long* seq_out)
{
    { std::lock_guard<std::mutex> lk(g_srv_mu); if (server_start_locked()) return 1; }
    volatile mw::MailSlot* q = g.req_slots + sl;
    q->box = ils - 1; q->imol = imol - 1;
    q->flags = 1 | (o1.idx >= 0 ? 2 : 0) | (o2.idx >= 0 ? 4 : 0);
    q->prev = o2.idx >= 0 ? o2.idx : 0;
    q->x1 = o1.x; q->y1 = o1.y; q->z1 = o1.z;
    q->x2 = o2.x; q->y2 = o2.y; q->z2 = o2.z;
    const unsigned long long seq = ++g.sseq[sl];
    // fields, then the sequence words: program order for write-back host memory; the store fence keeps it for a
    // write-combining mapping of device memory too
    // (the two sequence words need no order between themselves: the server acts when BOTH show the new number)
    std::atomic_thread_fence(std::memory_order_release); __builtin_ia32_sfence();
    q->seq_a = seq;
    q->seq_b = seq;
    __builtin_ia32_sfence();
    *seq_out = seq;
    return 0;
}

static int served_checks(int ils, int imol, const mw::Override& o2)
{
    if (!g.live) return fail("mw: engine not initialised (call mw_init / energy_init first)");
    if (ils < 1 || ils > g.nbox) return fail("mw: box index %d outside 1..%d", ils, g.nbox);
    if (imol < 1 || imol > g.N) return fail("mw: molecule index %d outside 1..%d", imol, g.N);
    if (o2.idx >= g.N) return fail("mw: molecule index %d outside 1..%d", o2.idx + 1, g.N);
    return 0;
}

// One request through the resident server: g_gate shared (no exclusive entry point is running; everything of the context is
// read under the gate: mw_finalize / mw_init rewrite it) + the slot's mutex.
static int local_energy_served(int ils, int imol, const mw::Override& o1, const mw::Override& o2, double* e)
{
    std::shared_lock<std::shared_mutex> gate(g_gate);
    if (served_checks(ils, imol, o2)) return 1;
    const int sl = (ils - 1) % g.nslots;
    std::lock_guard<std::mutex> slk(g_slot_mu[sl]);
    if (g.spend[sl]) {                                        // a posted request nobody collected: its reply first (the slot holds one request)
        const unsigned long long ps = g.spend[sl];
        g.spend[sl] = 0;
        if (server_wait(sl, ps, nullptr)) return 1;
    }
    unsigned long long seq = 0;
    if (server_post(sl, ils, imol, o1, o2, &seq)) return 1;
    return server_wait(sl, seq, e);
}

int mw_local_energy_patched(int ils, int imol, const double r_imol[3], int imol_prev, const double r_prev[3], double* e)
{
    mw::Override o1, o2;
    o1.idx = -1; o1.x = o1.y = o1.z = 0.0;
    o2 = o1;
    if (r_imol) { o1.idx = imol - 1; o1.x = r_imol[0]; o1.y = r_imol[1]; o1.z = r_imol[2]; }
    if (r_prev && imol_prev >= 1 && imol_prev != imol) {
        o2.idx = imol_prev - 1; o2.x = r_prev[0]; o2.y = r_prev[1]; o2.z = r_prev[2];
    }
    if (g_srv_enabled.load(std::memory_order_acquire)) return local_energy_served(ils, imol, o1, o2, e);

    // MW_LOCAL_SERVER=0: one launch per call (the path the server replaces; kept as its cross-check)
    MW_LOCK;
    if (check_live() || check_box(ils) || check_mol(imol)) return 1;
    if (o2.idx >= g.N) return fail("mw: molecule index %d outside 1..%d", o2.idx + 1, g.N);
    g.swm_count = 0;                     // (the call commits its positions)
    const unsigned long long seq = ++g.pin_seq;
    hipLaunchKernelGGL(mw::k_local_energy_single, dim3(1), dim3(64), 0, g.stream, g.d_pos, g.d_ivect, g.d_listm, g.d_nn,
                       ils - 1, imol - 1, o1, o2, 1, g.d_pin, g.N, g.ivcap,
                       reinterpret_cast<unsigned long long*>(g.d_pin + 8), seq);
    HIPCHK(hipGetLastError());
    // The kernel is the only thing in flight on this stream: wait for its completion word in host-visible memory
    // (a few microseconds less than a stream synchronisation); if it does not show up within a second, fall
    // back to the synchronisation, which also reports a fault.
    volatile unsigned long long* done = reinterpret_cast<volatile unsigned long long*>(g.h_pin + 8);
    bool seen = false;
    std::chrono::steady_clock::time_point t_first{};
    for (long spin = 1;; ++spin) {                       // (a wall clock, like the served path: one second, whatever the host's speed)
        if (*done == seq) { seen = true; break; }
        __builtin_ia32_pause();
        if ((spin & 0xffff) == 0) {
            const auto now = std::chrono::steady_clock::now();
            if (t_first == std::chrono::steady_clock::time_point{}) t_first = now;
            else if (std::chrono::duration<double>(now - t_first).count() > 1.0) break;
        }
    }
    if (!seen) HIPCHK(hipStreamSynchronize(g.stream));
    *e = g.h_pin[0];
    return 0;
}

int mw_local_energy(int ils, int imol, double* e) { return mw_local_energy_patched(ils, imol, nullptr, 0, nullptr, e); }

// The call split in two, for a host that knows its NEXT question while it still waits for the answer to this one (the two
// lattices of a move, mc_moves.F90:1006-1018): post does not wait, collect does.  One posted request per lattice at a time;
// any other single call on that lattice waits for it first.  Both return 2 -- not an error, no message -- when there is
// nothing to gain or to collect: the resident server is switched off (MW_LOCAL_SERVER=0), nothing was posted, or an entry
// point that changes device state ran in between (the reply may predate it: ask again).
int mw_local_energy_post(int ils, int imol, const double r_imol[3], int imol_prev, const double r_prev[3])
{
    if (!g_srv_enabled.load(std::memory_order_acquire)) return 2;
    mw::Override o1, o2;
    o1.idx = -1; o1.x = o1.y = o1.z = 0.0;
    o2 = o1;
    if (r_imol) { o1.idx = imol - 1; o1.x = r_imol[0]; o1.y = r_imol[1]; o1.z = r_imol[2]; }
    if (r_prev && imol_prev >= 1 && imol_prev != imol) { o2.idx = imol_prev - 1; o2.x = r_prev[0]; o2.y = r_prev[1]; o2.z = r_prev[2]; }
    std::shared_lock<std::shared_mutex> gate(g_gate);
    if (served_checks(ils, imol, o2)) return 1;
    const int sl = (ils - 1) % g.nslots;
    std::lock_guard<std::mutex> slk(g_slot_mu[sl]);
    if (g.spend[sl]) {
        const unsigned long long ps = g.spend[sl];
        g.spend[sl] = 0;
        if (server_wait(sl, ps, nullptr)) return 1;
    }
    unsigned long long seq = 0;
    if (server_post(sl, ils, imol, o1, o2, &seq)) return 1;
    g.spend[sl] = seq;
    g.spend_epoch[sl] = g_epoch.load(std::memory_order_relaxed);
    return 0;
}

int mw_local_energy_collect(int ils, double* e)
{
    std::shared_lock<std::shared_mutex> gate(g_gate);
    if (!g.live) return fail("mw: engine not initialised (call mw_init / energy_init first)");
    if (ils < 1 || ils > g.nbox) return fail("mw: box index %d outside 1..%d", ils, g.nbox);
    const int sl = (ils - 1) % g.nslots;
    std::lock_guard<std::mutex> slk(g_slot_mu[sl]);
    const unsigned long long ps = g.spend[sl];
    if (!ps) return 2;
    g.spend[sl] = 0;
    if (server_wait(sl, ps, e)) return 1;
    return g.spend_epoch[sl] == g_epoch.load(std::memory_order_relaxed) ? 0 : 2;
}

int mw_moves_upload(int n, const int* ils, const int* imol, const double* trial_xyz)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (n < 0) return fail("mw_moves_upload: n = %d", n);
    g.mn = 0;
    if (n == 0) return 0;
    if (!ils || !imol) return fail("mw_moves_upload: null request arrays");
    // Bucket the requests by box (stable counting sort): a workgroup then serves requests of ONE
    // box and can stage that box's positions in LDS.  perm maps sorted slot -> caller's index.
    std::vector<int> cnt((size_t)g.nbox + 1, 0);
    for (int m = 0; m < n; ++m) {
        if (ils[m] < 1 || ils[m] > g.nbox) return fail("mw_moves_upload: request %d has box %d outside 1..%d", m, ils[m], g.nbox);
        if (imol[m] < 1 || imol[m] > g.N) return fail("mw_moves_upload: request %d has molecule %d outside 1..%d", m, imol[m], g.N);
        ++cnt[(size_t)ils[m]];
    }
    std::vector<int> start((size_t)g.nbox + 1, 0);
    int used_boxes = 0;
    g.m_noself = true;
    g.m_boxlo = g.nbox; g.m_boxhi = -1; g.m_minreq = n;
    for (int b = 0; b < g.nbox; ++b) {
        start[(size_t)b + 1] = start[b] + cnt[(size_t)b + 1];
        if (cnt[(size_t)b + 1]) {
            ++used_boxes; if (!g.h_usegrid[(size_t)b]) g.m_noself = false;
            g.m_boxlo = std::min(g.m_boxlo, b); g.m_boxhi = std::max(g.m_boxhi, b); g.m_minreq = std::min(g.m_minreq, cnt[(size_t)b + 1]);
        }
    }
    std::vector<int> perm((size_t)n), i0((size_t)n), fill(start.begin(), start.end() - 1);
    std::vector<double> tr(trial_xyz ? (size_t)3 * n : 0);
    for (int m = 0; m < n; ++m) {
        const int s = fill[(size_t)ils[m] - 1]++;
        perm[s] = m; i0[s] = imol[m] - 1;
        if (trial_xyz) { tr[3 * (size_t)s] = trial_xyz[3 * (size_t)m]; tr[3 * (size_t)s + 1] = trial_xyz[3 * (size_t)m + 1]; tr[3 * (size_t)s + 2] = trial_xyz[3 * (size_t)m + 2]; }
    }
    // LDS staging pays when a box's 24N bytes are shared by enough requests
    g.mlds = lds_fits_move(g.N, g.ivcap) && ((long long)n * 2048 >= (long long)used_boxes * 24 * g.N);
    // Requests per work item.  Inside an item the wavefronts draw requests dynamically, so large items waste little
    // at their end and stage the box once for more work; but there must be enough items to fill the chip:
    // aim at >= 2 items per CU, between 256 and the LDS capacity kMoveChunk (measured on 512 x 2048 requests: items of
    // 256 / 512 / 1024 / 2048 requests take 1.375 / 1.316 / 1.291 / 1.286 ms; MW_MOVE_CHUNK overrides).
    int chunk = 16;
    if (g.mlds) {
        const long long want = (long long)n / (2LL * std::max(1, g.cu));
        chunk = 256;
        while (chunk < mw::kMoveChunk && chunk < want) chunk *= 2;
        if (const char* ev = std::getenv("MW_MOVE_CHUNK")) { const int v = std::atoi(ev); if (v >= 64 && v <= mw::kMoveChunk) chunk = v; }
    }
    g.mchunk = chunk;
    std::vector<int4> work;
    for (int b = 0; b < g.nbox; ++b) {
        const int s0 = start[b], cntb = start[(size_t)b + 1] - s0;
        if (cntb == 0) continue;
        const int nitems = (cntb + chunk - 1) / chunk;           // equal shares: no short item at the end of a box
        for (int k = 0; k < nitems; ++k) {
            int4 w; w.x = b; w.y = s0 + (int)((long long)cntb * k / nitems); w.z = s0 + (int)((long long)cntb * (k + 1) / nitems); w.w = 0;
            work.push_back(w);
        }
    }
    // XCD-aware order.  Workgroups are dealt to the 8 XCDs round-robin (workgroup w runs on XCD w % 8) and every XCD
    // has its own L2, so the work items of one box -- which all stage the same positions and walk the same list
    // rows -- are placed on ONE XCD, one after the other: slot k*8 + x holds the k-th item of the boxes with
    // (box index) % 8 == x.  (When the eight sequences differ in length the tail is dealt out as it comes.)
    if (getenv("MW_NO_XCD_ORDER") == nullptr && work.size() >= 16) {
        constexpr int kXcd = 8;
        std::vector<std::vector<int4>> seq(kXcd);
        int boxrank = -1, lastbox = -1;
        for (const int4& w : work) {
            if (w.x != lastbox) { ++boxrank; lastbox = w.x; }      // rank among the boxes that have requests
            seq[(size_t)(boxrank % kXcd)].push_back(w);
        }
        std::vector<int4> ordered;
        ordered.reserve(work.size());
        std::vector<size_t> at(kXcd, 0);
        while (ordered.size() < work.size())
            for (int x = 0; x < kXcd; ++x)
                if (at[x] < seq[x].size()) ordered.push_back(seq[x][at[x]++]);
        work.swap(ordered);
    }
    if (ensure_moves(n)) return 1;
    if ((int)work.size() > g.mwork_cap) {
        HIPCHK(hipStreamSynchronize(g.stream));
        if (g.d_mwork) HIPCHK(hipFree(g.d_mwork));
        g.mwork_cap = (int)work.size() * 2;
        HIPCHK(hipMalloc(&g.d_mwork, sizeof(int4) * g.mwork_cap));
    }
    HIPCHK(hipMemcpyAsync(g.d_mwork, work.data(), sizeof(int4) * work.size(), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_mperm, perm.data(), sizeof(int) * n, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_mimol, i0.data(), sizeof(int) * n, hipMemcpyHostToDevice, g.stream));
    if (trial_xyz) HIPCHK(hipMemcpyAsync(g.d_mtrial, tr.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    g.mwork_n = (int)work.size();
    g.mn = n;
    return 0;
}

static int launch_moves(int mode)
{
    if (g.mn == 0) return 0;
    if (g.mwork_n > g.mtot_cap) {
        HIPCHK(hipStreamSynchronize(g.stream));
        if (g.d_mtot) HIPCHK(hipFree(g.d_mtot));
        g.mtot_cap = 2 * g.mwork_n;
        HIPCHK(hipMalloc(&g.d_mtot, (size_t)g.mtot_cap * 4 * sizeof(unsigned int)));
    }
    g.mtot_n = 0;
    const size_t iv_bytes = kMoveScratch + mw::lds_vec_bytes((size_t)g.ivcap);
    const int kmode = mode | (g.mdecl_par << 2);                          // this launch's count word of the declined list (zeroed by the
    g.mdecl_par ^= 1;                                                     // previous launch's k_move_fallback, or at allocation)
    // The moment path (mw_move_energy.hip.h): boxes staged in LDS, no self-images, and enough requests per box to pay for the
    // full-box pass that makes the moments (one pass costs what ~300 requests save; MW_MOVE_MOMENTS=0 | 1 overrides the count rule).
    // The moments must be those of the positions as they are NOW: they are taken from the last full-box launch only when nothing
    // that can move a molecule has run since (mw_step_launch: the full-box pass of the same step), else made here.
    static const char* momenv = std::getenv("MW_MOVE_MOMENTS");
    const bool mom_ok = g.mlds && g.m_noself && model_geo(1).lds && g.m_boxhi >= g.m_boxlo;
    const bool fresh = g.d_mom && g.mom_count > 0 && g.mom_first - 1 <= g.m_boxlo && g.m_boxhi < g.mom_first - 1 + g.mom_count;
    // (a request saves ~0.3 ns of the launch; a box's moments cost 0.13 us as a by-product of the step's full-box pass, 0.32 us as a
    //  pass of their own: 512 / 1280 requests per box)
    const bool use_mom = mom_ok && (momenv ? momenv[0] != '0' : g.m_minreq >= (fresh ? 512 : 1280));
    if (use_mom) {
        if (!fresh && launch_model_energy(g.m_boxlo + 1, g.m_boxhi - g.m_boxlo + 1, true, false)) return 1;
        hipLaunchKernelGGL((mw::k_move_energy<true, mw::kLayoutSoA, false, true>), dim3(g.mwork_n), dim3(1024),
                           iv_bytes + mw::lds_vec_bytes((size_t)g.N) + (((size_t)g.N + 7) & ~(size_t)7) + (size_t)g.mchunk * sizeof(int), g.stream,
                           g.d_pos, g.d_ivect, g.d_nivect, g.d_listm, g.d_nn, g.d_mwork, g.d_mimol, g.d_mtrial, g.d_mperm,
                           g.d_meold, g.d_menew, g.d_mcnt, g.d_mdecl, g.N, g.ivcap, kmode, (const double*)g.d_mom, g.d_mtot);
        g.mtot_n = g.mwork_n;
    } else if (g.mlds && g.m_noself)
        hipLaunchKernelGGL((mw::k_move_energy<true, mw::kLayoutSoA, false>), dim3(g.mwork_n), dim3(1024),
                           iv_bytes + mw::lds_vec_bytes((size_t)g.N) + (((size_t)g.N + 7) & ~(size_t)7) + (size_t)g.mchunk * sizeof(int), g.stream,
                           g.d_pos, g.d_ivect, g.d_nivect, g.d_listm, g.d_nn, g.d_mwork, g.d_mimol, g.d_mtrial, g.d_mperm,
                           g.d_meold, g.d_menew, g.d_mcnt, g.d_mdecl, g.N, g.ivcap, kmode, (const double*)nullptr, (unsigned int*)nullptr);
    else if (g.mlds)
        hipLaunchKernelGGL(mw::k_move_energy<true>, dim3(g.mwork_n), dim3(1024),
                           iv_bytes + mw::lds_vec_bytes((size_t)g.N) + (((size_t)g.N + 7) & ~(size_t)7) + (size_t)g.mchunk * sizeof(int), g.stream,
                           g.d_pos, g.d_ivect, g.d_nivect, g.d_listm, g.d_nn, g.d_mwork, g.d_mimol, g.d_mtrial, g.d_mperm,
                           g.d_meold, g.d_menew, g.d_mcnt, g.d_mdecl, g.N, g.ivcap, kmode, (const double*)nullptr, (unsigned int*)nullptr);
    else
        hipLaunchKernelGGL(mw::k_move_energy<false>, dim3(g.mwork_n), dim3(1024), iv_bytes, g.stream,
                           g.d_pos, g.d_ivect, g.d_nivect, g.d_listm, g.d_nn, g.d_mwork, g.d_mimol, g.d_mtrial, g.d_mperm,
                           g.d_meold, g.d_menew, g.d_mcnt, g.d_mdecl, g.N, g.ivcap, kmode, (const double*)nullptr, (unsigned int*)nullptr);
    HIPCHK(hipGetLastError());
    // the requests the fused routine declined (none on ice): plain routine, one wavefront each
    hipLaunchKernelGGL(mw::k_move_fallback, dim3(std::min(1024, (g.mn + 3) / 4)), dim3(256), 0, g.stream, g.d_pos, g.d_ivect, g.d_listm, g.d_nn,
                       g.d_mimol, g.d_mtrial, g.d_mperm, g.d_meold, g.d_menew, g.d_mcnt, g.d_mdecl, g.N, g.ivcap, kmode);
    HIPCHK(hipGetLastError());
    g.mmode = mode;
    return 0;
}

int mw_moves_launch(void)
{
    MW_LOCK;
    if (check_live()) return 1;
    return launch_moves(3);
}

int mw_step_launch(int first_ils, int count, int timer_slot)
{
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    const bool timed = timer_slot >= 0;
    if (timed) {
        if (timer_slot + 1 >= kTimerSlots) return fail("mw_step_launch: timer slot %d outside 0..%d", timer_slot, kTimerSlots - 2);
        for (int s = timer_slot; s <= timer_slot + 1; ++s)
            if (!g.ev[s][0]) { HIPCHK(hipEventCreate(&g.ev[s][0])); HIPCHK(hipEventCreate(&g.ev[s][1])); }
        HIPCHK(hipEventRecord(g.ev[timer_slot][0], g.stream));
    }
    // (the step's full-box pass leaves every molecule's moments behind when the step's move kernel will take the moment path)
    static const char* momenv = std::getenv("MW_MOVE_MOMENTS");
    const bool want_mom = g.mn > 0 && g.mlds && g.m_noself && (momenv ? momenv[0] != '0' : g.m_minreq >= 512);
    if (launch_model_energy(first_ils, count, want_mom, true)) return 1;
    if (timed) { HIPCHK(hipEventRecord(g.ev[timer_slot][1], g.stream)); HIPCHK(hipEventRecord(g.ev[timer_slot + 1][0], g.stream)); }
    if (launch_moves(3)) return 1;
    if (timed) HIPCHK(hipEventRecord(g.ev[timer_slot + 1][1], g.stream));
    return 0;
}

int mw_moves_fetch(double* e_old, double* e_new)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (g.mn > 0) {
        if (e_old) HIPCHK(hipMemcpyAsync(e_old, g.d_meold, sizeof(double) * g.mn, hipMemcpyDeviceToHost, g.stream));
        if (e_new) HIPCHK(hipMemcpyAsync(e_new, g.d_menew, sizeof(double) * g.mn, hipMemcpyDeviceToHost, g.stream));
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_moves_counts(long long out[4])
{
    MW_LOCK;
    if (check_live()) return 1;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (g.mn == 0) return 0;
    std::vector<unsigned int> c((size_t)g.mn * 4);
    unsigned long long tot[4] = {0, 0, 0, 0};
    std::vector<unsigned int> it((size_t)g.mtot_n * 4);
    HIPCHK(hipMemcpyAsync(c.data(), g.d_mcnt, sizeof(unsigned int) * 4 * g.mn, hipMemcpyDeviceToHost, g.stream));
    if (g.mtot_n) HIPCHK(hipMemcpyAsync(it.data(), g.d_mtot, sizeof(unsigned int) * 4 * g.mtot_n, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int k = 0; k < g.mtot_n; ++k) for (int q = 0; q < 4; ++q) tot[q] += it[4 * (size_t)k + q];
    for (int m = 0; m < g.mn; ++m) {
        if (g.mmode & 1) { out[0] += c[4 * (size_t)m]; out[1] += c[4 * (size_t)m + 1]; }
        if (g.mmode & 2) { out[2] += c[4 * (size_t)m + 2]; out[3] += c[4 * (size_t)m + 3]; }
    }
    if (g.mmode & 1) { out[0] += (long long)tot[0]; out[1] += (long long)tot[1]; }      // (the moment path's requests: summed on the device)
    if (g.mmode & 2) { out[2] += (long long)tot[2]; out[3] += (long long)tot[3]; }
    return 0;
}

int mw_local_energy_batch(int n, const int* ils, const int* imol, const double* trial_xyz, double* e_out)
{
    MW_LOCK;
    if (mw_moves_upload(n, ils, imol, trial_xyz)) return 1;
    if (launch_moves(trial_xyz ? 2 : 1)) return 1;
    return trial_xyz ? mw_moves_fetch(nullptr, e_out) : mw_moves_fetch(e_out, nullptr);
}

int mw_delta_energy_batch(int n, const int* ils, const int* imol, const double* trial_xyz, double* e_old, double* e_new)
{
    MW_LOCK;
    if (!trial_xyz) return fail("mw_delta_energy_batch: trial positions are required");
    if (mw_moves_upload(n, ils, imol, trial_xyz)) return 1;
    if (launch_moves(3)) return 1;
    return mw_moves_fetch(e_old, e_new);
}

int mw_set_model_energy(int ils, double e)
{
    MW_LOCK;
    if (check_live() || check_box(ils)) return 1;
    HIPCHK(hipMemcpyAsync(g.d_energy + (ils - 1), &e, sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_configure(int nlat, double beta, double max_trans, int nbins, int eta_interp, int start_bin, int end_bin,
                       double r_pos, double a_pos, double r_neg, double a_neg, double mu_lo, double mu_hi,
                       const double* weight, const double* mu_bin, const double* binwidth)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (nlat != 1 && nlat != 2) return fail("mw_sweep_configure: num_lattices = %d (1 or 2)", nlat);
    if (g.nbox % nlat) return fail("mw_sweep_configure: %d boxes do not split into walkers of %d lattices", g.nbox, nlat);
    if (nlat == 2) {
        if (nbins < 3 || !weight || !mu_bin || !binwidth) return fail("mw_sweep_configure: two lattices need the weight tables");
        if (start_bin < 1 || end_bin > nbins || start_bin >= end_bin) return fail("mw_sweep_configure: bins %d..%d outside 1..%d", start_bin, end_bin, nbins);
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    g.sp.beta = beta; g.sp.max_trans = max_trans;
    g.sp.r_pos = r_pos; g.sp.a_pos = a_pos; g.sp.r_neg = r_neg; g.sp.a_neg = a_neg; g.sp.mu_lo = mu_lo; g.sp.mu_hi = mu_hi;
    g.sp.nlat = nlat; g.sp.nbins = nbins; g.sp.eta_interp = eta_interp; g.sp.start_bin = start_bin; g.sp.end_bin = end_bin; g.sp.pad = 0;
    g.sp.record = 0; g.sp.samplerun = 1; g.sp.always_switch = 0; g.sp.npt = 0;
    g.sp.av_binwidth = 1.0; g.sp.wl_factor = 0.0; g.sp.log_unbiased_norm = 0.0; g.sp.pressure = 0.0;
    g.sp.transP = 2.0; g.sp.dv_max = 0.0;       // translations only until mw_sweep_moves says otherwise
    if (g.d_sw_mubin) {
        HIPCHK(hipFree(g.d_sw_mubin)); HIPCHK(hipFree(g.d_sw_binwidth));
        HIPCHK(hipFree(g.d_wweight)); HIPCHK(hipFree(g.d_whist)); HIPCHK(hipFree(g.d_wuhist));
        g.d_sw_mubin = nullptr;
    }
    const size_t nb = (size_t)(nbins > 0 ? nbins : 1);
    const size_t nw = (size_t)(g.nbox / nlat);
    HIPCHK(hipMalloc(&g.d_sw_mubin, nb * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_sw_binwidth, nb * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_wweight, nw * nb * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_whist, nw * nb * sizeof(double)));
    HIPCHK(hipMalloc(&g.d_wuhist, nw * nb * sizeof(double)));
    HIPCHK(hipMemset(g.d_wweight, 0, nw * nb * sizeof(double)));
    HIPCHK(hipMemset(g.d_whist, 0, nw * nb * sizeof(double)));
    HIPCHK(hipMemset(g.d_wuhist, 0, nw * nb * sizeof(double)));
    if (nlat == 2) {
        HIPCHK(hipMemcpy(g.d_sw_mubin, mu_bin, nb * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(g.d_sw_binwidth, binwidth, nb * sizeof(double), hipMemcpyHostToDevice));
        std::vector<double> all(nw * nb);
        for (size_t w = 0; w < nw; ++w) std::memcpy(&all[w * nb], weight, nb * sizeof(double));
        HIPCHK(hipMemcpy(g.d_wweight, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    g.nwalkers = g.nbox / nlat;
    if (!g.d_wls) {
        HIPCHK(hipMalloc(&g.d_wls, sizeof(int) * g.nbox));
        HIPCHK(hipMalloc(&g.d_wmu, sizeof(double) * g.nbox));
        HIPCHK(hipMalloc(&g.d_wacc, sizeof(unsigned long long) * g.nbox));
        HIPCHK(hipMalloc(&g.d_wswitch, sizeof(unsigned long long) * g.nbox));
        HIPCHK(hipMalloc(&g.d_wshift, sizeof(double) * g.nbox));
        HIPCHK(hipMalloc(&g.d_wvol, sizeof(unsigned long long) * 2 * g.nbox));
        HIPCHK(hipMalloc(&g.d_wflag, sizeof(int) * g.nbox));
        HIPCHK(hipMalloc(&g.d_wwin, sizeof(double) * 4 * g.nbox));
        HIPCHK(hipMalloc(&g.d_wfac, sizeof(double) * g.nbox));
        HIPCHK(hipMalloc(&g.d_wsum, sizeof(double) * g.nbox));
        HIPCHK(hipMalloc(&g.d_winflag, sizeof(int) * g.nbox));
        HIPCHK(hipMalloc(&g.d_wstep, sizeof(double) * 2 * g.nbox));
    }
    HIPCHK(hipMemset(g.d_wwin, 0, sizeof(double) * 4 * g.nbox));
    HIPCHK(hipMemset(g.d_wfac, 0, sizeof(double) * g.nbox));
    HIPCHK(hipMemset(g.d_wsum, 0, sizeof(double) * g.nbox));
    HIPCHK(hipMemset(g.d_winflag, 0, sizeof(int) * g.nbox));
    g.has_windows = false; g.has_steps = false;
    g.sp.dref = 0.0; g.sp.ref1 = g.sp.ref2 = 0.0; g.sp.minu = 0; g.sp.pad_minu = 0; g.sp.swetnam = 0; g.sp.dd = 0; g.sp.wl_alpha = 1.0; g.sp.orig_wl_factor = 0.0;
    g.sp.mu_min = mu_lo; g.sp.mu_max = mu_hi; g.sp.eq_cycles = 0; g.sp.in_window = 1;
    HIPCHK(hipMemset(g.d_wvol, 0, sizeof(unsigned long long) * 2 * g.nbox));
    HIPCHK(hipMemset(g.d_wflag, 0, sizeof(int) * g.nbox));
    HIPCHK(hipMemset(g.d_wswitch, 0, sizeof(unsigned long long) * g.nbox));
    HIPCHK(hipMemset(g.d_wshift, 0, sizeof(double) * g.nbox));
    std::vector<int> one((size_t)g.nbox, 1);
    HIPCHK(hipMemcpy(g.d_wls, one.data(), sizeof(int) * g.nbox, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(g.d_wmu, 0, sizeof(double) * g.nbox));
    HIPCHK(hipMemset(g.d_wacc, 0, sizeof(unsigned long long) * g.nbox));
    g.sweep_ready = true;
    return 0;
}

static int check_walker(int first, int count)
{
    if (!g.sweep_ready) return fail("mw_sweep: call mw_sweep_configure first");
    if (first < 1 || count < 1 || first + count - 1 > g.nwalkers)
        return fail("mw_sweep: walker range %d..%d outside 1..%d", first, first + count - 1, g.nwalkers);
    return 0;
}

int mw_sweep_set_state(int walker, int ls, double ls_mu)
{
    MW_LOCK;
    if (check_live() || check_walker(walker, 1)) return 1;
    if (ls < 1 || ls > g.sp.nlat) return fail("mw_sweep_set_state: active lattice %d outside 1..%d", ls, g.sp.nlat);
    HIPCHK(hipMemcpyAsync(g.d_wls + (walker - 1), &ls, sizeof(int), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_wmu + (walker - 1), &ls_mu, sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_set_states_range(int first_walker, int count, const int* ls, const double* ls_mu)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    if (!ls || !ls_mu) return fail("mw_sweep_set_states_range: null pointer");
    for (int k = 0; k < count; ++k)
        if (ls[k] < 1 || ls[k] > g.sp.nlat) return fail("mw_sweep_set_states_range: active lattice %d of walker %d outside 1..%d", ls[k], first_walker + k, g.sp.nlat);
    HIPCHK(hipMemcpyAsync(g.d_wls + (first_walker - 1), ls, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(g.d_wmu + (first_walker - 1), ls_mu, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_get_state(int walker, int* ls, double* ls_mu, double* model_energy, long long* accepted)
{
    MW_LOCK;
    if (check_live() || check_walker(walker, 1)) return 1;
    int l = 0; double mu = 0.0; unsigned long long a = 0; double e[2] = {0.0, 0.0};
    HIPCHK(hipMemcpyAsync(&l, g.d_wls + (walker - 1), sizeof(int), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(&mu, g.d_wmu + (walker - 1), sizeof(double), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(&a, g.d_wacc + (walker - 1), sizeof a, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(e, g.d_energy + (size_t)(walker - 1) * g.sp.nlat, sizeof(double) * g.sp.nlat, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    if (ls) *ls = l;
    if (ls_mu) *ls_mu = mu;
    if (accepted) *accepted = (long long)a;
    if (model_energy) { model_energy[0] = e[0]; if (g.sp.nlat == 2) model_energy[1] = e[1]; }
    return 0;
}

int mw_sweep_options(int record, int samplerun, int always_switch, int npt,
                     double av_binwidth, double wl_factor, double log_unbiased_norm, double pressure)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (!g.sweep_ready) return fail("mw_sweep_options: call mw_sweep_configure first");
    if ((record || always_switch) && g.sp.nlat != 2) return fail("mw_sweep_options: histograms and lattice switches need two lattices");
    g.sp.record = record ? 1 : 0; g.sp.samplerun = samplerun ? 1 : 0; g.sp.always_switch = always_switch ? 1 : 0; g.sp.npt = npt ? 1 : 0;
    g.sp.av_binwidth = av_binwidth; g.sp.wl_factor = wl_factor; g.sp.log_unbiased_norm = log_unbiased_norm; g.sp.pressure = pressure;
    if (g.sp.nlat == 2 && !g.sp.swetnam && !g.sp.dd) {           // one increment for every walker ('mw'); per-walker values: mw_sweep_set_factors
        std::vector<double> f((size_t)g.nwalkers, wl_factor);
        HIPCHK(hipMemcpyAsync(g.d_wfac, f.data(), sizeof(double) * g.nwalkers, hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int mw_sweep_leshift(double ref_enthalpy_1, double ref_enthalpy_2)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (!g.sweep_ready) return fail("mw_sweep_leshift: call mw_sweep_configure first");
    g.sp.dref = ref_enthalpy_1 - ref_enthalpy_2;
    g.sp.ref1 = ref_enthalpy_1; g.sp.ref2 = ref_enthalpy_2;
    return 0;
}

int mw_sweep_minu(int on)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (!g.sweep_ready) return fail("mw_sweep_minu: call mw_sweep_configure first");
    if (on && g.sp.nlat != 2) return fail("mw_sweep_minu: needs two lattices per walker");
    g.sp.minu = on ? 1 : 0;
    return 0;
}

int mw_sweep_swetnam(int on, double wl_alpha, double orig_wl_factor, double mu_min, double mu_max)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (!g.sweep_ready) return fail("mw_sweep_swetnam: call mw_sweep_configure first");
    g.sp.swetnam = on ? 1 : 0; g.sp.wl_alpha = wl_alpha; g.sp.orig_wl_factor = orig_wl_factor;
    g.sp.mu_min = mu_min; g.sp.mu_max = mu_max;
    return 0;
}

int mw_sweep_dd(int on, int eq_mc_cycles)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (!g.sweep_ready) return fail("mw_sweep_dd: call mw_sweep_configure first");
    g.sp.dd = on ? 1 : 0; g.sp.eq_cycles = eq_mc_cycles;
    return 0;
}

int mw_sweep_windows(int first_walker, int count, const int* start_bin, const int* end_bin, const double* mu_lo, const double* mu_hi)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    if (!start_bin || !end_bin || !mu_lo || !mu_hi) { g.has_windows = false; return 0; }
    std::vector<double> w((size_t)count * 4);
    for (int k = 0; k < count; ++k) {
        if (start_bin[k] < 1 || end_bin[k] > g.sp.nbins || start_bin[k] >= end_bin[k])
            return fail("mw_sweep_windows: walker %d has bins %d..%d outside 1..%d", first_walker + k, start_bin[k], end_bin[k], g.sp.nbins);
        w[4 * (size_t)k] = start_bin[k]; w[4 * (size_t)k + 1] = end_bin[k]; w[4 * (size_t)k + 2] = mu_lo[k]; w[4 * (size_t)k + 3] = mu_hi[k];
    }
    if (!g.has_windows) {        // walkers outside the range given keep the window of mw_sweep_configure
        std::vector<double> all((size_t)g.nwalkers * 4);
        for (int k = 0; k < g.nwalkers; ++k) { all[4 * (size_t)k] = g.sp.start_bin; all[4 * (size_t)k + 1] = g.sp.end_bin; all[4 * (size_t)k + 2] = g.sp.mu_lo; all[4 * (size_t)k + 3] = g.sp.mu_hi; }
        HIPCHK(hipMemcpyAsync(g.d_wwin, all.data(), sizeof(double) * all.size(), hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    HIPCHK(hipMemcpyAsync(g.d_wwin + 4 * (size_t)(first_walker - 1), w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    g.has_windows = true;
    return 0;
}

int mw_sweep_set_factors(int first_walker, int count, const double* wl_factor, const double* sumhist)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    if (wl_factor) HIPCHK(hipMemcpyAsync(g.d_wfac + (first_walker - 1), wl_factor, sizeof(double) * count, hipMemcpyHostToDevice, g.stream));
    if (sumhist) HIPCHK(hipMemcpyAsync(g.d_wsum + (first_walker - 1), sumhist, sizeof(double) * count, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_get_factors(int first_walker, int count, double* wl_factor, double* sumhist, int* in_window)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    if (wl_factor) HIPCHK(hipMemcpyAsync(wl_factor, g.d_wfac + (first_walker - 1), sizeof(double) * count, hipMemcpyDeviceToHost, g.stream));
    if (sumhist) HIPCHK(hipMemcpyAsync(sumhist, g.d_wsum + (first_walker - 1), sizeof(double) * count, hipMemcpyDeviceToHost, g.stream));
    if (in_window) HIPCHK(hipMemcpyAsync(in_window, g.d_winflag + (first_walker - 1), sizeof(int) * count, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_steps(int first_walker, int count, const double* max_trans_bohr, const double* dv_max_bohr)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    if (!max_trans_bohr || !dv_max_bohr) { g.has_steps = false; return 0; }
    if (!g.has_steps) {          // walkers outside the range given keep the common values
        std::vector<double> all((size_t)g.nwalkers * 2);
        for (int k = 0; k < g.nwalkers; ++k) { all[2 * (size_t)k] = g.sp.max_trans; all[2 * (size_t)k + 1] = g.sp.dv_max; }
        HIPCHK(hipMemcpyAsync(g.d_wstep, all.data(), sizeof(double) * all.size(), hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    std::vector<double> w((size_t)count * 2);
    for (int k = 0; k < count; ++k) {
        if (!(max_trans_bohr[k] > 0.0) || !(dv_max_bohr[k] >= 0.0)) return fail("mw_sweep_steps: walker %d has step sizes %g, %g", first_walker + k, max_trans_bohr[k], dv_max_bohr[k]);
        w[2 * (size_t)k] = max_trans_bohr[k]; w[2 * (size_t)k + 1] = dv_max_bohr[k];
    }
    HIPCHK(hipMemcpyAsync(g.d_wstep + 2 * (size_t)(first_walker - 1), w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    g.has_steps = true;
    return 0;
}

int mw_sweep_get_counters(int first_walker, int count, long long* accepted, long long* vol_attempted, long long* vol_accepted)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    std::vector<unsigned long long> a((size_t)count), v((size_t)count * 2);
    HIPCHK(hipMemcpyAsync(a.data(), g.d_wacc + (first_walker - 1), sizeof(unsigned long long) * count, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(v.data(), g.d_wvol + 2 * (size_t)(first_walker - 1), sizeof(unsigned long long) * 2 * count, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int k = 0; k < count; ++k) {
        if (accepted) accepted[k] = (long long)a[(size_t)k];
        if (vol_attempted) vol_attempted[k] = (long long)v[2 * (size_t)k];
        if (vol_accepted) vol_accepted[k] = (long long)v[2 * (size_t)k + 1];
    }
    return 0;
}

int mw_sweep_moves(double transP, double dv_max_bohr)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (!g.sweep_ready) return fail("mw_sweep_moves: call mw_sweep_configure first");
    if (!(transP > 0.0)) return fail("mw_sweep_moves: transP = %g must be positive", transP);
    g.sp.transP = transP; g.sp.dv_max = dv_max_bohr;
    if (transP < 1.0) {
        // volume moves shrink cells on the device: keep room for one more shell of images along any one axis
        // (a move that still outgrows the table is rejected and flagged, mw_sweep_check_flags)
        int need = g.ivcap;
        for (int b = 0; b < g.nbox; ++b) {
            const int* im = g.h_grid[(size_t)b].im;
            if (g.h_nivect[(size_t)b] < 1) continue;
            const int w0 = 2 * im[0] + 1, w1 = 2 * im[1] + 1, w2 = 2 * im[2] + 1;
            need = std::max(need, std::max((w0 + 2) * w1 * w2, std::max(w0 * (w1 + 2) * w2, w0 * w1 * (w2 + 2))));
        }
        if (need > MW_MAX_IVECT) need = MW_MAX_IVECT;
        if (need > g.ivcap && grow_ivcap(need)) return 1;
    }
    return 0;
}

int mw_sweep_check_flags(int first_walker, int count)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    std::vector<int> flags((size_t)count, 0);
    HIPCHK(hipMemcpyAsync(flags.data(), g.d_wflag + (first_walker - 1), sizeof(int) * count, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int w = 0; w < count; ++w) {
        if (flags[(size_t)w] & 1)
            return fail("mw_sweep: a volume move of walker %d shrank a cell below what %d image vectors cover (the move was rejected)",
                        first_walker + w, g.ivcap);
        if (flags[(size_t)w] & 2)
            return fail("Error : Not all walkers have reached their designated window after %d MC cycles (walker %d)",
                        g.sp.eq_cycles, first_walker + w);
    }
    return 0;
}

int mw_sweep_get_volume_moves(int walker, long long* attempted, long long* accepted)
{
    MW_LOCK;
    if (check_live() || check_walker(walker, 1)) return 1;
    unsigned long long v[2];
    int flag = 0;
    HIPCHK(hipMemcpyAsync(v, g.d_wvol + 2 * (size_t)(walker - 1), sizeof v, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(&flag, g.d_wflag + (walker - 1), sizeof(int), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    if (attempted) *attempted = (long long)v[0];
    if (accepted) *accepted = (long long)v[1];
    if (flag & 1) return fail("mw_sweep: a volume move of walker %d shrank a cell below what %d image vectors cover", walker, g.ivcap);
    return 0;
}

// After volume moves on the device the host mirrors of the cells (image vectors, neighbour-grid descriptors)
// are stale: read the cells back and rebuild them exactly as mw_set_cell does.  Call before rebuilding lists.
int mw_sweep_sync_cells(int first_ils, int count, double* h_out)
{
    MW_LOCK;
    if (check_live() || check_range(first_ils, count)) return 1;
    std::vector<double> h((size_t)count * 9);
    std::vector<int> flags((size_t)g.nbox, 0);
    HIPCHK(hipMemcpyAsync(h.data(), g.d_hmat + 9 * (size_t)(first_ils - 1), h.size() * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    if (g.d_wflag) HIPCHK(hipMemcpyAsync(flags.data(), g.d_wflag, sizeof(int) * g.nbox, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int w = 0; w < g.nbox; ++w)
        if (flags[(size_t)w] & 1) return fail("mw_sweep: a volume move of walker %d shrank a cell below what %d image vectors cover", w + 1, g.ivcap);
    // host mirrors for every box, then four bulk uploads (the device already holds these image vectors: same arithmetic).
    // A farm calls this before every list rebuild for thousands of boxes: the boxes are shared out among the host's cores
    // (one thread did 16 384 boxes in 3.5 ms, ten times per hundred cycles -- an eighth of an NPT farm's wall time).
    std::atomic<int> bad_box{-1}, bad_n{0};
    auto mirror = [&](int lo, int hi) {
        std::vector<double> iv;
        for (int b = lo; b < hi; ++b) {
            const int box = first_ils - 1 + b;
            int imv[3] = {1, 1, 1};
            const int n = host_ivects(&h[(size_t)b * 9], iv, imv);
            if (n < 0 || n > g.ivcap) { int none = -1; if (bad_box.compare_exchange_strong(none, box)) bad_n = n; return; }
            std::memcpy(&g.h_ivect[(size_t)box * g.ivcap * 3], iv.data(), iv.size() * sizeof(double));
            g.h_nivect[box] = n;
            g.h_grid[box] = make_grid(&h[(size_t)b * 9], imv, g.cstride);
            g.h_usegrid[box] = (!g.force_brute && g.h_grid[box].nc[0] > 0) ? 1 : 0;
            if (!g.h_usegrid[box]) g.h_grid[box].nc[0] = 0;
            if (h_out) std::memcpy(h_out + (size_t)b * 9, &h[(size_t)b * 9], 9 * sizeof(double));
        }
    };
    const int nthr = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, count / 512}));
    if (nthr == 1) mirror(0, count);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t)
            pool.emplace_back(mirror, (int)((long long)count * t / nthr), (int)((long long)count * (t + 1) / nthr));
        for (auto& th : pool) th.join();
    }
    if (bad_box.load() >= 0)
        return fail("mw_sweep_sync_cells: box %d needs %d image vectors (capacity %d)", bad_box.load() + 1, bad_n.load(), g.ivcap);
    // What goes back to the device is what only the host works out: which boxes take the cell-grid list builder, and their
    // grid descriptors.  The image vectors do NOT: the volume moves rebuilt them on the device in the reference's order and
    // arithmetic (dev_compute_ivects), so the device's tables already equal the mirrors just computed -- 19 MB per call for
    // 16 384 boxes that used to be uploaded regardless, 3 ms of an idle GPU before every list rebuild of an NPT farm.
    // MW_SYNC_CELLS_VERIFY=1 reads the device's tables back instead and compares them bit for bit (tests).
    const size_t b0 = (size_t)(first_ils - 1);
    bool any_grid = g.grid_on_device;
    for (int b = 0; b < count; ++b) any_grid = any_grid || g.h_usegrid[b0 + b] != 0;
    if (any_grid) {
        HIPCHK(hipMemcpyAsync(g.d_grid + b0, &g.h_grid[b0], sizeof(mw::GridDesc) * count, hipMemcpyHostToDevice, g.stream));
        g.grid_on_device = true;
    }
    HIPCHK(hipMemcpyAsync(g.d_usegrid + b0, &g.h_usegrid[b0], sizeof(int) * count, hipMemcpyHostToDevice, g.stream));
    static const bool verify = [] { const char* e = std::getenv("MW_SYNC_CELLS_VERIFY"); return e && std::atoi(e) != 0; }();
    if (verify) {
        std::vector<double> div((size_t)count * g.ivcap * 3);
        std::vector<int> dn((size_t)count);
        HIPCHK(hipMemcpyAsync(div.data(), g.d_ivect + b0 * g.ivcap * 3, div.size() * sizeof(double), hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipMemcpyAsync(dn.data(), g.d_nivect + b0, dn.size() * sizeof(int), hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        for (int b = 0; b < count; ++b) {
            const int n = g.h_nivect[b0 + b];
            if (dn[(size_t)b] != n)
                return fail("mw_sweep_sync_cells: box %d has %d image vectors on the device, %d by the host's arithmetic", (int)b0 + b + 1, dn[(size_t)b], n);
            if (std::memcmp(&div[(size_t)b * g.ivcap * 3], &g.h_ivect[(b0 + b) * g.ivcap * 3], sizeof(double) * 3 * n) != 0)
                return fail("mw_sweep_sync_cells: the device's image vectors of box %d differ from the host's", (int)b0 + b + 1);
        }
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

static int tables_io(int walker, double* weight, double* hist, double* uhist, bool put)
{
    if (check_live() || check_walker(walker, 1)) return 1;
    const size_t nb = (size_t)g.sp.nbins, off = (size_t)(walker - 1) * nb;
    double* dev[3] = {g.d_wweight + off, g.d_whist + off, g.d_wuhist + off};
    double* host[3] = {weight, hist, uhist};
    for (int t = 0; t < 3; ++t) {
        if (!host[t]) continue;
        if (put) HIPCHK(hipMemcpyAsync(dev[t], host[t], nb * sizeof(double), hipMemcpyHostToDevice, g.stream));
        else     HIPCHK(hipMemcpyAsync(host[t], dev[t], nb * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_get_tables(int walker, double* weight, double* histogram, double* unbiased_hist)
{
    MW_LOCK;
    return tables_io(walker, weight, histogram, unbiased_hist, false);
}

int mw_sweep_set_tables(int walker, const double* weight, const double* histogram, const double* unbiased_hist)
{
    MW_LOCK;
    return tables_io(walker, const_cast<double*>(weight), const_cast<double*>(histogram), const_cast<double*>(unbiased_hist), true);
}

int mw_sweep_get_tables_range(int first_walker, int count, double* weight, double* histogram, double* unbiased_hist)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    const size_t nb = (size_t)g.sp.nbins, off = (size_t)(first_walker - 1) * nb, bytes = (size_t)count * nb * sizeof(double);
    if (weight) HIPCHK(hipMemcpyAsync(weight, g.d_wweight + off, bytes, hipMemcpyDeviceToHost, g.stream));
    if (histogram) HIPCHK(hipMemcpyAsync(histogram, g.d_whist + off, bytes, hipMemcpyDeviceToHost, g.stream));
    if (unbiased_hist) HIPCHK(hipMemcpyAsync(unbiased_hist, g.d_wuhist + off, bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_set_tables_range(int first_walker, int count, const double* weight, const double* histogram, const double* unbiased_hist)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    const size_t nb = (size_t)g.sp.nbins, off = (size_t)(first_walker - 1) * nb, bytes = (size_t)count * nb * sizeof(double);
    if (weight) HIPCHK(hipMemcpyAsync(g.d_wweight + off, weight, bytes, hipMemcpyHostToDevice, g.stream));
    if (histogram) HIPCHK(hipMemcpyAsync(g.d_whist + off, histogram, bytes, hipMemcpyHostToDevice, g.stream));
    if (unbiased_hist) HIPCHK(hipMemcpyAsync(g.d_wuhist + off, unbiased_hist, bytes, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_get_shifts_range(int first_walker, int count, double* shifts, int reset)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    if (shifts) HIPCHK(hipMemcpyAsync(shifts, g.d_wshift + (first_walker - 1), sizeof(double) * count, hipMemcpyDeviceToHost, g.stream));
    if (reset) HIPCHK(hipMemsetAsync(g.d_wshift + (first_walker - 1), 0, sizeof(double) * count, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

// sum over the walkers of (table + shift - last), per table and bin; NULL last_* / sum_* skips a table
int mw_sweep_reduce_tables(int first_walker, int count, const double* last_w, const double* last_h, const double* last_u,
                           double* sum_w, double* sum_h, double* sum_u, int use_shifts, int reset_shifts)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    const int nb = g.sp.nbins, nchunks = (count + mw::kTableChunk - 1) / mw::kTableChunk;
    const size_t need = (size_t)nb * (6 + (size_t)nchunks);
    if (need > g.tabscratch_n) {
        HIPCHK(hipStreamSynchronize(g.stream));
        if (g.d_tabscratch) HIPCHK(hipFree(g.d_tabscratch));
        HIPCHK(hipMalloc(&g.d_tabscratch, need * sizeof(double)));
        g.tabscratch_n = need;
    }
    const double* last[3] = {last_w, last_h, last_u};
    double* out[3] = {sum_w, sum_h, sum_u};
    const double* tabs[3] = {g.d_wweight, g.d_whist, g.d_wuhist};
    double* d_last = g.d_tabscratch, *d_out = g.d_tabscratch + 3 * (size_t)nb, *d_part = g.d_tabscratch + 6 * (size_t)nb;
    for (int t = 0; t < 3; ++t) {
        if (!last[t] || !out[t]) continue;
        HIPCHK(hipMemcpyAsync(d_last + (size_t)t * nb, last[t], sizeof(double) * nb, hipMemcpyHostToDevice, g.stream));
        hipLaunchKernelGGL(mw::k_tables_partial, dim3(nchunks), dim3(128), 0, g.stream, tabs[t],
                           (t == 0 && use_shifts) ? (const double*)g.d_wshift : (const double*)nullptr,
                           d_last + (size_t)t * nb, d_part, nb, first_walker - 1, count);
        hipLaunchKernelGGL(mw::k_tables_final, dim3(1), dim3(128), 0, g.stream, d_part, d_out + (size_t)t * nb, nb, nchunks);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out[t], d_out + (size_t)t * nb, sizeof(double) * nb, hipMemcpyDeviceToHost, g.stream));
    }
    if (reset_shifts) HIPCHK(hipMemsetAsync(g.d_wshift + (first_walker - 1), 0, sizeof(double) * count, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

// the same row for every walker of the range; NULL skips a table
int mw_sweep_broadcast_tables(int first_walker, int count, const double* weight, const double* histogram, const double* unbiased_hist)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    const int nb = g.sp.nbins;
    if ((size_t)nb * 6 > g.tabscratch_n) {
        HIPCHK(hipStreamSynchronize(g.stream));
        if (g.d_tabscratch) HIPCHK(hipFree(g.d_tabscratch));
        g.tabscratch_n = (size_t)nb * 8;
        HIPCHK(hipMalloc(&g.d_tabscratch, g.tabscratch_n * sizeof(double)));
    }
    const double* rows[3] = {weight, histogram, unbiased_hist};
    double* tabs[3] = {g.d_wweight, g.d_whist, g.d_wuhist};
    for (int t = 0; t < 3; ++t) {
        if (!rows[t]) continue;
        double* d_row = g.d_tabscratch + (size_t)t * nb;
        HIPCHK(hipMemcpyAsync(d_row, rows[t], sizeof(double) * nb, hipMemcpyHostToDevice, g.stream));
        hipLaunchKernelGGL(mw::k_tables_broadcast, dim3(count), dim3(128), 0, g.stream, tabs[t], (const double*)d_row, nb, first_walker - 1);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_sweep_get_switches(int walker, long long* switches)
{
    MW_LOCK;
    if (check_live() || check_walker(walker, 1)) return 1;
    unsigned long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, g.d_wswitch + (walker - 1), sizeof v, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    *switches = (long long)v;
    return 0;
}

int mw_sweep_lds_bytes(int nlat, int nwater, int nbins, int row_stride, int volume_moves, int samplerun, int image_capacity)
{
    if (nlat < 1 || nlat > 2 || nwater < 1 || nwater > 64 || nbins < 0 || row_stride < 2 || row_stride > 32 ||
        image_capacity < 0 || image_capacity > MW_MAX_IVECT) return -1;
    // image vectors per box: what the engine starts with (mw_init); with volume moves, room for one more shell of images
    // along one axis (mw_sweep_moves) -- 45 -> 48 for the 27-image cells of the reference's examples
    const int ivcap = image_capacity > 0 ? image_capacity : (volume_moves ? 48 : 32);
    return (int)mw::sweep_lds(nlat, nlat, ivcap, nwater, nbins, true, true, row_stride, volume_moves != 0, samplerun != 0).total;
}

int mw_sweep_translation_launch(int first_walker, int count, int nmoves, unsigned long long seed, unsigned long long move0, int want_log)
{
    MW_LOCK;
    if (check_live() || check_walker(first_walker, count)) return 1;
    if (nmoves < 0) return fail("mw_sweep_translation: nmoves = %d", nmoves);
    if (nmoves == 0) return 0;
    double* dlog = nullptr;
    if (want_log) {
        const size_t need = (size_t)count * nmoves * 8;
        if (need > g.swlog_cap) {
            HIPCHK(hipStreamSynchronize(g.stream));
            if (g.d_swlog) HIPCHK(hipFree(g.d_swlog));
            HIPCHK(hipMalloc(&g.d_swlog, need * sizeof(double)));
            g.swlog_cap = need;
        }
        dlog = g.d_swlog;
    }
    const int L = g.sp.nlat;
    const bool withvol = g.sp.transP < 1.0;          // volume moves: the build that carries mc_volume
    // Residency of a walker's data in LDS.  Small systems (the reference's own 48-molecule cells): positions, and -- when an
    // entry (j, image) fits 16 bits (N <= 64) and no row is longer than 32 -- list rows and row lengths too, so that nothing
    // in the move loop waits on global memory.  Eight walkers per CU (16 wavefronts of <= 128 VGPRs) want <= 20 KiB each.
    const size_t pos_bytes = (size_t)L * g.N * 3 * sizeof(double);
    const bool ldspos = pos_bytes <= 16 * 1024;
    bool ldslist = false;
    int rstride = 32;
    if (ldspos && g.N <= 64) {
        if (g.nnmax_version != g.list_version) {     // once per list rebuild: the longest row of ANY box
            std::vector<int> st((size_t)g.nbox * 2);
            HIPCHK(hipMemcpyAsync(st.data(), g.d_stats, st.size() * sizeof(int), hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipStreamSynchronize(g.stream));
            int mx = 0;
            for (size_t b = 0; b < st.size() / 2; ++b) mx = std::max(mx, st[2 * b + 1]);
            g.nnmax_cached = mx;                     // stats = {min nn, max nn} of the last list build of each box
            g.nnmax_version = g.list_version;
        }
        if (g.nnmax_cached <= 32) {                  // rows as short as the lists allow (entries are read one at a time: any even
            rstride = std::max(4, (g.nnmax_cached + 1) & ~1);      // stride will do): LDS per walker sets the occupancy -- a replica farm's
            ldslist = mw::sweep_lds(L, L, g.ivcap, g.N, g.sp.nbins, true, true, rstride, withvol, g.sp.samplerun != 0).total <= 24 * 1024;
        }                                            // longest row of 16 384 boxes grows from 22 to 26 entries as the walkers spread out
                                                     // (at a stride of 28 a walker went over 20 KiB: seven per CU instead of eight, -12 %)
    }
    // Look-ahead: as many moves at once as it takes to put ~4 wavefronts on every SIMD, at most 4; MW_SWEEP_AHEAD=1|2|4 overrides.
    // For walkers in global memory, and for walkers entirely in LDS (the reference's 48-molecule cells: a move reads most of such
    // a box, so any ACCEPTED move ends the round -- but nine moves in ten are rejected, and a handful of walkers, which is how the
    // reference itself runs, leaves the chip to their chains), and for the sizes in between.
    int spec = 1;
    {
        // ... as long as every walker of the launch still has a place on the chip: a compute unit holds 16 wavefronts of <= 128
        // VGPRs (12 of the one build that needs more), i.e. 16 / (lattices x look-ahead) workgroups.  Measured on 48-molecule pairs
        // (tools/sweep_measurements.py n48wl_<walkers> / n48npt_<walkers>): 4 ahead wins up to 512 walkers (256 with volume moves),
        // 2 ahead up to 1024 (768), and past that look-ahead only takes places away from other walkers.
        auto all_resident = [&](int ahead) {
            const int per_cu = ((L == 2 && ahead > 1) ? 12 : 16) / (L * ahead);   // (two lattices with look-ahead: the builds of <= 168 VGPRs)
            return (long long)count <= (long long)g.cu * per_cu;
        };
        spec = all_resident(4) ? 4 : (all_resident(2) ? 2 : 1);
        // eight in flight for one-lattice walkers in global memory (large boxes: consecutive moves seldom touch the same molecules)
        const bool has8 = L == 1 && !ldspos;
        if (has8 && all_resident(8)) spec = 8;
        // six for two-lattice walkers entirely in LDS, while there is a CU for each (a round ends with its first accepted move: 3.1 moves
        // per round of four at 16 % acceptance, 4.0 per round of six)
        const bool has6 = L == 2 && ldslist;
        if (has6 && all_resident(6)) spec = 6;
        if (const char* e = getenv("MW_SWEEP_AHEAD")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || (v == 8 && has8) || (v == 6 && has6)) spec = v; }
        if (want_log) spec = std::min(spec, g.sweep_log_ahead);
    }
    const mw::SweepLds lay = mw::sweep_lds(L, L * spec, g.ivcap, g.N, g.sp.nbins, ldspos, ldslist, rstride, withvol, g.sp.samplerun != 0, spec);
    const size_t static_lds = 1536;                  // cells and their backups, hand-over words, the walker's control block (generous bound)
    if (lay.total + static_lds > (size_t)160 * 1024 - 8 * 1024)
        return fail("mw_sweep: %u bytes of LDS per walker (image vectors %u, positions %u, list rows %u) exceed what a workgroup may have",
                    lay.total, lay.pos - lay.iv, lay.tab - lay.pos, lay.nn - lay.row);
    const void* kern = sweep_kernel(L, ldslist ? 2 : (ldspos ? 1 : 0), withvol, spec);
    const double* wwin = g.has_windows ? (const double*)g.d_wwin : (const double*)nullptr;
    const double* wstep = g.has_steps ? (const double*)g.d_wstep : (const double*)nullptr;
    int w0 = first_walker - 1;
    // the moment path of walkers entirely in LDS (mw_sweep.hip.h): 2 x L x N x kMomStride doubles of scratch per walker of the launch
    // (MW_SWEEP_MOMENTS=0: the row-scanning evaluation instead)
    double* wmom = nullptr;
    {
        const char* e = getenv("MW_SWEEP_MOMENTS");
        const int box_first = (first_walker - 1) * L + 1, nboxes = count * L;          // 1-based
        // (for launches that fill the chip -- four lattices per compute unit and up: 4096-molecule boxes x 512 / 1024 / 2048 walkers
        //  -12 / +4 / +25 %, 2048 x 1536 pairs +22 %; fewer walkers run their chains with look-ahead, where every moment is a global
        //  round trip on a chain's critical path.
        //  By the NUMBER of walkers, not by the look-ahead chosen for them: a launch's chain must not depend on its look-ahead.
        //  MW_SWEEP_MOMENTS=2 forces the path -- the tests', to hold it to the oracle and to itself across look-aheads on a few walkers)
        if (!ldslist && !withvol && !(e && e[0] == '0') && model_geo(nboxes).lds && g.N >= 128 && (nboxes >= 4 * g.cu || (e && e[0] == '2'))) {
            // walkers in global memory, translations only: the engine's own moments, made by the full-box kernel where the driver's
            // earlier launches have not kept them (its `MOMOUT` build: boxes that fit LDS), current afterwards for as long as nothing
            // else writes positions or cells (swm_first / swm_count)
            const bool current = g.d_mom && g.swm_count > 0 && g.swm_first <= box_first && box_first + nboxes <= g.swm_first + g.swm_count;
            if (!current) {
                if (launch_model_energy(box_first, nboxes, true, false)) return 1;
                g.mom_count = 0;                                   // (about to change under the batch kernels' feet)
            }
            g.swm_first = current ? g.swm_first : box_first; g.swm_count = current ? g.swm_count : nboxes;
            wmom = g.d_mom + (size_t)(box_first - 1) * g.N * mw::kMomStride;
        } else {
            g.swm_count = 0;                                       // (this launch moves molecules without keeping d_mom)
        }
        if (ldslist && !(e && e[0] == '0')) {
            const size_t need = (size_t)count * 2 * L * g.N * mw::kMomStride;
            if (need > g.wmom_cap) {
                HIPCHK(hipStreamSynchronize(g.stream));
                if (g.d_wmom) HIPCHK(hipFree(g.d_wmom));
                g.d_wmom = nullptr; g.wmom_cap = 0;
                HIPCHK(hipMalloc(&g.d_wmom, need * sizeof(double)));
                g.wmom_cap = need;
            }
            wmom = g.d_wmom;
        }
    }
    void* args[] = {&g.d_pos, &g.d_hmat, &g.d_ivect, &g.d_nivect, &g.d_listm, &g.d_list, &g.d_nn, &g.d_order, &g.d_nns, &g.d_cmax,
                    &g.d_energy, &g.d_wls, &g.d_wmu, &g.d_wacc, &g.d_wswitch, &g.d_wshift, &g.sp, &g.d_wweight, &g.d_whist, &g.d_wuhist,
                    &g.d_sw_mubin, &g.d_sw_binwidth, &g.d_volume, &g.d_wvol, &g.d_wflag, &g.N, &g.S, &g.ivcap, &nmoves, &seed, &move0,
                    &w0, &dlog, &rstride, &wwin, &g.d_wfac, &g.d_wsum, &g.d_winflag, &wstep, &wmom};
    HIPCHK(hipLaunchKernel(kern, dim3(count), dim3(64 * L * spec), args, lay.total, g.stream));
    HIPCHK(hipGetLastError());
    g.last_sweep[0] = L; g.last_sweep[1] = spec; g.last_sweep[2] = ldslist ? 2 : (ldspos ? 1 : 0); g.last_sweep[3] = withvol ? 1 : 0;
    g.last_sweep[4] = (int)lay.total; g.last_sweep[5] = ldslist ? rstride : 0;
    return 0;
}

int mw_sweep_last_launch(int* nlat, int* ahead, int* residency, int* volume_moves, int* lds_bytes, int* row_stride)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (g.last_sweep[0] == 0) return fail("mw_sweep_last_launch: no launch of the driver yet");
    int* out[6] = {nlat, ahead, residency, volume_moves, lds_bytes, row_stride};
    for (int k = 0; k < 6; ++k) if (out[k]) *out[k] = g.last_sweep[k];
    return 0;
}

int mw_sweep_translation(int first_walker, int count, int nmoves, unsigned long long seed, unsigned long long move0, double* log)
{
    MW_LOCK;
    if (mw_sweep_translation_launch(first_walker, count, nmoves, seed, move0, log != nullptr)) return 1;
    if (log && nmoves > 0)
        HIPCHK(hipMemcpyAsync(log, g.d_swlog, sizeof(double) * (size_t)count * nmoves * 8, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

#ifdef MW_SWEEP_STAMPS
// Diagnostic build only (tools/sweep_stamps.py): the cycle sums of walker 0's first wavefront; reset != 0 zeroes them afterwards.
int mw_debug_sweep_stamps(unsigned long long* out, int n, int reset)
{
    MW_LOCK;
    if (check_live()) return 1;
    unsigned long long st[48];
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpyFromSymbol(st, HIP_SYMBOL(mw::g_sweep_stamps), sizeof st));
    for (int k = 0; k < n && k < 48; ++k) out[k] = st[k];
    if (reset) { memset(st, 0, sizeof st); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mw::g_sweep_stamps), st, sizeof st)); }
    return 0;
}
#endif

int mw_sync(void)
{
    MW_LOCK;
    if (check_live()) return 1;
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int mw_timer_start(int slot)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (slot < 0 || slot >= kTimerSlots) return fail("mw_timer: slot %d outside 0..%d", slot, kTimerSlots - 1);
    if (!g.ev[slot][0]) { HIPCHK(hipEventCreate(&g.ev[slot][0])); HIPCHK(hipEventCreate(&g.ev[slot][1])); }
    HIPCHK(hipEventRecord(g.ev[slot][0], g.stream));
    return 0;
}

int mw_timer_stop(int slot)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (slot < 0 || slot >= kTimerSlots) return fail("mw_timer: slot %d outside 0..%d", slot, kTimerSlots - 1);
    if (!g.ev[slot][1]) return fail("mw_timer_stop: slot %d was never started", slot);
    HIPCHK(hipEventRecord(g.ev[slot][1], g.stream));
    return 0;
}

int mw_timer_elapsed_ms(int slot, float* ms)
{
    MW_LOCK;
    if (check_live()) return 1;
    if (slot < 0 || slot >= kTimerSlots) return fail("mw_timer: slot %d outside 0..%d", slot, kTimerSlots - 1);
    if (!g.ev[slot][1]) return fail("mw_timer_elapsed_ms: slot %d was never started", slot);
    HIPCHK(hipEventSynchronize(g.ev[slot][1]));
    HIPCHK(hipEventElapsedTime(ms, g.ev[slot][0], g.ev[slot][1]));
    return 0;
}

}  // extern "C"
