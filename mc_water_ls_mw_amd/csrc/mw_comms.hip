// libmw_comms.so -- the exchange layer of include/mw_comms.h: RCCL collectives behind a C ABI for the Fortran host.
// One process per GPU; every entry stages the host buffer through pinned memory and a device buffer on one stream.
// The payloads are tiny (3 x 808 bytes for the tables, one value for the broadcasts): latency is what counts, so
// the three table reductions can go out as one collective (mw_comms_allreduce3) and nothing is ever split.
#include "../../include/mw_comms.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>

namespace {

struct Comms {
    bool live = false;
    int rank = 0, size = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    char* h_stage = nullptr;        // pinned
    char* d_buf = nullptr;
    char* d_out = nullptr;          // all-gather result
    size_t cap = 0, cap_out = 0;
    std::string id_file;
    bool own_id_file = false;
} c;

char g_err[512] = "";

int fail(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return 1;
}

#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail("%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define NCCLOK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail("%s: %s", #call, ncclGetErrorString(r_)); } while (0)

int env_int(const char* a, const char* b, int dflt)
{
    const char* e = getenv(a);
    if ((!e || !*e) && b) e = getenv(b);
    return (e && *e) ? atoi(e) : dflt;
}

int reserve(size_t bytes, size_t bytes_out)
{
    if (bytes > c.cap) {
        if (c.h_stage) { HIPOK(hipHostFree(c.h_stage)); HIPOK(hipFree(c.d_buf)); }
        c.cap = bytes < 4096 ? 4096 : bytes;
        HIPOK(hipHostMalloc(&c.h_stage, c.cap, hipHostMallocDefault));
        HIPOK(hipMalloc(&c.d_buf, c.cap));
    }
    if (bytes_out > c.cap_out) {
        if (c.d_out) HIPOK(hipFree(c.d_out));
        c.cap_out = bytes_out < 4096 ? 4096 : bytes_out;
        HIPOK(hipMalloc(&c.d_out, c.cap_out));
    }
    return 0;
}

int check_live(const char* who)
{
    if (!c.live) return fail("%s: call mw_comms_init first", who);
    return 0;
}

// The RCCL unique id travels through a file: rank 0 writes <file>.tmp and renames it, the others wait for <file>.
// The file says WHOSE id it holds -- world size, rank 0's pid and that process's start time (/proc/<pid>/stat) -- and a
// reader only takes an id whose writer is alive: a file left behind by a job that failed or was killed (its rank 0 never
// reached mw_comms_finalize) is skipped until this job's rank 0 has replaced it, instead of sending the ranks into
// ncclCommInitRank with a dead job's id, where they would wait for each other for ever while holding their GPUs.
// One node by construction (the farm is 8 GPUs of one node; the reference's MPI ranks map to them one to one).
struct IdRecord {
    char magic[8];
    int world, pid;
    unsigned long long start;
    ncclUniqueId id;
};

unsigned long long proc_start_time(int pid)
{
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%d/stat", pid);
    FILE* fh = fopen(path, "r");
    if (!fh) return 0ull;
    const size_t n = fread(buf, 1, sizeof buf - 1, fh);
    fclose(fh);
    buf[n] = 0;
    const char* p = strrchr(buf, ')');                    // the command name may hold spaces and parentheses
    if (!p) return 0ull;
    unsigned long long v = 0ull;
    int field = 2;                                        // p points at the end of field 2; starttime is field 22
    for (++p; *p; ++p) {
        if (*p == ' ') { if (++field == 22) { v = strtoull(p + 1, nullptr, 10); break; } }
    }
    return v;
}

int exchange_id(ncclUniqueId* id)
{
    if (c.rank == 0) {
        (void)unlink(c.id_file.c_str());                  // whatever an earlier job left here is not ours
        IdRecord rec;
        std::memset(&rec, 0, sizeof rec);
        std::memcpy(rec.magic, "MWCOMMS1", 8);
        rec.world = c.size; rec.pid = (int)getpid(); rec.start = proc_start_time(rec.pid);
        NCCLOK(ncclGetUniqueId(&rec.id));
        *id = rec.id;
        const std::string tmp = c.id_file + ".tmp";
        FILE* fh = fopen(tmp.c_str(), "wb");
        if (!fh) return fail("mw_comms_init: cannot write %s", tmp.c_str());
        const size_t w = fwrite(&rec, 1, sizeof rec, fh);
        fclose(fh);
        if (w != sizeof rec) return fail("mw_comms_init: short write to %s", tmp.c_str());
        if (rename(tmp.c_str(), c.id_file.c_str()) != 0) return fail("mw_comms_init: cannot rename %s", tmp.c_str());
        c.own_id_file = true;
        return 0;
    }
    const int timeout_s = env_int("MW_COMMS_TIMEOUT", nullptr, 120);
    const auto t0 = std::chrono::steady_clock::now();
    const char* why = "no such file";
    for (;;) {
        FILE* fh = fopen(c.id_file.c_str(), "rb");
        if (fh) {
            IdRecord rec;
            const size_t r = fread(&rec, 1, sizeof rec, fh);
            fclose(fh);
            if (r != sizeof rec || std::memcmp(rec.magic, "MWCOMMS1", 8) != 0) why = "not an id record (yet)";
            else if (rec.world != c.size) why = "written for another world size";
            else if (rec.start == 0ull || proc_start_time(rec.pid) != rec.start) why = "its writer is gone (left by an earlier job)";
            else { *id = rec.id; return 0; }
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s))
            return fail("mw_comms_init: rank %d waited %d s for %s (%s)", c.rank, timeout_s, c.id_file.c_str(), why);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}

template <typename F>
int staged(const void* in, void* out, size_t bytes_in, size_t bytes_out, F&& collective)
{
    if (reserve(bytes_in > bytes_out ? bytes_in : bytes_out, 0)) return 1;
    if (in) std::memcpy(c.h_stage, in, bytes_in);
    if (in) HIPOK(hipMemcpyAsync(c.d_buf, c.h_stage, bytes_in, hipMemcpyHostToDevice, c.stream));
    if (int rc = collective()) return rc;
    if (out) HIPOK(hipMemcpyAsync(c.h_stage, c.d_buf, bytes_out, hipMemcpyDeviceToHost, c.stream));
    HIPOK(hipStreamSynchronize(c.stream));
    if (out) std::memcpy(out, c.h_stage, bytes_out);
    return 0;
}

}  // namespace

extern "C" {

const char* mw_comms_last_error(void) { return g_err; }
int mw_comms_rank(void) { return c.rank; }
int mw_comms_size(void) { return c.size; }

int mw_comms_init(int* rank_out, int* size_out)
{
    if (c.live) return fail("mw_comms_init: already initialised");
    c.rank = env_int("MW_COMMS_RANK", "RANK", 0);
    c.size = env_int("MW_COMMS_SIZE", "WORLD_SIZE", 1);
    if (c.size < 1 || c.rank < 0 || c.rank >= c.size) return fail("mw_comms_init: rank %d of %d", c.rank, c.size);
    const char* f = getenv("MW_COMMS_ID_FILE");
    if (f && *f) c.id_file = f;
    else {
        const char* port = getenv("MASTER_PORT");
        if (!(port && *port)) {
            // two jobs on one host would meet at the same default path: a job of more than one rank has to say which it is
            if (c.size > 1) return fail("mw_comms_init: %d ranks need MASTER_PORT or MW_COMMS_ID_FILE to find each other", c.size);
            c.id_file = std::string("/tmp/mw_comms_id.solo.") + std::to_string((long long)getpid());
        } else {
            c.id_file = std::string("/tmp/mw_comms_id.") + port;
        }
    }
    ncclUniqueId id;
    if (exchange_id(&id)) return 1;                       // before anything touches the GPU
    int ndev = 0;
    HIPOK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail("mw_comms_init: no GPU visible");
    c.device = env_int("MW_COMMS_DEVICE", "LOCAL_RANK", c.rank % ndev);
    if (c.device < 0 || c.device >= ndev) return fail("mw_comms_init: device %d of %d", c.device, ndev);
    HIPOK(hipSetDevice(c.device));
    HIPOK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    if (reserve(4096, 4096)) return 1;
    NCCLOK(ncclCommInitRank(&c.comm, c.size, id, c.rank));
    if (c.own_id_file) {                                  // every rank has read it (they are all in the communicator now)
        (void)unlink(c.id_file.c_str());
        c.own_id_file = false;
    }
    c.live = true;
    if (rank_out) *rank_out = c.rank;
    if (size_out) *size_out = c.size;
    return 0;
}

int mw_comms_allreduce_sum(double* buf, int n)
{
    if (check_live("mw_comms_allreduce_sum")) return 1;
    if (n < 1 || !buf) return fail("mw_comms_allreduce_sum: empty buffer");
    HIPOK(hipSetDevice(c.device));
    const size_t bytes = sizeof(double) * (size_t)n;
    return staged(buf, buf, bytes, bytes, [&]() -> int {
        NCCLOK(ncclAllReduce(c.d_buf, c.d_buf, (size_t)n, ncclDouble, ncclSum, c.comm, c.stream));
        return 0;
    });
}

int mw_comms_allreduce3(double* a, double* b, double* d, int n)
{
    if (check_live("mw_comms_allreduce3")) return 1;
    if (n < 1 || !a || !b) return fail("mw_comms_allreduce3: empty buffer");
    HIPOK(hipSetDevice(c.device));
    const int parts = d ? 3 : 2;
    const size_t one = sizeof(double) * (size_t)n, bytes = one * parts;
    if (reserve(bytes, 0)) return 1;
    std::memcpy(c.h_stage, a, one);
    std::memcpy(c.h_stage + one, b, one);
    if (d) std::memcpy(c.h_stage + 2 * one, d, one);
    HIPOK(hipMemcpyAsync(c.d_buf, c.h_stage, bytes, hipMemcpyHostToDevice, c.stream));
    NCCLOK(ncclAllReduce(c.d_buf, c.d_buf, (size_t)n * parts, ncclDouble, ncclSum, c.comm, c.stream));
    HIPOK(hipMemcpyAsync(c.h_stage, c.d_buf, bytes, hipMemcpyDeviceToHost, c.stream));
    HIPOK(hipStreamSynchronize(c.stream));
    std::memcpy(a, c.h_stage, one);
    std::memcpy(b, c.h_stage + one, one);
    if (d) std::memcpy(d, c.h_stage + 2 * one, one);
    return 0;
}

int mw_comms_allreduce_max(double* buf, int n)
{
    if (check_live("mw_comms_allreduce_max")) return 1;
    if (n < 1 || !buf) return fail("mw_comms_allreduce_max: empty buffer");
    HIPOK(hipSetDevice(c.device));
    const size_t bytes = sizeof(double) * (size_t)n;
    return staged(buf, buf, bytes, bytes, [&]() -> int {
        NCCLOK(ncclAllReduce(c.d_buf, c.d_buf, (size_t)n, ncclDouble, ncclMax, c.comm, c.stream));
        return 0;
    });
}

int mw_comms_allgather(const double* mine, double* all, int n)
{
    if (check_live("mw_comms_allgather")) return 1;
    if (n < 1 || !mine || !all) return fail("mw_comms_allgather: empty buffer");
    HIPOK(hipSetDevice(c.device));
    const size_t one = sizeof(double) * (size_t)n, total = one * (size_t)c.size;
    if (reserve(total, total)) return 1;
    std::memcpy(c.h_stage, mine, one);
    HIPOK(hipMemcpyAsync(c.d_buf, c.h_stage, one, hipMemcpyHostToDevice, c.stream));
    NCCLOK(ncclAllGather(c.d_buf, c.d_out, (size_t)n, ncclDouble, c.comm, c.stream));
    HIPOK(hipMemcpyAsync(c.h_stage, c.d_out, total, hipMemcpyDeviceToHost, c.stream));
    HIPOK(hipStreamSynchronize(c.stream));
    std::memcpy(all, c.h_stage, total);
    return 0;
}

int mw_comms_bcast(void* buf, long nbytes, int root)
{
    if (check_live("mw_comms_bcast")) return 1;
    if (nbytes < 1 || !buf) return fail("mw_comms_bcast: empty buffer");
    if (root < 0 || root >= c.size) return fail("mw_comms_bcast: root %d of %d", root, c.size);
    HIPOK(hipSetDevice(c.device));
    const size_t bytes = (size_t)nbytes;
    return staged(buf, buf, bytes, bytes, [&]() -> int {
        NCCLOK(ncclBroadcast(c.d_buf, c.d_buf, bytes, ncclChar, root, c.comm, c.stream));
        return 0;
    });
}

int mw_comms_sendrecv(void* buf, long nbytes, int snode, int rnode)
{
    if (check_live("mw_comms_sendrecv")) return 1;
    if (nbytes < 1 || !buf) return fail("mw_comms_sendrecv: empty buffer");
    if (snode < 0 || snode >= c.size || rnode < 0 || rnode >= c.size || snode == rnode)
        return fail("mw_comms_sendrecv: %d -> %d of %d", snode, rnode, c.size);
    if (c.rank != snode && c.rank != rnode) return 0;     // comms_p2preal: only the two ranks act (comms_mpi.f90:181-190)
    HIPOK(hipSetDevice(c.device));
    const size_t bytes = (size_t)nbytes;
    if (c.rank == snode)
        return staged(buf, nullptr, bytes, 0, [&]() -> int { NCCLOK(ncclSend(c.d_buf, bytes, ncclChar, rnode, c.comm, c.stream)); return 0; });
    return staged(nullptr, buf, 0, bytes, [&]() -> int { NCCLOK(ncclRecv(c.d_buf, bytes, ncclChar, snode, c.comm, c.stream)); return 0; });
}

int mw_comms_barrier(void)
{
    if (check_live("mw_comms_barrier")) return 1;
    double one = 1.0;                                     // RCCL has no barrier: a one-element all-reduce is one
    return mw_comms_allreduce_sum(&one, 1);
}

int mw_comms_finalize(void)
{
    if (c.own_id_file) { unlink(c.id_file.c_str()); c.own_id_file = false; }   // also after an initialisation that failed half way
    if (!c.live) return 0;
    HIPOK(hipSetDevice(c.device));
    HIPOK(hipStreamSynchronize(c.stream));
    NCCLOK(ncclCommDestroy(c.comm));
    if (c.h_stage) HIPOK(hipHostFree(c.h_stage));
    if (c.d_buf) HIPOK(hipFree(c.d_buf));
    if (c.d_out) HIPOK(hipFree(c.d_out));
    HIPOK(hipStreamDestroy(c.stream));
    c = Comms{};
    return 0;
}

// The failure path: a collective failed or a peer is gone.  Nothing here may wait for the other ranks -- no stream
// synchronisation, no ncclCommDestroy (both can block for ever with a peer missing, and the failing rank would keep its GPU):
// the id file goes, the communicator is aborted, and the caller stops the process.
int mw_comms_abort(void)
{
    if (c.own_id_file) { unlink(c.id_file.c_str()); c.own_id_file = false; }
    if (!c.live) return 0;
    c.live = false;
    (void)ncclCommAbort(c.comm);
    return 0;
}

}  // extern "C"
