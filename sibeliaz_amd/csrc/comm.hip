// comm.hip — multi-GPU exchange of the round engine: RCCL all-gather over xGMI, called from the C++ host.
//
// The reference has no distributed path (OpenMP only). Here every launch of the round engine (engine.cpp)
// is dealt to the ranks — one rank per MI355X, tables and `used` bitmap replicated — and the per-seed results and
// footprints (KBs to a few MB per launch: latency-bound, far from the ~153 GB/s per xGMI link) are all-gathered so that
// every rank runs the identical ordered commit. Two ways to form the ranks:
//   * one PROCESS per GPU (bench.py under torch.distributed.run, or any launcher): rank 0 obtains an RCCL unique id
//     (lcb_comm_unique_id), ships its 128 bytes to the others by whatever means it has, every rank calls lcb_comm_create;
//   * one THREAD per GPU inside one process (the sibeliaz-lcb executable with LCB_GPUS=N): lcb_find_blocks_gpus.
// librccl is opened on first use (dlopen), so the library loads — and everything single-GPU works — where RCCL is absent.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>) && !defined(LCB_LOCAL_RCCL_DECLS)
#include <rccl/rccl.h>
#else
// A ROCm installation without RCCL's development headers still builds the library: these are the few declarations of RCCL's
// public C ABI that this file binds with dlsym.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum : int { ncclSuccess = 0 } ncclResult_t;        // (fixed underlying type: RCCL returns other codes than the one named here)
typedef enum : int { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
#endif

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <memory>
#include <thread>
#include <vector>

#include "lcb_device.h"
#include "lcb_host.h"

#define HIP_CHECK(x)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) throw LcbError(std::string(#x) + " failed: " + hipGetErrorString(e_)); \
    } while (0)

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [&]() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
        if (!r.lib) { err = std::string("cannot open librccl: ") + dlerror(); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    if (!err.empty()) throw LcbError(err);
    return r;
}

#define RCCL_CHECK(x)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (x);                                                                                 \
        if (r_ != ncclSuccess) throw LcbError(std::string(#x) + " failed: " + rccl().GetErrorString(r_));      \
    } while (0)

}  // namespace

struct lcb_comm {
    ncclComm_t comm = nullptr;
    std::atomic<bool> aborted{false};    // set (never cleared) when another rank thread of the set failed: no further collective is issued
    std::mutex issue;                    // held while a collective is ISSUED on `comm` and while the communicator is aborted: the check of `aborted` and
                                         // the call into RCCL are one step for the aborting thread (a thread that waits for a collective holds nothing)
    int rank = 0, world = 1, ordinal = 0;
    hipStream_t stream = nullptr;
    void* dSend = nullptr; void* dRecv = nullptr;
    size_t cap = 0;                      // bytes per rank the staging buffers hold
    int64_t gathers = 0; uint64_t bytes = 0;
    void use() { HIP_CHECK(hipSetDevice(ordinal)); }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        if (dSend) HIP_CHECK(hipFree(dSend));
        if (dRecv) HIP_CHECK(hipFree(dRecv));
        cap = n < 4096 ? 4096 : n + n / 2;
        HIP_CHECK(hipMalloc(&dSend, cap));
        HIP_CHECK(hipMalloc(&dRecv, cap * (size_t)world));
    }
    // The engine's all-gather: `bytes` from every rank into recv[world * bytes], through device memory so that the payload
    // travels GPU to GPU over xGMI.
    void allgather(const void* send, uint64_t n, void* recv)
    {
        if (aborted.load(std::memory_order_acquire)) throw LcbError("the communicator was aborted (another rank failed)");
        use();
        reserve((size_t)n);
        HIP_CHECK(hipMemcpyAsync(dSend, send, (size_t)n, hipMemcpyHostToDevice, stream));
        {
            std::lock_guard<std::mutex> lock(issue);
            if (aborted.load(std::memory_order_acquire)) throw LcbError("the communicator was aborted (another rank failed)");
            RCCL_CHECK(rccl().AllGather(dSend, dRecv, (size_t)n, ncclChar, comm, stream));
        }
        HIP_CHECK(hipMemcpyAsync(recv, dRecv, (size_t)n * (size_t)world, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        gathers++; bytes += n * (uint64_t)world;
    }
};

static int commAllgather(void* user, const void* send, uint64_t n, void* recv)
{
    try { ((lcb_comm*)user)->allgather(send, n, recv); return 0; }
    catch (std::exception& e) { lcb_set_error(e.what()); return 1; }
}

void lcb_comm_unique_id_impl(unsigned char* id)
{
    static_assert(sizeof(ncclUniqueId) == 128, "LCB_COMM_ID_BYTES");
    ncclUniqueId u;
    RCCL_CHECK(rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
}

static lcb_comm* wrapComm(ncclComm_t c, int ordinal, int rank, int world)
{
    lcb_comm* cm = new lcb_comm();
    cm->comm = c; cm->rank = rank; cm->world = world; cm->ordinal = ordinal;
    try { cm->use(); HIP_CHECK(hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking)); }
    catch (...) { delete cm; throw; }
    return cm;
}

lcb_comm* lcb_comm_create_impl(int ordinal, const unsigned char* id, int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) throw LcbError("lcb_comm_create: bad rank / world");
    HIP_CHECK(hipSetDevice(ordinal));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    RCCL_CHECK(rccl().CommInitRank(&c, world, u, rank));
    return wrapComm(c, ordinal, rank, world);
}

void lcb_comm_destroy_impl(lcb_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->ordinal);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->dSend) (void)hipFree(c->dSend);
    if (c->dRecv) (void)hipFree(c->dRecv);
    if (c->comm && !c->aborted.load()) (void)rccl().CommDestroy(c->comm);      // (an aborted communicator has been freed by ncclCommAbort)
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void lcb_comm_fill_config(lcb_comm* c, LcbEngineConfig& cfg)
{
    cfg.rank = c->rank; cfg.world = c->world; cfg.allgather = commAllgather; cfg.allgatherUser = c;
}

// BlocksFinder::FindBlocks on n GPUs of this node from ONE process: a device and a host thread per GPU, an RCCL
// communicator over them (ncclCommInitAll); every thread runs the engine as one rank. The set is persistent: tables are uploaded
// and RCCL is initialised once (lcb_gpus_create), every lcb_gpus_find_blocks is one pass. Returns rank 0's blocks and stats
// (all ranks hold the same; that is checked).
struct lcb_gpus_impl {
    const lcb_graph* g = nullptr;
    lcb_params p{};
    std::vector<int> ordinals;
    std::vector<lcb_device*> dev;
    std::vector<lcb_comm*> comm;
    bool broken = false;
    ~lcb_gpus_impl() { for (auto c : comm) lcb_comm_destroy_impl(c); for (auto d : dev) lcb_device_destroy_impl(d); }
};

lcb_gpus_impl* lcb_gpus_create_impl(const lcb_graph* g, const int* ordinals, int n, const lcb_params* p, const lcb_device_opts* opts, bool alwaysComm)
{
    if (n < 1) throw LcbError("lcb_gpus_create: no devices");
    std::unique_ptr<lcb_gpus_impl> m(new lcb_gpus_impl());
    m->g = g; m->p = *p; m->ordinals.assign(ordinals, ordinals + n);
    m->dev.assign((size_t)n, nullptr); m->comm.assign((size_t)n, nullptr);
    std::vector<std::string> err((size_t)n);
    {   // tables are uploaded to all GPUs at the same time
        std::vector<std::thread> th;
        for (int r = 0; r < n; r++) th.emplace_back([&, r]() { try { m->dev[(size_t)r] = lcb_device_create_impl(g, p, ordinals[r], opts); } catch (std::exception& e) { err[(size_t)r] = e.what(); } });
        for (auto& t : th) t.join();
        for (int r = 0; r < n; r++) if (!err[(size_t)r].empty()) throw LcbError("GPU " + std::to_string(ordinals[r]) + ": " + err[(size_t)r]);
    }
    if (n > 1 || alwaysComm) {
        std::vector<ncclComm_t> cs((size_t)n);
        RCCL_CHECK(rccl().CommInitAll(cs.data(), n, ordinals));
        for (int r = 0; r < n; r++) m->comm[(size_t)r] = wrapComm(cs[(size_t)r], ordinals[r], r, n);
    }
    return m.release();
}

void lcb_gpus_destroy_impl(lcb_gpus_impl* m) { delete m; }
int lcb_gpus_count_impl(const lcb_gpus_impl* m) { return (int)m->dev.size(); }

void lcb_gpus_find_blocks_impl(lcb_gpus_impl* m, const lcb_seed* seeds, int64_t nSeeds, LcbEngineConfig cfg, std::vector<lcb_block>& blocks, lcb_stats* stats)
{
    const int n = (int)m->dev.size();
    std::vector<std::vector<lcb_block>> out((size_t)n);
    std::vector<lcb_stats> st((size_t)n);
    std::vector<std::string> err((size_t)n);
    if (m->broken) throw LcbError("lcb_gpus: the set was stopped by an earlier failure (its communicators were aborted)");
    std::vector<std::thread> th;
    std::mutex abortMutex;
    for (int r = 0; r < n; r++)
        th.emplace_back([&, r]() {
            try {
                LcbEngineConfig c = cfg;
                c.progress = cfg.progress && r == 0;
                if (m->comm[(size_t)r]) lcb_comm_fill_config(m->comm[(size_t)r], c);
                lcb_find_blocks_impl(m->g, m->dev[(size_t)r], &m->p, seeds, nSeeds, c, out[(size_t)r], &st[(size_t)r]);
            } catch (std::exception& e) {
                err[(size_t)r] = e.what();
                // A rank that fails outside the engine's own error exchange (a HIP error, a view pool that cannot grow ...) would leave
                // the others blocked in their next all-gather and this call in join(): the communicators are aborted, once. The handles
                // stay where they are (other rank threads may be reading them at this moment): a flag keeps those threads from issuing
                // another collective, the handles are dropped with the set.
                std::lock_guard<std::mutex> lock(abortMutex);
                if (!m->broken && n > 1) {
                    m->broken = true;
                    for (auto c : m->comm) if (c) c->aborted.store(true, std::memory_order_release);
                    for (auto c : m->comm) if (c && c->comm) {
                        // `issue` is held by a thread that is ENQUEUEING a collective - normally microseconds. An enqueue that itself blocks (RCCL's lazy
                        // connection set-up on a communicator whose peer has already failed) would keep it for ever: after two seconds the abort goes
                        // ahead without the lock - aborting under a blocked enqueue is what unblocks it.
                        std::unique_lock<std::mutex> issuing(c->issue, std::defer_lock);
                        for (int tries = 0; tries < 200 && !issuing.try_lock(); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(10));
                        (void)rccl().CommAbort(c->comm);
                    }
                }
            }
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < n; r++) if (!err[(size_t)r].empty()) throw LcbError("rank " + std::to_string(r) + ": " + err[(size_t)r]);
    for (int r = 1; r < n; r++)
        if (out[(size_t)r].size() != out[0].size() || (out[0].size() && memcmp(out[(size_t)r].data(), out[0].data(), out[0].size() * sizeof(lcb_block)) != 0))
            throw LcbError("ranks ended with different block lists");
    blocks.swap(out[0]);
    if (stats) {
        *stats = st[0];
        for (int r = 1; r < n; r++) {
            stats->kernel_ms = std::max(stats->kernel_ms, st[(size_t)r].kernel_ms); stats->kernel_busy_ms = std::max(stats->kernel_busy_ms, st[(size_t)r].kernel_busy_ms);
            stats->launches = std::max(stats->launches, st[(size_t)r].launches);
        }
    }
}

// the one-shot form (devices, tables and communicator live for one call)
void lcb_find_blocks_gpus_impl(const lcb_graph* g, const int* ordinals, int n, const lcb_params* p, const lcb_device_opts* opts,
                               const lcb_seed* seeds, int64_t nSeeds, LcbEngineConfig cfg, std::vector<lcb_block>& blocks, lcb_stats* stats)
{
    std::unique_ptr<lcb_gpus_impl> m(lcb_gpus_create_impl(g, ordinals, n, p, opts, cfg.exchangeAlways));
    lcb_gpus_find_blocks_impl(m.get(), seeds, nSeeds, cfg, blocks, stats);
}
