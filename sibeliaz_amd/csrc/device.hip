// device.hip — HBM residency, workspaces and launches for the gfx950 block-finder kernels, plus the
// single-GPU phase loop (BlocksFinder::FindBlocks, blocksfinder.h:453-530).
//
// Data layout in HBM (one copy per GPU, read-only except `used`):
//   chrStart u32[C+1] | posId i32[P] | posPos u32[P] | posCh u8[P] | posRevCh u8[P]
//   occStart u32[V+1] | occG u32[P] | occChr u32[P] | used u32[ceil(P/32)+1]        ~ 18.1 B per occurrence
// Seeds and results travel through pinned, device-mapped host memory (the kernel reads 8 B per seed
// and writes a 96 B header + 16 B per result instance), so a launch needs no explicit copies.
#include <hip/hip_runtime.h>

#include <time.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <unordered_map>
#include <vector>

#include "lcb_device.h"
#include "lcb_kernel.h"

#define HIP_CHECK(x)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) throw LcbError(std::string(#x) + " failed: " + hipGetErrorString(e_)); \
    } while (0)

// MODE 0/1/2 = small / medium / big (lcb_kernel.h): where the per-path instance pool and vote table live.
// NW wavefronts per workgroup: wave 0 runs the per-seed algorithm, the rest help with the votes (lcb_kernel.h).
#define LCB_NW_SMALL 16
#define LCB_NW_MEDIUM 16
#define LCB_NW_BIG 4
// PROF adds the flight recorder and the in-kernel section timers (LCB_DEBUG / LCB_TRACE_SEEDS); compiled out otherwise.
template <int MODE, bool STATS, int NW, bool PROF>
__global__ __launch_bounds__(64 * NW) void lcb_process_kernel(LcbTables T, LcbKParams P, const LcbKSeed* seeds, uint32_t nSeeds,
                                                         LcbWork W, LcbSeedOut* out, uint4* arena, unsigned long long arenaCap,
                                                         uint2* fpArena, unsigned long long fpCap)
{
    lcb_process_body<MODE, STATS, NW, PROF>(T, P, seeds, nSeeds, W, out, arena, arenaCap, fpArena, fpCap);
}

// Workspace slots start with an empty path set (and, in big mode, an empty vote table); the process
// kernel leaves them empty again after every seed.
__global__ __launch_bounds__(256) void lcb_init_slots_kernel(uint8_t* base, uint64_t slotBytes, LcbSlotLayout L,
                                                             uint32_t pathCap, uint32_t voteCap)
{
    uint8_t* slot = base + (uint64_t)blockIdx.x * slotBytes;
    int32_t* pKeys = (int32_t*)(slot + L.pKeys);
    for (uint32_t i = threadIdx.x; i < pathCap; i += blockDim.x) pKeys[i] = LCB_EMPTY_KEY;
    int32_t* vKey = (int32_t*)(slot + L.vKey);
    uint32_t* vCount = (uint32_t*)(slot + L.vCount);
    unsigned long long* vLast = (unsigned long long*)(slot + L.vLast);
    for (uint32_t i = threadIdx.x; i < voteCap; i += blockDim.x) { vKey[i] = LCB_EMPTY_KEY; vCount[i] = 0; vLast[i] = 0; }
}

// Predicted views 1..nViews of the `used` bitmap start as copies of the live state (view 0); stride is a multiple of 4 words.
__global__ __launch_bounds__(256) void lcb_copy_views_kernel(uint32_t* used, uint32_t strideWords, uint32_t nViews)
{
    const uint4* src = (const uint4*)used;
    const uint32_t n4 = strideWords / 4;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        for (uint32_t w = 1; w <= nViews; w++) ((uint4*)(used + (size_t)w * strideWords))[i] = v;
    }
}

// MarkUsed over [lo, hi) (junctionstorage.h:285-295): one workgroup per range.
__global__ __launch_bounds__(256) void lcb_mark_kernel(uint32_t* used, const uint64_t* ranges, uint32_t n)
{
    const uint32_t r = blockIdx.x;
    if (r >= n) return;
    const uint64_t lo = ranges[2 * r], hi = ranges[2 * r + 1];
    if (hi <= lo) return;
    const uint64_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
    for (uint64_t w = w0 + threadIdx.x; w <= w1; w += blockDim.x) {
        uint32_t m = 0xFFFFFFFFu;
        if (w == w0) m &= 0xFFFFFFFFu << (lo & 31);
        if (w == w1) m &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
        atomicOr(&used[w], m);
    }
}

namespace {

struct WorkSet {
    uint8_t* base = nullptr;
    uint64_t slotBytes = 0;
    uint32_t nSlots = 0, pathCap = 0, bodyCap = 0, bestCap = 0, instCap = 0, voteCap = 0;
    int mode = 0;          // 0 small, 1 medium, 2 big
    bool big = false;      // mode == 2: instance pool and vote table in the workspace
};

uint32_t envU32(const char* name, uint32_t dflt)
{
    const char* v = getenv(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

}  // namespace

struct lcb_device_impl {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const lcb_graph* g = nullptr;
    lcb_params p{};
    LcbTables T{};
    LcbKParams KP{};
    std::vector<void*> owned;
    uint32_t* dUsed = nullptr;
    size_t usedWords = 0;      // words of one view (a multiple of 4)
    int maxViews = 0;          // predicted views allocated behind the live bitmap (view 0)
    uint32_t* dCursor = nullptr;                 // [0] work tickets, [2..3] arena allocator (u64), [4..5] footprint allocator (u64)
    uint32_t cursorBase = 0;
    unsigned long long arenaBase = 0;
    WorkSet small, medium, big;
    // pinned, device-mapped host buffers
    LcbKSeed* hSeeds = nullptr;
    LcbSeedOut* hOut = nullptr;
    uint4* hArena = nullptr;
    uint2* hFp = nullptr;                        // footprint arena (pinned)
    unsigned long long fpCap = 0, fpBase = 0;
    uint64_t* hRanges = nullptr;
    uint32_t* hDbg = nullptr;                    // flight recorder (LCB_DEBUG=1): 16 words per workgroup
    uint32_t dbgSlots = 0;
    bool forceProf = false;                      // LCB_FORCE_PROF=1: always use the instrumented kernel variants
    bool seedTrace = false;                      // LCB_TRACE_SEEDS=1: add per-seed profile lines to the launch trace
    FILE* traceFile = nullptr;                   // LCB_TRACE_LAUNCHES=<file>: one line per launch (seeds, grid, mode, ms)
    double watchdogS = 0;                        // LCB_WATCHDOG_S: abort a launch that runs longer (0 = wait forever)
    uint32_t batchCap = 0;
    unsigned long long arenaCap = 0;
    uint32_t rangeCap = 0;
    bool stats = false;
    bool wantFp = false;                         // emit footprints (speculative engine)
    // seeds that overflowed the LDS capacities before: (vid, ch) -> kernel mode to start with next time, so that a
    // recomputation does not repeat the doomed small-mode attempt
    std::unordered_map<uint64_t, uint8_t> modeHint;
    std::vector<uint64_t> hintBits = std::vector<uint64_t>(1024, 0);   // 65 536-bit prefilter in front of modeHint (most seeds have no hint)
    double kernelMs = 0;
    int64_t launches = 0, bigRetries = 0;

    void use() { HIP_CHECK(hipSetDevice(ordinal)); }

    template <class T_>
    T_* upload(const T_* src, size_t n)
    {
        void* d = nullptr;
        HIP_CHECK(hipMalloc(&d, (n ? n : 1) * sizeof(T_)));
        owned.push_back(d);
        if (n) HIP_CHECK(hipMemcpy(d, src, n * sizeof(T_), hipMemcpyHostToDevice));
        return (T_*)d;
    }

    void allocWork(WorkSet& w)
    {
        if (w.base) { HIP_CHECK(hipFree(w.base)); w.base = nullptr; }
        const LcbSlotLayout L = lcb_slot_layout(w.pathCap, w.bodyCap, w.bestCap, w.big ? w.instCap : 0, w.big ? w.voteCap : 0);
        w.slotBytes = L.total;
        HIP_CHECK(hipMalloc((void**)&w.base, (size_t)w.slotBytes * w.nSlots));
        hipLaunchKernelGGL(lcb_init_slots_kernel, dim3(w.nSlots), dim3(256), 0, stream, w.base, w.slotBytes, L, w.pathCap,
                           w.big ? w.voteCap : 0u);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    void allocArena(unsigned long long cap)
    {
        if (hArena) HIP_CHECK(hipHostFree(hArena));
        if (hFp) HIP_CHECK(hipHostFree(hFp));
        arenaCap = cap; fpCap = cap;
        HIP_CHECK(hipHostMalloc((void**)&hArena, (size_t)cap * sizeof(uint4), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&hFp, (size_t)cap * sizeof(uint2), hipHostMallocDefault));
    }

    // One launch over hSeeds[0..m): returns after the stream has drained.
    void launch(WorkSet& w, uint32_t m)
    {
        LcbWork W;
        W.base = w.base; W.slotBytes = w.slotBytes; W.pathCap = w.pathCap; W.bodyCap = w.bodyCap; W.bestCap = w.bestCap;
        W.instCap = w.instCap; W.voteCap = w.voteCap;
        W.cursor = dCursor; W.cursorBase = cursorBase;
        W.arenaCursor = (unsigned long long*)(dCursor + 2); W.arenaBase = arenaBase;
        W.fpCursor = (unsigned long long*)(dCursor + 4); W.fpBase = fpBase;
        const uint32_t grid = m < w.nSlots ? m : w.nSlots;
        W.dbg = (hDbg && grid <= dbgSlots) ? hDbg : nullptr;
        if (W.dbg) memset(hDbg, 0, (size_t)grid * 16 * sizeof(uint32_t));
        if (watchdogS > 0) for (uint32_t i = 0; i < m; i++) hOut[i].status = 0xFFFFFFFFu;   // lets the watchdog name unfinished seeds
        HIP_CHECK(hipEventRecord(ev0, stream));
#define LCB_NW(MODE) (MODE == 2 ? LCB_NW_BIG : (MODE == 1 ? LCB_NW_MEDIUM : LCB_NW_SMALL))
#define LCB_LAUNCH(MODE, ST, PF) hipLaunchKernelGGL((lcb_process_kernel<MODE, ST, LCB_NW(MODE), PF>), dim3(grid), dim3(64 * LCB_NW(MODE)), 0, stream, \
                                                  T, KP, hSeeds, m, W, hOut, hArena, arenaCap, wantFp ? hFp : nullptr, fpCap)
#define LCB_LAUNCH_MODE(MODE) do { if (stats) LCB_LAUNCH(MODE, true, false); else if (prof) LCB_LAUNCH(MODE, false, true); else LCB_LAUNCH(MODE, false, false); } while (0)
        const bool prof = W.dbg != nullptr || seedTrace || forceProf;
        if (w.mode == 2) LCB_LAUNCH_MODE(2);
        else if (w.mode == 1) LCB_LAUNCH_MODE(1);
        else LCB_LAUNCH_MODE(0);
#undef LCB_LAUNCH_MODE
#undef LCB_LAUNCH
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(ev1, stream));
        if (watchdogS > 0) {
            // bounded wait: a kernel that does not finish is reported with its flight recorder instead of hanging the caller
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                const hipError_t q = hipEventQuery(ev1);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) HIP_CHECK(q);
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el > watchdogS) {
                    fprintf(stderr, "lcb: kernel watchdog: launch of %u seeds (%s mode, grid %u) still running after %.1f s\n", m, w.mode == 2 ? "big" : (w.mode == 1 ? "medium" : "small"), grid, el);
                    {
                        int shownSeeds = 0;
                        uint32_t unfinished = 0;
                        for (uint32_t i = 0; i < m; i++) if (hOut[i].status == 0xFFFFFFFFu) unfinished++;
                        fprintf(stderr, "  %u of %u seeds unfinished; first ones:", unfinished, m);
                        for (uint32_t i = 0; i < m && shownSeeds < 8; i++)
                            if (hOut[i].status == 0xFFFFFFFFu) { fprintf(stderr, " [%u] vid=%d ch=%d", i, hSeeds[i].vid, hSeeds[i].ch); shownSeeds++; }
                        fprintf(stderr, "\n");
                    }
                    if (W.dbg) {
                        int shown = 0;
                        for (uint32_t b = 0; b < grid && shown < 8; b++) {
                            const uint32_t* r = hDbg + 16 * b;
                            if (r[0] == 2) continue;
                            shown++;
                            fprintf(stderr, "  wg %u: state=%u seed=%u stage=%u nInst=%u nRight=%u nLeft=%u ext=%u next=%d g=%u\n", b, r[0], r[1], r[2], r[3], r[4], r[5], r[6], (int)r[7], r[8]);
                        }
                    }
                    fflush(stderr);
                    throw LcbError("kernel watchdog expired (LCB_WATCHDOG_S)");
                }
                if (el > 0.002) { struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr); }
            }
        }
        HIP_CHECK(hipStreamSynchronize(stream));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        kernelMs += ms;
        launches++;
        if (traceFile) {
            fprintf(traceFile, "%lld\t%u\t%u\t%s\t%.4f\n", (long long)launches, m, grid, w.mode == 2 ? "big" : (w.mode == 1 ? "medium" : "small"), ms);
            if (!stats && seedTrace)          // per-seed profile of the slowest seeds of the launch (ticks are 10 ns)
                for (uint32_t i = 0; i < m; i++)
                    if (hOut[i].ctr[0] > 2000) fprintf(traceFile, "#seed\t%lld\t%u\t%d\tst=%u\tn=%u\tticks=%llu\tpush=%llu\tvote=%llu\tprobe=%llu\tinst=%llu\ttv=%llu\ttp=%llu\tts=%llu\n", (long long)launches, i, hSeeds[i].vid,
                            hOut[i].status, hOut[i].nInst, (unsigned long long)hOut[i].ctr[0], (unsigned long long)hOut[i].ctr[1], (unsigned long long)hOut[i].ctr[2],
                            (unsigned long long)hOut[i].ctr[3], (unsigned long long)hOut[i].ctr[4], (unsigned long long)hOut[i].ctr[5], (unsigned long long)hOut[i].ctr[6],
                            (unsigned long long)hOut[i].ctr[7]);
        }
        cursorBase += m + grid;                    // every workgroup consumed exactly one ticket past the end
        for (uint32_t i = 0; i < m; i++) { arenaBase += hOut[i].nInst; fpBase += hOut[i].nFp; }
    }
};

lcb_device* lcb_device_create_impl(const lcb_graph* g, const lcb_params* p, int ordinal)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw LcbError("no HIP device available: the block-finder hot path runs only on the GPU (there is no CPU fallback)");
    if (ordinal < 0 || ordinal >= count) throw LcbError("HIP device ordinal out of range");
    auto* d = new lcb_device_impl();
    auto* handle = new lcb_device{d};
    try {
        d->ordinal = ordinal; d->g = g; d->p = *p;
        d->use();
        HIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreate(&d->ev0));
        HIP_CHECK(hipEventCreate(&d->ev1));
        const uint64_t P = g->nPos();
        std::vector<uint32_t> cs(g->chrStart.begin(), g->chrStart.end());
        d->T.chrStart = d->upload(cs.data(), cs.size());
        d->T.posId = d->upload(g->posId.data(), P);
        d->T.posPos = d->upload(g->posPos.data(), P);
        d->T.posCh = d->upload(g->posCh.data(), P);
        d->T.posRevCh = d->upload(g->posRevCh.data(), P);
        d->T.occStart = d->upload(g->occStart.data(), g->occStart.size());
        {
            std::vector<uint4> rec((size_t)P);
            for (uint64_t j = 0; j < P; j++) {
                const uint32_t q = g->occG[j];
                rec[j] = uint4{q, g->occChr[j], g->posPos[q], (uint32_t)g->posId[q]};
            }
            d->T.occRec = d->upload(rec.data(), rec.size());
        }
        d->usedWords = ((size_t)(P / 32 + 2) + 3) & ~(size_t)3;
        // predicted `used` views for the engine's dry-run launches: at most LCB_VIEWS (default 256), within 2 GiB
        d->maxViews = (int)std::min<uint64_t>(envU32("LCB_VIEWS", 256), (2ull << 30) / (d->usedWords * 4));
        HIP_CHECK(hipMalloc((void**)&d->dUsed, d->usedWords * 4 * (size_t)(d->maxViews + 1)));
        HIP_CHECK(hipMemset(d->dUsed, 0, d->usedWords * 4));
        d->T.used = d->dUsed; d->T.usedStride = (uint32_t)d->usedWords;
        d->T.nChr = g->nChr(); d->T.nVertex = g->nVertex; d->T.nPos = (uint32_t)P;
        d->KP.k = p->k; d->KP.minBlock = p->min_block; d->KP.maxBranch = p->max_branch; d->KP.maxFlank = p->max_flank;
        d->KP.depth = p->looking_depth;
        HIP_CHECK(hipMalloc((void**)&d->dCursor, 32));
        HIP_CHECK(hipMemset(d->dCursor, 0, 32));
        // small (LDS) mode: instances / vote table in LDS, path set + bodies + snapshot in a 1/4-MB global slot
        d->small.big = false; d->small.mode = 0; d->small.nSlots = envU32("LCB_SLOTS", 512);
        d->small.pathCap = envU32("LCB_PATH_CAP", 32768); d->small.bodyCap = d->small.pathCap / 2; d->small.bestCap = LCB_IC_SMALL;
        d->allocWork(d->small);
        // medium mode: 4x the LDS capacities, one workgroup per CU
        d->medium.big = false; d->medium.mode = 1; d->medium.nSlots = envU32("LCB_MEDIUM_SLOTS", 256);
        d->medium.pathCap = d->small.pathCap < 131072 ? 131072 : d->small.pathCap; d->medium.bodyCap = d->medium.pathCap / 2; d->medium.bestCap = LCB_IC_MEDIUM;
        if (envU32("LCB_FORCE_BIG", 0)) { d->medium.pathCap = d->small.pathCap; d->medium.bodyCap = d->small.bodyCap; }
        d->allocWork(d->medium);
        // big (global-memory) mode for seeds that overflow the LDS capacities; grows on demand
        d->big.big = true; d->big.mode = 2; d->big.nSlots = envU32("LCB_BIG_SLOTS", 64);
        d->big.pathCap = 262144; d->big.bodyCap = 131072; d->big.instCap = 4096; d->big.voteCap = 65536; d->big.bestCap = 4096;
        d->allocWork(d->big);
        d->batchCap = envU32("LCB_BATCH", 65536);
        HIP_CHECK(hipHostMalloc((void**)&d->hSeeds, (size_t)d->batchCap * sizeof(LcbKSeed), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&d->hOut, (size_t)d->batchCap * sizeof(LcbSeedOut), hipHostMallocDefault));
        d->allocArena(1u << 20);
        const char* tf = getenv("LCB_TRACE_LAUNCHES");
        if (tf && *tf) d->traceFile = fopen(tf, "w");
        d->seedTrace = envU32("LCB_TRACE_SEEDS", 0) != 0;
        d->forceProf = envU32("LCB_FORCE_PROF", 0) != 0;
        const char* wd = getenv("LCB_WATCHDOG_S");
        d->watchdogS = wd && *wd ? atof(wd) : 0;
        if (envU32("LCB_DEBUG", 0)) {
            d->dbgSlots = d->small.nSlots;
            if (d->medium.nSlots > d->dbgSlots) d->dbgSlots = d->medium.nSlots;
            if (d->big.nSlots > d->dbgSlots) d->dbgSlots = d->big.nSlots;
            HIP_CHECK(hipHostMalloc((void**)&d->hDbg, (size_t)d->dbgSlots * 16 * sizeof(uint32_t), hipHostMallocDefault));
        }
        d->rangeCap = 65536;
        HIP_CHECK(hipHostMalloc((void**)&d->hRanges, (size_t)d->rangeCap * 2 * sizeof(uint64_t), hipHostMallocDefault));
    } catch (...) {
        lcb_device_destroy_impl(handle);
        throw;
    }
    return handle;
}

void lcb_device_destroy_impl(lcb_device* h)
{
    if (!h) return;
    lcb_device_impl* d = h->impl;
    if (d) {
        (void)hipSetDevice(d->ordinal);
        if (d->stream) (void)hipStreamSynchronize(d->stream);
        for (void* p : d->owned) (void)hipFree(p);
        if (d->dUsed) (void)hipFree(d->dUsed);
        if (d->dCursor) (void)hipFree(d->dCursor);
        if (d->small.base) (void)hipFree(d->small.base);
        if (d->medium.base) (void)hipFree(d->medium.base);
        if (d->big.base) (void)hipFree(d->big.base);
        if (d->hSeeds) (void)hipHostFree(d->hSeeds);
        if (d->hOut) (void)hipHostFree(d->hOut);
        if (d->hArena) (void)hipHostFree(d->hArena);
        if (d->hFp) (void)hipHostFree(d->hFp);
        if (d->hRanges) (void)hipHostFree(d->hRanges);
        if (d->hDbg) (void)hipHostFree(d->hDbg);
        if (d->traceFile) fclose(d->traceFile);
        if (d->ev0) (void)hipEventDestroy(d->ev0);
        if (d->ev1) (void)hipEventDestroy(d->ev1);
        if (d->stream) (void)hipStreamDestroy(d->stream);
        delete d;
    }
    delete h;
}

void lcb_device_reset_used_impl(lcb_device* h)
{
    lcb_device_impl* d = h->impl;
    d->use();
    d->modeHint.clear();          // a new pass starts from scratch: no knowledge carried over from an earlier run
    std::fill(d->hintBits.begin(), d->hintBits.end(), 0ull);
    HIP_CHECK(hipMemsetAsync(d->dUsed, 0, d->usedWords * 4, d->stream));
    HIP_CHECK(hipStreamSynchronize(d->stream));
}

void lcb_device_set_used_impl(lcb_device* h, const uint32_t* words, int64_t nWords)
{
    lcb_device_impl* d = h->impl;
    d->use();
    if ((size_t)nWords > d->usedWords) throw LcbError("used bitmap has too many words");
    HIP_CHECK(hipMemcpy(d->dUsed, words, (size_t)nWords * 4, hipMemcpyHostToDevice));
}

void lcb_device_mark_used_impl(lcb_device* h, const uint64_t* ranges, int64_t n)
{
    lcb_device_impl* d = h->impl;
    d->use();
    const uint64_t P = d->g->nPos();
    for (int64_t done = 0; done < n;) {
        const uint32_t m = (uint32_t)((n - done) < (int64_t)d->rangeCap ? (n - done) : d->rangeCap);
        for (uint32_t i = 0; i < m; i++) {
            if (ranges[2 * (done + i) + 1] > P) throw LcbError("used range beyond the position table");
            d->hRanges[2 * i] = ranges[2 * (done + i)]; d->hRanges[2 * i + 1] = ranges[2 * (done + i) + 1];
        }
        hipLaunchKernelGGL(lcb_mark_kernel, dim3(m), dim3(256), 0, d->stream, d->dUsed, d->hRanges, m);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(d->stream));   // hRanges is reused
        done += m;
    }
}

void lcb_device_build_views_impl(lcb_device* h, int nViews, const LcbViewMark* marks, int64_t nMarks)
{
    lcb_device_impl* d = h->impl;
    d->use();
    if (nViews < 0 || nViews > d->maxViews) throw LcbError("more predicted views requested than the device holds");
    if (nViews == 0) return;
    const uint32_t n4 = (uint32_t)(d->usedWords / 4);
    hipLaunchKernelGGL(lcb_copy_views_kernel, dim3(std::min<uint32_t>((n4 + 255) / 256, 2048u)), dim3(256), 0, d->stream, d->dUsed,
                       (uint32_t)d->usedWords, (uint32_t)nViews);
    HIP_CHECK(hipGetLastError());
    // a mark of view v is set in the views v..nViews: expand to bit ranges of the whole allocation
    const uint64_t P = d->g->nPos(), viewBits = (uint64_t)d->usedWords * 32;
    uint32_t m = 0;
    auto flushRanges = [&]() {
        if (!m) return;
        hipLaunchKernelGGL(lcb_mark_kernel, dim3(m), dim3(256), 0, d->stream, d->dUsed, d->hRanges, m);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(d->stream));   // hRanges is reused
        m = 0;
    };
    for (int64_t k = 0; k < nMarks; k++) {
        if (marks[k].hi > P || marks[k].firstView < 1) throw LcbError("bad predicted mark");
        for (uint32_t v = marks[k].firstView; v <= (uint32_t)nViews; v++) {
            d->hRanges[2 * m] = v * viewBits + marks[k].lo; d->hRanges[2 * m + 1] = v * viewBits + marks[k].hi;
            if (++m == d->rangeCap) flushRanges();
        }
    }
    flushRanges();
}

int lcb_device_max_views_impl(lcb_device* h) { return h->impl->maxViews; }

void lcb_device_set_stats_impl(lcb_device* h, bool on) { h->impl->stats = on; }

void lcb_device_kernel_time_impl(lcb_device* h, double* ms, int64_t* launches)
{
    if (ms) *ms = h->impl->kernelMs;
    if (launches) *launches = h->impl->launches;
    h->impl->kernelMs = 0; h->impl->launches = 0;
}

int64_t lcb_device_big_retries_impl(lcb_device* h) { return h->impl->bigRetries; }

void lcb_device_process_impl(lcb_device* h, const lcb_seed* seeds, int64_t n, std::vector<uint64_t>& offsets,
                             std::vector<lcb_instance>& inst, int64_t* bestScore, lcb_counters* ctr,
                             std::vector<uint64_t>* fpOffsets, std::vector<lcb_fp>* fpOut, const uint32_t* view)
{
    lcb_device_impl* d = h->impl;
    d->use();
    offsets.assign((size_t)n + 1, 0);
    inst.clear();
    d->wantFp = fpOffsets != nullptr && fpOut != nullptr;
    std::vector<lcb_fp> fpFlat;                         // footprints in arrival order (only when they are wanted) ...
    std::vector<uint64_t> fpAt;                         // ... and where each seed's intervals start / how many there are
    std::vector<uint32_t> fpCnt;
    if (d->wantFp) { fpAt.assign((size_t)n, 0); fpCnt.assign((size_t)n, 0); fpFlat.reserve((size_t)n * 2); }
    auto takeFp = [&](int64_t s, const LcbSeedOut& o) {
        if (!d->wantFp) return;
        fpAt[(size_t)s] = fpFlat.size(); fpCnt[(size_t)s] = o.nFp;
        const uint2* src = d->hFp + o.fpOff;
        for (uint32_t e = 0; e < o.nFp; e++) fpFlat.push_back(lcb_fp{src[e].x, src[e].y});
    };
    // per-seed results are gathered out of order (retries), then laid out in seed order
    std::vector<std::vector<lcb_instance>> late;        // results of retried seeds
    std::vector<int64_t> lateOf((size_t)n, -1);
    std::vector<uint32_t> cnt((size_t)n, 0);
    std::vector<lcb_instance> flat;                     // first-pass results in arena order
    std::vector<uint64_t> flatOff((size_t)n, 0);
    auto addCtr = [&](const LcbSeedOut& o) {
        if (!ctr) return;
        ctr->n_walk += o.ctr[0]; ctr->n_occ += o.ctr[1]; ctr->n_compat_call += o.ctr[2]; ctr->n_compat_step += o.ctr[3];
        ctr->n_inst_out += o.ctr[4]; ctr->n_vote += o.ctr[5]; ctr->n_push += o.ctr[6]; ctr->n_process += o.ctr[7];
    };
    std::vector<int64_t> retry;                         // seeds that need larger workspaces (medium, then big)
    std::vector<int64_t> retryBig;                      // seeds known to need the big workspaces
    auto keyOf = [](const lcb_seed& sd) { return ((uint64_t)(uint32_t)sd.vid << 8) | (uint64_t)(uint8_t)sd.ch; };
    auto setHint = [&](const lcb_seed& sd, uint8_t mode) {
        const uint64_t key = keyOf(sd);
        const uint32_t hb = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 48);
        d->hintBits[hb >> 6] |= 1ull << (hb & 63);
        d->modeHint[key] = mode;
    };
    std::vector<int64_t> firstPass;
    firstPass.reserve((size_t)n);
    if (d->modeHint.empty()) for (int64_t s = 0; s < n; s++) firstPass.push_back(s);
    else
        for (int64_t s = 0; s < n; s++) {
            const uint64_t key = keyOf(seeds[s]);
            const uint32_t hb = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 48);
            if (!((d->hintBits[hb >> 6] >> (hb & 63)) & 1ull)) { firstPass.push_back(s); continue; }
            auto it = d->modeHint.find(key);
            if (it == d->modeHint.end()) firstPass.push_back(s);
            else if (it->second == 1) retry.push_back(s);
            else retryBig.push_back(s);
        }
    for (size_t base = 0; base < firstPass.size(); base += d->batchCap) {
        const uint32_t m = (uint32_t)((firstPass.size() - base) < d->batchCap ? (firstPass.size() - base) : d->batchCap);
        for (uint32_t i = 0; i < m; i++) {
            const int64_t s = firstPass[base + i];
            d->hSeeds[i].vid = seeds[s].vid; d->hSeeds[i].ch = seeds[s].ch; d->hSeeds[i].view = view ? view[s] : 0u; d->hSeeds[i].pad = 0;
        }
        d->launch(d->small, m);
        for (uint32_t i = 0; i < m; i++) {
            const LcbSeedOut& o = d->hOut[i];
            const int64_t s = firstPass[base + i];
            if (o.status == LCB_ST_OK) {
                cnt[(size_t)s] = o.nInst;
                flatOff[(size_t)s] = flat.size();
                for (uint32_t e = 0; e < o.nInst; e++) {
                    const uint4 r = d->hArena[o.arenaOff + e];
                    flat.push_back(lcb_instance{r.x, r.y, r.z, r.w});
                }
                if (bestScore) bestScore[s] = o.bestScore;
                takeFp(s, o);
                addCtr(o);
            } else if (o.status == LCB_ST_DIST_OVF) {
                throw LcbError("a path longer than 2^31 bp is not supported");
            } else { retry.push_back(s); if (o.status != LCB_ST_ARENA_OVF) setHint(seeds[s], 1); }
        }
    }
    // retries: medium mode (4x LDS capacities) first, then big mode, growing its capacities while seeds keep overflowing
    for (int round = 0; !retry.empty() || !retryBig.empty(); round++) {
        if (round > 13) throw LcbError("a seed keeps overflowing the device workspaces");
        WorkSet& ws = round == 0 ? d->medium : d->big;
        if (round == 1) { retry.insert(retry.end(), retryBig.begin(), retryBig.end()); retryBig.clear(); }
        std::vector<int64_t> again;
        for (size_t base = 0; base < retry.size(); base += d->batchCap) {
            const uint32_t m = (uint32_t)((retry.size() - base) < d->batchCap ? (retry.size() - base) : d->batchCap);
            for (uint32_t i = 0; i < m; i++) {
                const int64_t s = retry[base + i];
                d->hSeeds[i].vid = seeds[s].vid; d->hSeeds[i].ch = seeds[s].ch; d->hSeeds[i].view = view ? view[s] : 0u; d->hSeeds[i].pad = 0;
            }
            if (round > 0) d->bigRetries += m;
            d->launch(ws, m);
            for (uint32_t i = 0; i < m; i++) {
                const LcbSeedOut& o = d->hOut[i];
                const int64_t s = retry[base + i];
                if (o.status == LCB_ST_OK) {
                    cnt[(size_t)s] = o.nInst;
                    lateOf[(size_t)s] = (int64_t)late.size();
                    late.emplace_back();
                    for (uint32_t e = 0; e < o.nInst; e++) {
                        const uint4 r = d->hArena[o.arenaOff + e];
                        late.back().push_back(lcb_instance{r.x, r.y, r.z, r.w});
                    }
                    if (bestScore) bestScore[s] = o.bestScore;
                    takeFp(s, o);
                    addCtr(o);
                } else if (o.status == LCB_ST_DIST_OVF) {
                    throw LcbError("a path longer than 2^31 bp is not supported");
                } else { again.push_back(s); if (o.status != LCB_ST_ARENA_OVF) setHint(seeds[s], 2); }
            }
        }
        if (!again.empty() && round > 0) {
            // (statuses are per seed; growing everything keeps the logic simple and this path is rare)
            d->allocArena(d->arenaCap * 4);
            d->big.pathCap *= 2; d->big.bodyCap *= 2; d->big.instCap *= 2; d->big.voteCap *= 2; d->big.bestCap *= 2;
            d->allocWork(d->big);
        }
        retry.swap(again);
    }
    uint64_t total = 0;
    for (int64_t s = 0; s < n; s++) { offsets[(size_t)s] = total; total += cnt[(size_t)s]; }
    offsets[(size_t)n] = total;
    inst.resize((size_t)total);
    for (int64_t s = 0; s < n; s++) {
        if (!cnt[(size_t)s]) continue;
        const lcb_instance* src = lateOf[(size_t)s] >= 0 ? late[(size_t)lateOf[(size_t)s]].data() : flat.data() + flatOff[(size_t)s];
        memcpy(inst.data() + offsets[(size_t)s], src, (size_t)cnt[(size_t)s] * sizeof(lcb_instance));
    }
    if (d->wantFp) {
        fpOffsets->assign((size_t)n + 1, 0);
        uint64_t tf = 0;
        for (int64_t s = 0; s < n; s++) { (*fpOffsets)[(size_t)s] = tf; tf += fpCnt[(size_t)s]; }
        (*fpOffsets)[(size_t)n] = tf;
        fpOut->resize((size_t)tf);
        for (int64_t s = 0; s < n; s++)
            if (fpCnt[(size_t)s]) memcpy(fpOut->data() + (*fpOffsets)[(size_t)s], fpFlat.data() + fpAt[(size_t)s], (size_t)fpCnt[(size_t)s] * sizeof(lcb_fp));
    }
    d->wantFp = false;
}

namespace {

// The product's per-rank engine: the HIP kernels on one MI355X.
struct DeviceProcessor : LcbProcessor {
    lcb_device* dev;
    explicit DeviceProcessor(lcb_device* d) : dev(d) {}
    void process(const lcb_seed* seeds, const uint32_t* view, int64_t n, std::vector<uint64_t>& off, std::vector<lcb_instance>& inst,
                 std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) override
    {
        lcb_device_process_impl(dev, seeds, n, off, inst, nullptr, nullptr, &fpOff, &fp, view);
    }
    int maxViews() const override { return lcb_device_max_views_impl(dev); }
    void buildViews(int nViews, const LcbViewMark* marks, int64_t nMarks) override { lcb_device_build_views_impl(dev, nViews, marks, nMarks); }
    void mark(const uint64_t* ranges, int64_t n) override { lcb_device_mark_used_impl(dev, ranges, n); }
    void reset() override { lcb_device_reset_used_impl(dev); }
};

}  // namespace

void lcb_find_blocks_impl(const lcb_graph* g, lcb_device* dev, const lcb_params* p, const lcb_seed* seeds, int64_t nSeeds,
                          const LcbEngineConfig& cfg, std::vector<lcb_block>& blocks, lcb_stats* stats)
{
    double ms0 = 0; int64_t l0 = 0;
    lcb_device_kernel_time_impl(dev, &ms0, &l0);
    const int64_t retries0 = lcb_device_big_retries_impl(dev);
    DeviceProcessor proc(dev);
    LcbEngineStats es;
    lcb_engine_run(g, p, seeds, nSeeds, proc, cfg, blocks, &es);
    if (stats) {
        double ms = 0; int64_t l = 0;
        lcb_device_kernel_time_impl(dev, &ms, &l);
        stats->seeds = nSeeds; stats->blocks_found = es.blocksFound; stats->failures = es.failures;
        stats->launches = l; stats->kernel_ms = ms; stats->big_retries = lcb_device_big_retries_impl(dev) - retries0;
        stats->wall_ms = es.wallMs;
        stats->rounds = es.rounds; stats->recompute_launches = es.recomputeLaunches; stats->recomputed_seeds = es.recomputedSeeds;
        stats->conflict_launches = es.conflictLaunches; stats->conflict_seeds = es.conflictSeeds; stats->exchanges = es.exchanges;
            stats->jobs_used = es.jobsUsed; stats->views_built = es.viewsBuilt; stats->over_predicted = es.overPredicted;
            stats->process_ms = es.processMs; stats->plan_ms = es.planMs;
    }
}
