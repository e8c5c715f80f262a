// device.hip — HBM residency, workspaces and launches for the gfx950 block-finder kernels, plus the
// single-GPU phase loop (BlocksFinder::FindBlocks, blocksfinder.h:453-530).
//
// Data layout in HBM (one copy per GPU, read-only except `used`):
//   chrLoHi uint2[C] | segBase u64[32] | posId i32[P] | posPos u32[P] | posCh u8[P] | posRevCh u8[P]
//   occStart u64[V+1] | occRec uint4[P] = {g, segment | chr, pos, id} | used u32[ceil(P/32)+1] + private pages of the views      ~ 30 B per occurrence
// A position is (segment, 32-bit g), flat index segBase[segment] + g (lcb_segments.h): inputs below 2^32 occurrences are one segment.
// Seeds and results travel through pinned, device-mapped host memory (the kernels read 16 B per seed and write a 40-B
// header + 16 B per result instance + 8 B per footprint interval), so a launch needs no explicit copies.
//
// A launch is (optionally) a screening kernel — one thread per seed decides whether Path::Init would create any
// instance at all and finalises the header of the seeds for which it would not — followed by the process kernel over
// the surviving seeds in one of four variants (lcb_kernel.h): compact (2 wavefronts per seed, 5 seeds per CU: launches
// with many seeds are throughput-bound), wide (16 wavefronts share the votes of one seed: launches with few seeds are
// as long as their longest seed), big (4096 instances: index, lists and vote table in LDS, instance fields in HBM) and
// huge (all per-path state in HBM, capacities grown on demand) for seeds that overflow the smaller ones.
#include <hip/hip_runtime.h>

#include <time.h>

#include <atomic>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "lcb_device.h"
#include "lcb_kernel.h"
#include "lcb_segments.h"

#define HIP_CHECK(x)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) throw LcbError(std::string(#x) + " failed: " + hipGetErrorString(e_)); \
    } while (0)

// MODE 0/1/2/3 = compact / wide / big / huge (lcb_kernel.h): where the per-path instance pool and vote table live.
// NW wavefronts per workgroup: wave 0 runs the per-seed algorithm, the rest share the votes.
#ifndef LCB_NW_WIDE
#define LCB_NW_WIDE 16
#endif
#ifndef LCB_NW_BIG
#define LCB_NW_BIG 16         // 16 wavefronts share the votes of the seeds with dozens of voters: -3.5 % on a config-3 pass against 8 (same box,
#endif                        // profiles/r03/ab_calls_5_to_12.txt), nothing on the k = 25 workloads (their paths have few voters)
#ifndef LCB_NW_COMPACT
#define LCB_NW_COMPACT 2      // one helper wavefront: -10 % on the compact launches of the 62-strain workload; 4 brings nothing more
#endif
#define LCB_NW_HUGE 8
// PROF adds the in-kernel section timers (LCB_DEBUG / LCB_TRACE_SEEDS); compiled out otherwise.
// SEG: the input has several segments (2^32 positions or more; lcb_kernel.h) - inputs of one segment run kernels without any segment code.
template <int MODE, bool STATS, int NW, bool PROF, bool SEG>
__global__ __launch_bounds__(64 * NW) void lcb_process_kernel(LcbTables T, LcbKParams P, const LcbKSeed* seeds, uint32_t nSeeds,
                                                         LcbWork W, LcbSeedOut* out, uint4* arena, unsigned long long arenaCap,
                                                         LcbFpOut* fpArena, unsigned long long fpCap)
{
    lcb_process_body<MODE, STATS, NW, PROF, SEG>(T, P, seeds, nSeeds, W, out, arena, arenaCap, fpArena, fpCap);
}

__global__ __launch_bounds__(256) void lcb_screen_kernel(LcbTables T, const LcbKSeed* seeds, uint32_t nSeeds, LcbSeedOut* out, uint32_t* live, uint32_t* nLive)
{
    lcb_screen_body(T, seeds, nSeeds, out, live, nLive);
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
    unsigned long long* vLast = (unsigned long long*)(slot + L.vLast);      // (slots with a vote table are the huge variant's: LcbCfg<3>::VLast)
    for (uint32_t i = threadIdx.x; i < voteCap; i += blockDim.x) { vKey[i] = LCB_EMPTY_KEY; vCount[i] = 0; vLast[i] = 0; }
}

// MarkUsed over [lo, hi) (junctionstorage.h:285-295) in the live bitmap: one workgroup per range.
struct LcbMarkRange { uint64_t lo, hi; };
__global__ __launch_bounds__(256) void lcb_mark_kernel(uint32_t* used, const LcbMarkRange* ranges, uint32_t n)
{
    const uint32_t r = blockIdx.x;
    if (r >= n) return;
    const uint64_t lo = ranges[r].lo, hi = ranges[r].hi;
    if (hi <= lo) return;
    const uint64_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
    for (uint64_t w = w0 + threadIdx.x; w <= w1; w += blockDim.x) {
        uint32_t m = 0xFFFFFFFFu;
        if (w == w0) m &= 0xFFFFFFFFu << (lo & 31);
        if (w == w1) m &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
        atomicOr(&used[w], m);
    }
}

// Predicted views are copy-on-write over 4-KB pages of the live bitmap. The views of a launch are nested (view v = view v-1 + the
// marks that first appear in v), so a page has one private VERSION per view that adds marks to it, and a view uses the newest version
// not younger than itself: one workgroup builds the chain of versions of one page - the first from the live page, each later one
// from its predecessor - and points the page-table entries of the views each version serves at it. Pages no predicted mark touches
// have no version: they read through to the live state (lcb_uword, lcb_kernel.h). (Round 2 copied a page once per VIEW that sees it
// changed: with a few hundred views of dense marks that was quadratic - 4.8 M page copies and a 2-GB pool per config-3 pass.)
static_assert(LCB_PAGE_SHIFT == 10u, "lcb_build_view_pages_kernel copies one page as 256 threads x 16 bytes");
struct LcbViewVersion { uint32_t view, nextView, poolPage, pieceBegin, pieceEnd; };   // serves the views [view, nextView); its new pieces [pieceBegin, pieceEnd)
struct LcbViewPage { uint32_t page, verBegin, verEnd; };                              // versions [verBegin, verEnd) of the launch's version list, ascending in view
struct LcbViewPiece { uint32_t lo, hi; };                                             // bit range inside the page, [lo, hi)
__global__ __launch_bounds__(256) void lcb_build_view_pages_kernel(uint32_t* live, uint32_t poolWords, uint32_t* viewTab, uint32_t nPages,
                                                                   const LcbViewPage* pages, const LcbViewVersion* versions, const LcbViewPiece* pieces)
{
    const LcbViewPage pg = pages[blockIdx.x];
    const uint4* src = (const uint4*)(live + ((size_t)pg.page << LCB_PAGE_SHIFT));
    for (uint32_t v = pg.verBegin; v < pg.verEnd; v++) {
        const LcbViewVersion e = versions[v];
        uint32_t* dstW = live + poolWords + ((size_t)e.poolPage << LCB_PAGE_SHIFT);      // the pool lies behind the live bitmap (poolWords: where it starts)
        ((uint4*)dstW)[threadIdx.x] = src[threadIdx.x];            // 256 threads x 16 B = one page
        __syncthreads();
        for (uint32_t q = e.pieceBegin; q < e.pieceEnd; q++) {
            const uint32_t lo = pieces[q].lo, hi = pieces[q].hi;   // hi > lo
            const uint32_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
            for (uint32_t w = w0 + threadIdx.x; w <= w1; w += 256) {
                uint32_t m = 0xFFFFFFFFu;
                if (w == w0) m &= 0xFFFFFFFFu << (lo & 31);
                if (w == w1) m &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
                dstW[w] |= m;                                       // pieces of one version are applied one after the other
            }
            __syncthreads();
        }
        // word offset from the live page to this copy (lcb_uword adds it to the word index), for every view the version serves
        const uint32_t off = poolWords + (e.poolPage << LCB_PAGE_SHIFT) - (pg.page << LCB_PAGE_SHIFT);
        for (uint32_t t = e.view + threadIdx.x; t < e.nextView; t += 256) viewTab[(size_t)t * nPages + pg.page] = off;
        src = (const uint4*)dstW;                                   // the next version of the page builds on this one
    }
}


// STREAM triad a = b + s * c over 16-B words: the measured HBM rate the roofline figure is put beside (bench.py).
__global__ __launch_bounds__(256) void lcb_triad_kernel(float4* a, const float4* b, const float4* c, float s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = b[i], y = c[i];
        a[i] = float4{x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w};
    }
}

namespace {

struct WorkSet {
    uint8_t* base = nullptr;
    uint64_t slotBytes = 0;
    uint32_t nSlots = 0, pathCap = 0, bodyCap = 0, bestCap = 0, instCap = 0, voteCap = 0;
    int mode = 0;          // 0 compact, 1 wide, 2 big, 3 huge
};

uint32_t envU32(const char* name, uint32_t dflt)
{
    const char* v = getenv(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

const char* modeName(int m) { return m == 3 ? "huge" : (m == 2 ? "big" : (m == 1 ? "wide" : "compact")); }

}  // namespace

// The predicted views of one launch: a page table per view and a pool of private pages behind the live bitmap.
struct ViewSpace {
    uint32_t* tab = nullptr;                     // [(maxViews + 1) * nPages]: word offset from a live page to the view's copy (0 = shared)
    int lastViews = 0;                           // views whose tables hold entries from the previous build
    uint32_t poolBasePage = 0;                   // first page of the pool, counted from the end of the live bitmap
    uint32_t poolPages = 0;                      // its capacity
    LcbViewPage* dEntries = nullptr; LcbViewVersion* dVersions = nullptr; LcbViewPiece* dPieces = nullptr;
    size_t entryCap = 0, versionCap = 0, pieceCap = 0;
};

// An asynchronous job batch (LcbProcessor::sideBegin): its own streams, pinned result buffers, workspace slots and views, so that it
// runs beside the synchronous launches of the commit. Wide and big kernels only (a job that needs another variant gets no result).
struct SideLane {
    hipStream_t sw = nullptr, sb = nullptr;      // the wide and the big kernel of a batch run side by side
    hipEvent_t w0 = nullptr, w1 = nullptr, b0 = nullptr, b1 = nullptr;
    LcbKSeed* hSeeds = nullptr; LcbSeedOut* hOut = nullptr; uint4* hArena = nullptr; LcbFpOut* hFp = nullptr;
    uint32_t* hList = nullptr;                   // pinned: ticket -> job index, [cap] for the wide kernel then [cap] for the big one
    uint32_t* hCtl = nullptr;                    // pinned staging of the control words
    uint32_t* dCtl = nullptr;                    // device: [0] wide tickets [1] big tickets [2..3] arena [4..5] footprints [6] wide jobs [7] big jobs [8] stop flag
    uint32_t cap = 0;
    unsigned long long arenaCap = 0;
    ViewSpace views;
    WorkSet wide, big;
    bool busy = false, released = false, ranW = false, ranB = false;
    int64_t n = 0;
    std::vector<lcb_seed> seeds;
};

struct lcb_device_impl {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const lcb_graph* g = nullptr;
    lcb_params p{};
    lcb_device_opts o{};
    LcbTables T{};
    LcbKParams KP{};
    LcbSegPlan plan;                             // host positions <-> (segment, g) of the device tables
    bool seg = false;                            // the SEG instantiations of the kernels run (several segments)
    std::vector<void*> owned;
    uint32_t* dUsed = nullptr;                   // the live bitmap
    size_t usedWords = 0;                        // its words (a multiple of the page size)
    uint32_t nPages = 0;
    int maxViews = 0;                            // predicted views (page tables) available per launch
    ViewSpace views;                             // ... of the synchronous launches (its pool lies behind the lanes' pools and grows on demand)
    std::vector<SideLane> lanes;                 // asynchronous job batches
    hipStream_t ctlStream = nullptr;             // stop flags of the lanes are written from here
    uint32_t lanePoolPages = 0;                  // private pages per lane (fixed: the live bitmap cannot move while a lane is running)
    int64_t sideBatches = 0, sideJobs = 0, sideNoLane = 0, sideNoFit = 0;
    struct lcb_async_call* async = nullptr;      // the call begun with processBegin and not yet ended
    uint32_t* dCursor = nullptr;                 // [0] work tickets, [1] live seeds, [2..3] arena allocator (u64), [4..5] footprint allocator (u64)
    uint32_t* dLive = nullptr;                   // ticket -> seed index of a screened launch
    uint32_t* hLive = nullptr;                   // ... copied to the host after the launch, [batchCap + 1]: the last word is their number
    WorkSet ws[4];                               // compact, wide, big, huge
    // pinned, device-mapped host buffers
    LcbKSeed* hSeeds = nullptr;
    LcbSeedOut* hOut = nullptr;
    LcbSeedCtr* hCtr = nullptr;                  // stats / instrumented variants only
    uint4* hArena = nullptr;
    LcbFpOut* hFp = nullptr;                     // footprint arena (pinned)
    unsigned long long fpCap = 0;
    LcbMarkRange* hRanges = nullptr;
    // second set of the host buffers, for the launch that runs while the host still commits the previous round (processBegin/End)
    LcbKSeed* hSeedsB = nullptr; LcbSeedOut* hOutB = nullptr; uint4* hArenaB = nullptr; LcbFpOut* hFpB = nullptr; uint32_t* hLiveB = nullptr;
    unsigned long long arenaCapB = 0, fpCapB = 0;
    hipEvent_t ev2 = nullptr, ev3 = nullptr;
    void swapBufs()
    {
        std::swap(hSeeds, hSeedsB); std::swap(hOut, hOutB); std::swap(hArena, hArenaB); std::swap(hFp, hFpB); std::swap(hLive, hLiveB);
        std::swap(arenaCap, arenaCapB); std::swap(fpCap, fpCapB);
    }
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
    // recomputation does not repeat the doomed attempt
    std::unordered_map<uint64_t, uint8_t> modeHint;
    std::vector<uint64_t> hintBits = std::vector<uint64_t>(1024, 0);   // 65 536-bit prefilter in front of modeHint (most seeds have no hint)
    double kernelMs = 0;                         // sum of the hipEvent-timed durations of the process kernels of every stream ...
    // ... and the intervals themselves, relative to evBase (recorded when the counters are read and reset): the kernels of the side
    // lanes run beside the synchronous ones, so the GPU-busy time of a pass is the UNION of the intervals, not their sum
    hipEvent_t evBase = nullptr;
    double sideKernelMs = 0;                     // ... the part of kernelMs that ran on the side lanes' streams
    std::vector<std::pair<float, float>> kernelSpans;
    void noteSpan(hipEvent_t a, hipEvent_t b)
    {
        float ta = 0, tb = 0;
        if (evBase && hipEventElapsedTime(&ta, evBase, a) == hipSuccess && hipEventElapsedTime(&tb, evBase, b) == hipSuccess) kernelSpans.emplace_back(ta, tb);
    }
    void resetSpans()
    {
        kernelSpans.clear();
        if (!evBase) HIP_CHECK(hipEventCreate(&evBase));
        HIP_CHECK(hipEventRecord(evBase, stream));
        HIP_CHECK(hipEventSynchronize(evBase));
    }
    double busyMs()                              // union of the recorded intervals
    {
        std::sort(kernelSpans.begin(), kernelSpans.end());
        double busy = 0; float hi = -1e30f;
        for (auto& q : kernelSpans) { if (q.second <= hi) continue; busy += q.second - std::max(q.first, hi); hi = q.second; }
        return busy;
    }
    int64_t launches = 0, bigRetries = 0;
    int64_t modeSeeds[4] = {0, 0, 0, 0};         // seeds handed to each kernel variant since creation
    // the compact variant's pools (lcb_device_opts.compact_pools): small = LcbCfg<4> (128 instances / 512 vote slots, 8 workgroups per CU) instead of
    // LcbCfg<0> (256 / 1 024, 5 per CU); chosen by the input, and given up for good if too many live seeds overflow the small pools
    bool compactSmall = false, compactAuto = false;
    uint32_t compactLargeSlots = 0;              // ws[0].nSlots with the large pools
    int64_t compactDone = 0, compactPoolOvf = 0; // seeds that ended in the compact variant with a result / by overflowing its instance pool or vote table
    int64_t longPathHints = 0;                   // results of the compact variant whose instances span more junctions than the wide variant's path set holds vertices
    int compactFellBack = 0;
    uint32_t compactIC() const { return compactSmall ? LcbCfg<4>::IC : LcbCfg<0>::IC; }
    double modeMs[4] = {0, 0, 0, 0};             // hipEvent-timed kernel time of each variant since creation (all streams) ...
    int64_t modeLaunches[4] = {0, 0, 0, 0};      // ... and its launches
    int64_t screened = 0, screenedDead = 0, viewPagesBuilt = 0;
    int64_t overflow[4][8] = {};                 // [variant][LcbStatus]: seeds that left a variant with that status
    int compactPathGrown = 0;                    // times the compact path set was enlarged (x4 each)
    int arenaGrown = 0;                          // times the result arena was enlarged (x4 each)
    double recentBigFrac = 0;                    // share of the seeds of the last sizeable call that needed the big variant
    int64_t joinedBig = 0;                       // seeds that ran in the big variant because their call had to go there anyway

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

    // a per-position table: the segments of the host array go to their places in the device's flat index space (contiguous unless a
    // test put gaps between them)
    template <class T_>
    T_* uploadSeg(const T_* src, const LcbSegPlan& pl)
    {
        void* dv = nullptr;
        HIP_CHECK(hipMalloc(&dv, (size_t)(pl.devPositions ? pl.devPositions : 1) * sizeof(T_)));
        owned.push_back(dv);
        for (uint32_t sg = 0; sg < pl.nSeg(); sg++) {
            const uint64_t a = pl.segStart[sg], b = pl.segStart[sg + 1];
            if (b > a) HIP_CHECK(hipMemcpy((T_*)dv + pl.segDev[sg], src + a, (size_t)(b - a) * sizeof(T_), hipMemcpyHostToDevice));
        }
        return (T_*)dv;
    }

    void allocWork(WorkSet& w)
    {
        if (w.base) { HIP_CHECK(hipFree(w.base)); w.base = nullptr; }
        // instCap != 0: instance fields in the slot (big, huge); voteCap != 0: index, lists and vote table too (huge)
        const LcbSlotLayout L = lcb_slot_layout(w.pathCap, w.bodyCap, w.bestCap, w.instCap, w.voteCap);
        w.slotBytes = L.total;
        HIP_CHECK(hipMalloc((void**)&w.base, (size_t)w.slotBytes * w.nSlots));
        hipLaunchKernelGGL(lcb_init_slots_kernel, dim3(w.nSlots), dim3(256), 0, stream, w.base, w.slotBytes, L, w.pathCap, w.voteCap);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    void allocArena(unsigned long long cap)
    {
        if (hArena) HIP_CHECK(hipHostFree(hArena));
        if (hFp) HIP_CHECK(hipHostFree(hFp));
        arenaCap = cap; fpCap = cap;
        HIP_CHECK(hipHostMalloc((void**)&hArena, (size_t)cap * sizeof(uint4), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&hFp, (size_t)cap * sizeof(LcbFpOut), hipHostMallocDefault));
    }

    // One launch over hSeeds[0..m): optional screening, then the process kernel of w's variant. Returns after the stream
    // has drained; hOut[0..m) then holds every seed's header. wait = false only enqueues (events evA/evB); finishLaunch()
    // then waits for it.
    void launch(WorkSet& w, uint32_t m, bool screen, bool wait = true)
    {
        hipEvent_t evA = wait ? ev0 : ev2, evB = wait ? ev1 : ev3;
        LcbWork W;
        W.base = w.base; W.slotBytes = w.slotBytes; W.pathCap = w.pathCap; W.bodyCap = w.bodyCap; W.bestCap = w.bestCap;
        W.instCap = w.instCap; W.voteCap = w.voteCap;
        W.cursor = dCursor; W.cursorBase = 0;
        W.live = screen ? dLive : nullptr; W.nLive = screen ? dCursor + 1 : nullptr;
        W.arenaCursor = (unsigned long long*)(dCursor + 2); W.arenaBase = 0;
        W.fpCursor = (unsigned long long*)(dCursor + 4); W.fpBase = 0;
        const bool prof = (hDbg != nullptr) || seedTrace || forceProf;
        W.ctr = (stats || prof) ? hCtr : nullptr;
        const uint32_t grid = m < w.nSlots ? m : w.nSlots;
        W.dbg = (hDbg && grid <= dbgSlots) ? hDbg : nullptr;
        W.abort = nullptr;
        if (W.dbg) memset(hDbg, 0, (size_t)grid * 16 * sizeof(uint32_t));
        if (W.ctr && !stats) memset(hCtr, 0, (size_t)m * sizeof(LcbSeedCtr));   // screened-out seeds write no profile
        if (watchdogS > 0 || screen) for (uint32_t i = 0; i < m; i++) hOut[i].status = LCB_ST_PENDING;   // unfinished seeds can be named (and a header nobody wrote is noticed)
        HIP_CHECK(hipMemsetAsync(dCursor, 0, 32, stream));
        HIP_CHECK(hipEventRecord(evA, stream));
        if (screen) {
            hipLaunchKernelGGL(lcb_screen_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, T, hSeeds, m, hOut, dLive, dCursor + 1);
            HIP_CHECK(hipGetLastError());
        }
#define LCB_NW(MODE) (MODE == 3 ? LCB_NW_HUGE : (MODE == 2 ? LCB_NW_BIG : (MODE == 1 ? LCB_NW_WIDE : LCB_NW_COMPACT)))     /* (0 and 4: the compact variant) */
#define LCB_LAUNCH(MODE, ST, PF, SG) hipLaunchKernelGGL((lcb_process_kernel<MODE, ST, LCB_NW(MODE), PF, SG>), dim3(grid), dim3(64 * LCB_NW(MODE)), 0, stream, \
                                                  T, KP, hSeeds, m, W, hOut, hArena, arenaCap, wantFp ? hFp : nullptr, fpCap)
#define LCB_LAUNCH_SEG(MODE, ST, PF) do { if (seg) LCB_LAUNCH(MODE, ST, PF, true); else LCB_LAUNCH(MODE, ST, PF, false); } while (0)
#define LCB_LAUNCH_MODE(MODE) do { if (stats) LCB_LAUNCH_SEG(MODE, true, false); else if (prof) LCB_LAUNCH_SEG(MODE, false, true); else LCB_LAUNCH_SEG(MODE, false, false); } while (0)
        if (w.mode == 3) LCB_LAUNCH_MODE(3);
        else if (w.mode == 2) LCB_LAUNCH_MODE(2);
        else if (w.mode == 1) LCB_LAUNCH_MODE(1);
        else if (compactSmall) LCB_LAUNCH_MODE(4);
        else LCB_LAUNCH_MODE(0);
#undef LCB_LAUNCH_MODE
#undef LCB_LAUNCH_SEG
#undef LCB_LAUNCH
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(evB, stream));       // [evA, evB]: screening + process kernel
        if (screen) {      // the host only looks at the seeds that survived the screening
            HIP_CHECK(hipMemcpyAsync(hLive + batchCap, dCursor + 1, 4, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(hLive, dLive, (size_t)m * 4, hipMemcpyDeviceToHost, stream));
        }
        if (!wait) return;
        finishLaunch(w, m, grid, W.dbg != nullptr, evA, evB);
    }

    void finishLaunch(WorkSet& w, uint32_t m, uint32_t grid, bool haveDbg, hipEvent_t evA, hipEvent_t evB)
    {
        if (watchdogS > 0) {
            // bounded wait: a kernel that does not finish is reported with its flight recorder instead of hanging the caller
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                const hipError_t q = hipEventQuery(evB);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) HIP_CHECK(q);
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el > watchdogS) {
                    fprintf(stderr, "lcb: kernel watchdog: launch of %u seeds (%s mode, grid %u) still running after %.1f s\n", m, modeName(w.mode), grid, el);
                    {
                        int shownSeeds = 0;
                        uint32_t unfinished = 0;
                        for (uint32_t i = 0; i < m; i++) if (hOut[i].status == LCB_ST_PENDING) unfinished++;
                        fprintf(stderr, "  %u of %u seeds unfinished; first ones:", unfinished, m);
                        for (uint32_t i = 0; i < m && shownSeeds < 8; i++)
                            if (hOut[i].status == LCB_ST_PENDING) { fprintf(stderr, " [%u] vid=%d ch=%d", i, hSeeds[i].vid, hSeeds[i].ch); shownSeeds++; }
                        fprintf(stderr, "\n");
                    }
                    if (haveDbg) {
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
        HIP_CHECK(hipEventElapsedTime(&ms, evA, evB));
        kernelMs += ms;
        launches++;
        noteSpan(evA, evB);
        modeSeeds[w.mode] += m;
        modeMs[w.mode] += ms; modeLaunches[w.mode]++;
        if (traceFile) {
            fprintf(traceFile, "%lld\t%u\t%u\t%s\t%.4f\n", (long long)launches, m, grid, modeName(w.mode), ms);
            if (!stats && seedTrace && hCtr)  // per-seed profile of the slowest seeds of the launch (ticks are 10 ns)
                for (uint32_t i = 0; i < m; i++)
                    if (hCtr[i].c[0] > 2000)
                        fprintf(traceFile, "#seed\t%lld\t%u\t%d\tst=%u\tn=%u\tticks=%llu\tpush=%llu\tvote=%llu\tprobe=%llu\tinst=%llu\ttv=%llu\ttp=%llu\tts=%llu\tcwalk=%llu\tcwaitb=%llu\tcreduce=%llu\tcscan=%llu\tvoters=%llu\tchunks=%llu\ttouch=%llu\n", (long long)launches, i, hSeeds[i].vid,
                            hOut[i].status, hOut[i].nInst, (unsigned long long)hCtr[i].c[0], (unsigned long long)hCtr[i].c[1], (unsigned long long)hCtr[i].c[2],
                            (unsigned long long)hCtr[i].c[3], (unsigned long long)hCtr[i].c[4], (unsigned long long)hCtr[i].c[5], (unsigned long long)hCtr[i].c[6],
                            (unsigned long long)hCtr[i].c[7], (unsigned long long)hCtr[i].c[8], (unsigned long long)hCtr[i].c[9], (unsigned long long)hCtr[i].c[10],
                            (unsigned long long)hCtr[i].c[11], (unsigned long long)hCtr[i].c[12], (unsigned long long)hCtr[i].c[13], (unsigned long long)hCtr[i].c[14]);
        }
    }
};

lcb_device* lcb_device_create_impl(const lcb_graph* g, const lcb_params* p, int ordinal, const lcb_device_opts* opts)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw LcbError("no HIP device available: the block-finder hot path runs only on the GPU (there is no CPU fallback)");
    if (ordinal < 0 || ordinal >= count) throw LcbError("HIP device ordinal out of range");
    if (p->max_branch >= 65000 || p->looking_depth >= 65000) throw LcbError("maxBranchSize / lookingDepth of 65000 or more are not supported by the device vote table");
    auto* d = new lcb_device_impl();
    auto* handle = new lcb_device{d};
    try {
        d->ordinal = ordinal; d->g = g; d->p = *p;
        if (opts) d->o = *opts;
        lcb_device_opts& o = d->o;
        d->use();
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, ordinal));
        const uint32_t nCu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
        // positions: (segment, g) pairs (lcb_segments.h); seg_cap / seg_gap are test hooks (small segments on small inputs, flat indices
        // beyond 2^32 through unused table space between the segments)
        d->plan = lcb_plan_segments(*g, o.seg_cap, o.seg_gap);
        d->seg = d->plan.nSeg() > 1 || o.seg_cap != 0;
        // defaults: compact 5 workgroups per CU (LDS-bound; 4 with the segment tables of the SEG kernels), wide 1 per CU, big 1 per CU
        // the compact variant's pools: small ones where a vertex has few occurrences (paths of few instances: k = 25 inputs of 8-16 genomes, 6-12
        // occurrences per vertex; the 62-strain shape has 26 at 1/40 of its size and more at full size, and its paths outgrow 128 instances)
        if (o.compact_pools > 2) throw LcbError("lcb_device_opts.compact_pools: 0 (by the input), 1 (256 instances / 1 024 vote slots) or 2 (128 / 512)");
        d->compactAuto = o.compact_pools == 0;
        // (the occurrence-weighted mean of the occurrences per vertex: "the typical occurrence belongs to a vertex with that many" - 6-14 for the k = 25 shapes of 8 / 16
        // genomes, 35+ for the 62-strain shape, 25 for the 10-strain one, whose many strain-private vertices would hide that in a plain mean)
        long double occSq = 0;
        for (size_t v = 0; v + 1 < g->occStart.size(); v++) { const long double nv = (long double)(g->occStart[v + 1] - g->occStart[v]); occSq += nv * nv; }
        d->compactSmall = o.compact_pools == 2 || (o.compact_pools == 0 && g->nPos() > 0 && occSq <= 20.0L * (long double)g->nPos());
        d->compactLargeSlots = o.compact_slots ? o.compact_slots : (d->seg ? 4 : 5) * nCu;
        if (!o.compact_slots) o.compact_slots = d->compactSmall ? 8 * nCu : d->compactLargeSlots;     // (8: four wavefronts per SIMD by registers, two per workgroup)
        if (!o.wide_slots) o.wide_slots = nCu;
        if (!o.big_slots) o.big_slots = nCu;
        if (!o.huge_slots) o.huge_slots = nCu / 4 ? nCu / 4 : 1;
        if (!o.path_cap) o.path_cap = 32768;
        if (!o.path_cap_max) o.path_cap_max = 1u << 20;
        if (!o.arena) o.arena = 1u << 22;
        if (o.path_cap_max < o.path_cap) o.path_cap_max = o.path_cap;
        if (!o.max_views) o.max_views = 256;
        if (!o.batch) o.batch = 65536;
        if (!o.wide_threshold) o.wide_threshold = 2 * o.wide_slots;
        if (!o.screen_min) o.screen_min = 2048;
        if (o.path_cap & (o.path_cap - 1)) throw LcbError("lcb_device_opts.path_cap must be a power of two");
        // needed results ahead of speculation: the stream of the synchronous launches gets the highest HIP stream priority, the side
        // lanes' streams the lowest (numerically lower = higher priority; -4 % per pass together with the early critical launch,
        // profiles/r04/ab_first.txt)
        int prioLow = 0, prioHigh = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh));
        HIP_CHECK(hipStreamCreateWithPriority(&d->stream, hipStreamNonBlocking, prioHigh));
        HIP_CHECK(hipEventCreate(&d->ev0));
        HIP_CHECK(hipEventCreate(&d->ev1));
        HIP_CHECK(hipEventCreate(&d->ev2));
        HIP_CHECK(hipEventCreate(&d->ev3));
        const LcbSegPlan& plan = d->plan;
        const uint64_t nOcc = g->nPos(), P = plan.devPositions;      // P: length of the device tables (the occurrences + the test gaps)
        {
            std::vector<uint2> lh(g->nChr());
            for (size_t c = 0; c < lh.size(); c++) lh[c] = uint2{plan.chrLo[c], plan.chrHi[c]};
            d->T.chrLoHi = d->upload(lh.data(), lh.size());
            d->T.segBase = d->upload(plan.segDev.data(), plan.segDev.size());
        }
        d->T.posId = d->uploadSeg(g->posId.data(), plan);
        d->T.posPos = d->uploadSeg(g->posPos.data(), plan);
        { const std::vector<uint32_t> win = lcb_window_table(*g, (uint32_t)p->max_branch); d->T.posWin = d->uploadSeg(win.data(), plan); }   // look-ahead windows (lcb_segments.h)
        d->T.posCh = d->uploadSeg(g->posCh.data(), plan);
        d->T.posRevCh = d->uploadSeg(g->posRevCh.data(), plan);
        if (d->seg) { d->T.occStart64 = d->upload(g->occStart.data(), g->occStart.size()); d->T.occStart32 = nullptr; }
        else {      // one segment: fewer than 2^32 occurrences
            std::vector<uint32_t> os(g->occStart.begin(), g->occStart.end());
            d->T.occStart32 = d->upload(os.data(), os.size()); d->T.occStart64 = nullptr;
        }
        {
            std::vector<uint4> rec((size_t)nOcc);
            #pragma omp parallel for schedule(static)
            for (int64_t j = 0; j < (int64_t)nOcc; j++) {
                const uint64_t q = g->occG[j];
                const uint32_t cw = plan.chrWord[g->occChr[j]];
                rec[j] = uint4{(uint32_t)(q - plan.segStart[cw >> LCB_SEG_SHIFT]), cw, g->posPos[q], (uint32_t)g->posId[q]};
            }
            d->T.occRec = d->upload(rec.data(), rec.size());
        }
        const size_t pageWords = (size_t)1 << LCB_PAGE_SHIFT;
        d->usedWords = ((size_t)(P / 32 + 2) + pageWords - 1) & ~(pageWords - 1);
        d->nPages = (uint32_t)(d->usedWords >> LCB_PAGE_SHIFT);
        // predicted `used` views for the engine's dry-run launches: a page table per view, private pages from a pool that grows on demand
        d->maxViews = (int)o.max_views;
        const uint32_t nLanes = o.side_lanes == 0xFFFFFFFFu ? 0u : (o.side_lanes ? o.side_lanes : 4u);
        d->lanePoolPages = nLanes ? 32768u : 0u;
        d->views.poolBasePage = nLanes * d->lanePoolPages;
        d->views.poolPages = 4096;
        // (+ 1 guard page: a vote walk that leaves its voter's page reads a few words past the private copy before it repairs the lane)
        HIP_CHECK(hipMalloc((void**)&d->dUsed, (d->usedWords + ((size_t)d->views.poolBasePage + d->views.poolPages + 1) * pageWords) * 4));
        HIP_CHECK(hipMemset(d->dUsed, 0, d->usedWords * 4));
        HIP_CHECK(hipMalloc((void**)&d->views.tab, (size_t)(d->maxViews + 1) * d->nPages * 4));
        HIP_CHECK(hipMemset(d->views.tab, 0, (size_t)(d->maxViews + 1) * d->nPages * 4));
        d->T.used = d->dUsed; d->T.viewTab = d->views.tab; d->T.nPages = d->nPages;
        d->T.nChr = g->nChr(); d->T.nVertex = g->nVertex; d->T.nPos = P; d->T.nSeg = plan.nSeg();
        if (d->usedWords + (((uint64_t)d->views.poolBasePage + d->views.poolPages + 1) << LCB_PAGE_SHIFT) >= (1ull << 32)) throw LcbError("the `used` bitmap and its view pages need more than 2^32 words");
        d->KP.k = p->k; d->KP.minBlock = p->min_block; d->KP.maxBranch = p->max_branch; d->KP.maxFlank = p->max_flank;
        d->KP.depth = p->looking_depth;
        HIP_CHECK(hipMalloc((void**)&d->dCursor, 32));
        HIP_CHECK(hipMemset(d->dCursor, 0, 32));
        // compact and wide: per-path state in LDS; bodies, result snapshot, checkpoint (and the compact path set) in a global slot
        WorkSet& c = d->ws[0];
        c.mode = 0; c.nSlots = o.compact_slots;
        c.pathCap = o.path_cap; c.bodyCap = c.pathCap / 2; c.bestCap = LcbCfg<0>::IC;
        d->allocWork(c);
        WorkSet& w = d->ws[1];
        w.mode = 1; w.nSlots = o.wide_slots;
        w.pathCap = o.wide_path_cap ? o.wide_path_cap : LcbCfg<1>::PC;      // the set itself lives in LDS (PC slots); the slot keeps its insertion list
        if (w.pathCap > LcbCfg<1>::PC || (w.pathCap & (w.pathCap - 1))) throw LcbError("lcb_device_opts.wide_path_cap must be a power of two, at most 8192");
        w.bodyCap = w.pathCap / 2; w.bestCap = LcbCfg<1>::IC;
        d->allocWork(w);
        // big: instance fields in the slot; huge: everything in the slot, grown on demand
        WorkSet& b = d->ws[2];
        b.mode = 2; b.nSlots = o.big_slots;
        b.pathCap = 262144; b.bodyCap = 131072; b.instCap = LcbCfg<2>::IC; b.voteCap = 0; b.bestCap = LcbCfg<2>::IC;
        d->allocWork(b);
        WorkSet& hg = d->ws[3];
        hg.mode = 3; hg.nSlots = o.huge_slots;
        hg.pathCap = 262144; hg.bodyCap = 131072; hg.instCap = 8192; hg.voteCap = 65536; hg.bestCap = 8192;
        d->allocWork(hg);
        d->batchCap = o.batch;
        HIP_CHECK(hipMalloc((void**)&d->dLive, (size_t)d->batchCap * sizeof(uint32_t)));
        HIP_CHECK(hipHostMalloc((void**)&d->hLive, ((size_t)d->batchCap + 1) * sizeof(uint32_t), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&d->hSeeds, (size_t)d->batchCap * sizeof(LcbKSeed), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&d->hOut, (size_t)d->batchCap * sizeof(LcbSeedOut), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&d->hCtr, (size_t)d->batchCap * sizeof(LcbSeedCtr), hipHostMallocDefault));
        d->allocArena(o.arena);
        d->swapBufs();
        HIP_CHECK(hipHostMalloc((void**)&d->hSeeds, (size_t)d->batchCap * sizeof(LcbKSeed), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&d->hOut, (size_t)d->batchCap * sizeof(LcbSeedOut), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&d->hLive, ((size_t)d->batchCap + 1) * sizeof(uint32_t), hipHostMallocDefault));
        d->allocArena(o.arena);
        d->swapBufs();
        const char* tf = getenv("LCB_TRACE_LAUNCHES");
        if (tf && *tf) d->traceFile = fopen(tf, "w");
        d->seedTrace = envU32("LCB_TRACE_SEEDS", 0) != 0;
        d->forceProf = envU32("LCB_FORCE_PROF", 0) != 0;
        const char* wd = getenv("LCB_WATCHDOG_S");
        d->watchdogS = wd && *wd ? atof(wd) : 0;
        if (envU32("LCB_DEBUG", 0)) {
            d->dbgSlots = std::max(std::max(c.nSlots, hg.nSlots), std::max(w.nSlots, b.nSlots));
            HIP_CHECK(hipHostMalloc((void**)&d->hDbg, (size_t)d->dbgSlots * 16 * sizeof(uint32_t), hipHostMallocDefault));
        }
        d->rangeCap = 65536;
        HIP_CHECK(hipHostMalloc((void**)&d->hRanges, (size_t)d->rangeCap * sizeof(LcbMarkRange), hipHostMallocDefault));
        // side lanes: asynchronous job batches beside the synchronous launches (engine.cpp)
        HIP_CHECK(hipStreamCreateWithFlags(&d->ctlStream, hipStreamNonBlocking));
        d->lanes.resize(nLanes);
        for (uint32_t l = 0; l < nLanes; l++) {
            SideLane& L = d->lanes[l];
            HIP_CHECK(hipStreamCreateWithPriority(&L.sw, hipStreamNonBlocking, prioLow));
            HIP_CHECK(hipStreamCreateWithPriority(&L.sb, hipStreamNonBlocking, prioLow));
            for (hipEvent_t* e : {&L.w0, &L.w1, &L.b0, &L.b1}) HIP_CHECK(hipEventCreate(e));
            L.cap = 2048; L.arenaCap = 1ull << 20;
            // the host polls these while the kernels run: fine-grained (coherent) pinned memory
            HIP_CHECK(hipHostMalloc((void**)&L.hSeeds, (size_t)L.cap * sizeof(LcbKSeed), hipHostMallocCoherent));
            HIP_CHECK(hipHostMalloc((void**)&L.hOut, (size_t)L.cap * sizeof(LcbSeedOut), hipHostMallocCoherent));
            HIP_CHECK(hipHostMalloc((void**)&L.hArena, (size_t)L.arenaCap * sizeof(uint4), hipHostMallocCoherent));
            HIP_CHECK(hipHostMalloc((void**)&L.hFp, (size_t)L.arenaCap * sizeof(LcbFpOut), hipHostMallocCoherent));
            HIP_CHECK(hipHostMalloc((void**)&L.hList, (size_t)2 * L.cap * sizeof(uint32_t), hipHostMallocCoherent));
            HIP_CHECK(hipHostMalloc((void**)&L.hCtl, 64, hipHostMallocCoherent));
            HIP_CHECK(hipMalloc((void**)&L.dCtl, 64));
            HIP_CHECK(hipMemset(L.dCtl, 0, 64));
            HIP_CHECK(hipMalloc((void**)&L.views.tab, (size_t)(d->maxViews + 1) * d->nPages * 4));
            HIP_CHECK(hipMemset(L.views.tab, 0, (size_t)(d->maxViews + 1) * d->nPages * 4));
            L.views.poolBasePage = l * d->lanePoolPages; L.views.poolPages = d->lanePoolPages;
            L.wide = d->ws[1]; L.wide.base = nullptr; L.wide.nSlots = std::max(1u, std::min(o.wide_slots, nCu / 2));
            d->allocWork(L.wide);
            L.big = d->ws[2]; L.big.base = nullptr; L.big.nSlots = std::max(1u, std::min(o.big_slots, (nCu * 3) / 8));
            d->allocWork(L.big);
            // A batch runs at most one wave of big-variant jobs (seeds known to need that variant: one workgroup per CU for tens of milliseconds,
            // 2 GB of workspace writes per launch): beyond it heavy speculation is in the way of the commit's own launches more often than
            // it is used. Config 3 -8 % per pass, the k = 25 shapes (few such jobs) unchanged; a third of that wave gains nothing
            // (profiles/r05/ab_eighth_*.txt). The jobs over the cap get no result here: the commit computes them when it needs them.
            if (!o.side_big_cap) o.side_big_cap = L.big.nSlots;
        }
    } catch (...) {
        lcb_device_destroy_impl(handle);
        throw;
    }
    return handle;
}

static void lcb_device_drop_async(lcb_device_impl* d);
static void lcb_device_drain_lanes(lcb_device_impl* d);

void lcb_device_destroy_impl(lcb_device* h)
{
    if (!h) return;
    lcb_device_impl* d = h->impl;
    if (d) {
        lcb_device_drop_async(d);
        lcb_device_drain_lanes(d);
        if (getenv("LCB_VERBOSE")) {
            fprintf(stderr, "lcb device %d: side lanes %zu: %lld batches, %lld jobs (%lld plans found no free lane, %lld did not fit a lane)\n", d->ordinal, d->lanes.size(), (long long)d->sideBatches, (long long)d->sideJobs, (long long)d->sideNoLane, (long long)d->sideNoFit);
            fprintf(stderr, "lcb device %d: seeds per variant compact %lld wide %lld big %lld huge %lld | screened %lld (dead %lld)\n", d->ordinal, (long long)d->modeSeeds[0],
                    (long long)d->modeSeeds[1], (long long)d->modeSeeds[2], (long long)d->modeSeeds[3], (long long)d->screened, (long long)d->screenedDead);
            fprintf(stderr, "   private view pages built: %lld (4 KB each; pool %u pages)\n", (long long)d->viewPagesBuilt, d->views.poolPages);
            fprintf(stderr, "   compact path set: %u vertices per slot (enlarged %d times)\n", d->ws[0].pathCap, d->compactPathGrown);
            fprintf(stderr, "   results remembered as too long for the wide variant's path set: %lld\n", (long long)d->longPathHints);
            fprintf(stderr, "   compact pools: %s (%u workgroups), %lld live seeds ended there, %lld outgrew its pools%s\n", d->compactSmall ? "128 instances / 512 vote slots" : "256 instances / 1 024 vote slots", d->ws[0].nSlots,
                    (long long)d->compactDone, (long long)d->compactPoolOvf, d->compactFellBack ? " - the small pools were given up" : "");
            fprintf(stderr, "   result arena: %llu instances (enlarged %d times)\n", d->arenaCap, d->arenaGrown);
            fprintf(stderr, "   seeds that joined a call's big launch instead of a wide launch in front of it: %lld\n", (long long)d->joinedBig);
            for (int m = 0; m < 4; m++)
                fprintf(stderr, "   overflows out of %-7s: instances %lld vote table %lld path %lld snapshot %lld\n", modeName(m), (long long)d->overflow[m][LCB_ST_INST_OVF],
                        (long long)d->overflow[m][LCB_ST_VOTE_OVF], (long long)d->overflow[m][LCB_ST_PATH_OVF], (long long)d->overflow[m][LCB_ST_BEST_OVF]);
        }
        (void)hipSetDevice(d->ordinal);
        if (d->stream) (void)hipStreamSynchronize(d->stream);
        for (void* p : d->owned) (void)hipFree(p);
        if (d->dUsed) (void)hipFree(d->dUsed);
        for (SideLane& L : d->lanes) {
            for (hipStream_t q : {L.sw, L.sb}) if (q) (void)hipStreamDestroy(q);
            for (hipEvent_t e : {L.w0, L.w1, L.b0, L.b1}) if (e) (void)hipEventDestroy(e);
            for (void* q : {(void*)L.hSeeds, (void*)L.hOut, (void*)L.hArena, (void*)L.hFp, (void*)L.hList, (void*)L.hCtl}) if (q) (void)hipHostFree(q);
            for (void* q : {(void*)L.dCtl, (void*)L.views.tab, (void*)L.views.dEntries, (void*)L.views.dVersions, (void*)L.views.dPieces, (void*)L.wide.base, (void*)L.big.base}) if (q) (void)hipFree(q);
        }
        if (d->ctlStream) (void)hipStreamDestroy(d->ctlStream);
        if (d->views.tab) (void)hipFree(d->views.tab);
        if (d->views.dEntries) (void)hipFree(d->views.dEntries);
        if (d->views.dPieces) (void)hipFree(d->views.dPieces);
        if (d->views.dVersions) (void)hipFree(d->views.dVersions);
        if (d->dCursor) (void)hipFree(d->dCursor);
        if (d->dLive) (void)hipFree(d->dLive);
        if (d->hLive) (void)hipHostFree(d->hLive);
        for (auto& w : d->ws) if (w.base) (void)hipFree(w.base);
        for (void* q : {(void*)d->hSeedsB, (void*)d->hOutB, (void*)d->hArenaB, (void*)d->hFpB, (void*)d->hLiveB}) if (q) (void)hipHostFree(q);
        if (d->ev2) (void)hipEventDestroy(d->ev2);
        if (d->ev3) (void)hipEventDestroy(d->ev3);
        if (d->hSeeds) (void)hipHostFree(d->hSeeds);
        if (d->hOut) (void)hipHostFree(d->hOut);
        if (d->hCtr) (void)hipHostFree(d->hCtr);
        if (d->hArena) (void)hipHostFree(d->hArena);
        if (d->hFp) (void)hipHostFree(d->hFp);
        if (d->hRanges) (void)hipHostFree(d->hRanges);
        if (d->hDbg) (void)hipHostFree(d->hDbg);
        if (d->traceFile) fclose(d->traceFile);
        if (d->evBase) (void)hipEventDestroy(d->evBase);
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
    lcb_device_drain_lanes(d);
    lcb_device_drop_async(d);       // (a launch begun for an overlapped round that never ended: the engine threw in between)
    if (d->views.lastViews) { HIP_CHECK(hipMemsetAsync(d->views.tab + d->nPages, 0, (size_t)d->views.lastViews * d->nPages * 4, d->stream)); d->views.lastViews = 0; }
    d->modeHint.clear();          // a new pass starts from scratch: no knowledge carried over from an earlier run
    d->recentBigFrac = 0;
    std::fill(d->hintBits.begin(), d->hintBits.end(), 0ull);
    HIP_CHECK(hipMemsetAsync(d->dUsed, 0, d->usedWords * 4, d->stream));
    HIP_CHECK(hipStreamSynchronize(d->stream));
}

void lcb_device_set_used_impl(lcb_device* h, const uint32_t* words, int64_t nWords)
{
    lcb_device_impl* d = h->impl;
    d->use();
    if ((uint64_t)nWords > d->g->nPos() / 32 + 2) throw LcbError("used bitmap has too many words");
    if (!d->plan.gap) { HIP_CHECK(hipMemcpy(d->dUsed, words, (size_t)nWords * 4, hipMemcpyHostToDevice)); return; }
    // (tests) unused table space between the segments: every segment's bits go to their place, shifted where the gap is not a multiple of 32
    HIP_CHECK(hipMemset(d->dUsed, 0, d->usedWords * 4));
    for (uint32_t sg = 0; sg < d->plan.nSeg(); sg++) {
        const uint64_t a = d->plan.segStart[sg], b = std::min<uint64_t>(d->plan.segStart[sg + 1], (uint64_t)nWords * 32), da = d->plan.segDev[sg];
        if (b <= a) continue;
        std::vector<uint32_t> buf((size_t)(((da & 31) + (b - a) + 31) / 32), 0u);
        for (uint64_t q = a; q < b; q++) if ((words[q >> 5] >> (q & 31)) & 1u) { const uint64_t t = (da & 31) + (q - a); buf[(size_t)(t >> 5)] |= 1u << (t & 31); }
        HIP_CHECK(hipMemcpy(d->dUsed + (da >> 5), buf.data(), buf.size() * 4, hipMemcpyHostToDevice));     // (the neighbouring segments are a gap away: whole words)
    }
}

void lcb_device_mark_used_impl(lcb_device* h, const uint64_t* ranges, int64_t n)
{
    lcb_device_impl* d = h->impl;
    d->use();
    const uint64_t P = d->g->nPos();
    for (int64_t i = 0; i < n; i++) if (ranges[2 * i + 1] > P) throw LcbError("used range beyond the position table");
    for (int64_t done = 0; done < n;) {
        const uint32_t m = (uint32_t)((n - done) < (int64_t)d->rangeCap ? (n - done) : d->rangeCap);
        for (uint32_t i = 0; i < m; i++) { LcbMarkRange r; d->plan.rangeToDev(ranges[2 * (done + i)], ranges[2 * (done + i) + 1], r.lo, r.hi); d->hRanges[i] = r; }   // (a range lies inside one chromosome)
        hipLaunchKernelGGL(lcb_mark_kernel, dim3(m), dim3(256), 0, d->stream, d->dUsed, d->hRanges, m);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(d->stream));   // the pinned staging buffer is reused
        done += m;
    }
}

// Predicted views 1..nViews = live state + the marks with firstView <= v (engine.cpp): views are nested, so a page that a
// mark of view v touches is private in the views v..nViews, each with the marks up to its own index. Builds them into the view
// space V on `stream` (returns after the build kernel has finished). false: the private pages do not fit a pool that cannot grow.
static bool lcb_build_views_into(lcb_device_impl* d, ViewSpace& V, hipStream_t stream, bool growable, int nViews, const LcbViewMark* marks, int64_t nMarks)
{
    if (nViews < 0 || nViews > d->maxViews) throw LcbError("more predicted views requested than the device holds");
    if (V.lastViews) HIP_CHECK(hipMemsetAsync(V.tab + d->nPages, 0, (size_t)V.lastViews * d->nPages * 4, stream));
    V.lastViews = nViews;
    if (nViews == 0 || nMarks == 0) return true;
    const uint64_t P = d->g->nPos();
    const uint32_t pageBits = 32u << LCB_PAGE_SHIFT;
    struct Piece { uint32_t page, firstView, lo, hi; };
    std::vector<Piece> pc;
    for (int64_t k = 0; k < nMarks; k++) {
        if (marks[k].hi > P || marks[k].firstView < 1 || marks[k].hi <= marks[k].lo) throw LcbError("bad predicted mark");
        uint64_t mlo, mhi;
        d->plan.rangeToDev(marks[k].lo, marks[k].hi, mlo, mhi);
        for (uint64_t a = mlo; a < mhi;) {
            const uint32_t page = (uint32_t)(a / pageBits);
            const uint64_t end = std::min<uint64_t>(mhi, (uint64_t)(page + 1) * pageBits);
            pc.push_back(Piece{page, marks[k].firstView, (uint32_t)(a - (uint64_t)page * pageBits), (uint32_t)(end - (uint64_t)page * pageBits)});
            a = end;
        }
    }
    std::sort(pc.begin(), pc.end(), [](const Piece& x, const Piece& y) { return x.page != y.page ? x.page < y.page : x.firstView < y.firstView; });
    std::vector<LcbViewPiece> pieces(pc.size());
    for (size_t i = 0; i < pc.size(); i++) pieces[i] = LcbViewPiece{pc[i].lo, pc[i].hi};
    std::vector<LcbViewPage> entries;                       // one per page that a predicted mark touches
    std::vector<LcbViewVersion> versions;                   // one per (page, view that adds marks to it)
    for (size_t i = 0; i < pc.size();) {
        size_t j = i;
        while (j < pc.size() && pc[j].page == pc[i].page) j++;
        const uint32_t verBegin = (uint32_t)versions.size();
        for (size_t a = i; a < j;) {
            size_t b = a;
            while (b < j && pc[b].firstView == pc[a].firstView) b++;
            if (!versions.empty() && versions.size() > verBegin) versions.back().nextView = pc[a].firstView;
            versions.push_back(LcbViewVersion{pc[a].firstView, (uint32_t)nViews + 1u, (uint32_t)versions.size(), (uint32_t)a, (uint32_t)b});
            a = b;
        }
        entries.push_back(LcbViewPage{pc[i].page, verBegin, (uint32_t)versions.size()});
        i = j;
    }
    if (versions.size() > V.poolPages) {
        if (!growable) { V.lastViews = 0; return false; }
        // the pools lie behind the live bitmap in one allocation (a table entry is a word offset): grow = move the live state,
        // which nobody may be reading
        lcb_device_drain_lanes(d);
        HIP_CHECK(hipStreamSynchronize(d->stream));
        uint32_t np = V.poolPages;
        while (np < versions.size()) np *= 2;
        if (d->usedWords + (((uint64_t)V.poolBasePage + np) << LCB_PAGE_SHIFT) >= (1ull << 32)) throw LcbError("predicted views need more private pages than a 32-bit word offset reaches");
        uint32_t* nu = nullptr;
        HIP_CHECK(hipMalloc((void**)&nu, (d->usedWords + (((size_t)V.poolBasePage + np + 1) << LCB_PAGE_SHIFT)) * 4));
        HIP_CHECK(hipMemcpy(nu, d->dUsed, d->usedWords * 4, hipMemcpyDeviceToDevice));
        HIP_CHECK(hipFree(d->dUsed));
        d->dUsed = nu; V.poolPages = np; d->T.used = nu;
    }
    if (entries.size() > V.entryCap) { if (V.dEntries) HIP_CHECK(hipFree(V.dEntries)); V.entryCap = entries.size() * 2; HIP_CHECK(hipMalloc((void**)&V.dEntries, V.entryCap * sizeof(LcbViewPage))); }
    if (versions.size() > V.versionCap) { if (V.dVersions) HIP_CHECK(hipFree(V.dVersions)); V.versionCap = versions.size() * 2; HIP_CHECK(hipMalloc((void**)&V.dVersions, V.versionCap * sizeof(LcbViewVersion))); }
    if (pieces.size() > V.pieceCap) { if (V.dPieces) HIP_CHECK(hipFree(V.dPieces)); V.pieceCap = pieces.size() * 2; HIP_CHECK(hipMalloc((void**)&V.dPieces, V.pieceCap * sizeof(LcbViewPiece))); }
    HIP_CHECK(hipMemcpyAsync(V.dEntries, entries.data(), entries.size() * sizeof(LcbViewPage), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(V.dVersions, versions.data(), versions.size() * sizeof(LcbViewVersion), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(V.dPieces, pieces.data(), pieces.size() * sizeof(LcbViewPiece), hipMemcpyHostToDevice, stream));
    // (the private pages of this view space start poolBasePage pages behind the live bitmap)
    hipLaunchKernelGGL(lcb_build_view_pages_kernel, dim3((uint32_t)entries.size()), dim3(256), 0, stream, d->dUsed, (uint32_t)(d->usedWords + ((size_t)V.poolBasePage << LCB_PAGE_SHIFT)), V.tab, d->nPages,
                       V.dEntries, V.dVersions, V.dPieces);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(stream));                // the host vectors above are the copy sources
    d->viewPagesBuilt += (int64_t)versions.size();
    return true;
}

void lcb_device_build_views_impl(lcb_device* h, int nViews, const LcbViewMark* marks, int64_t nMarks)
{
    lcb_device_impl* d = h->impl;
    d->use();
    lcb_build_views_into(d, d->views, d->stream, true, nViews, marks, nMarks);
}

// Three arrays of `bytes` each, `reps` timed sweeps after one warm-up; returns GB/s (3 x bytes per sweep: two reads, one write).
double lcb_device_hbm_triad_impl(lcb_device* h, uint64_t bytes, int reps)
{
    lcb_device_impl* d = h->impl;
    d->use();
    const size_t n = (size_t)(bytes / sizeof(float4));
    float4 *a = nullptr, *b = nullptr, *c = nullptr;
    HIP_CHECK(hipMalloc((void**)&a, n * sizeof(float4)));
    HIP_CHECK(hipMalloc((void**)&b, n * sizeof(float4)));
    HIP_CHECK(hipMalloc((void**)&c, n * sizeof(float4)));
    HIP_CHECK(hipMemsetAsync(b, 0, n * sizeof(float4), d->stream));
    HIP_CHECK(hipMemsetAsync(c, 0, n * sizeof(float4), d->stream));
    float ms = 0;
    for (int r = 0; r <= reps; r++) {
        if (r == 1) HIP_CHECK(hipEventRecord(d->ev0, d->stream));
        hipLaunchKernelGGL(lcb_triad_kernel, dim3(256 * 32), dim3(256), 0, d->stream, a, b, c, 3.0f, n);
    }
    HIP_CHECK(hipEventRecord(d->ev1, d->stream));
    HIP_CHECK(hipStreamSynchronize(d->stream));
    HIP_CHECK(hipEventElapsedTime(&ms, d->ev0, d->ev1));
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
    return ms > 0 ? 3.0 * (double)(n * sizeof(float4)) * reps / (ms * 1e-3) / 1e9 : 0.0;
}

int lcb_device_max_views_impl(lcb_device* h) { return h->impl->maxViews; }
int lcb_device_ordinal_impl(lcb_device* h) { return h->impl->ordinal; }
int lcb_device_concurrency_impl(lcb_device* h) { return (int)h->impl->ws[0].nSlots; }

void lcb_device_set_stats_impl(lcb_device* h, bool on) { h->impl->stats = on; }

void lcb_device_kernel_time_impl(lcb_device* h, double* ms, int64_t* launches, double* busyMs, double* sideMs)
{
    lcb_device_impl* d = h->impl;
    d->use();
    if (ms) *ms = d->kernelMs;
    if (launches) *launches = d->launches;
    if (busyMs) *busyMs = d->busyMs();
    if (sideMs) *sideMs = d->sideKernelMs;
    d->kernelMs = 0; d->sideKernelMs = 0; d->launches = 0;
    d->resetSpans();
}

int64_t lcb_device_big_retries_impl(lcb_device* h) { return h->impl->bigRetries; }
void lcb_device_mode_seeds_impl(lcb_device* h, int64_t out[4]) { for (int i = 0; i < 4; i++) out[i] = h->impl->modeSeeds[i]; }
void lcb_device_mode_time_impl(lcb_device* h, double ms[4], int64_t launches[4])
{
    // (the batches still on the lanes are accounted when they retire: a caller that wants a closed figure asks after a pass - the engine drains the lanes at its end)
    for (int i = 0; i < 4; i++) { ms[i] = h->impl->modeMs[i]; launches[i] = h->impl->modeLaunches[i]; }
}

namespace {

// The accumulators of one process() call: per-seed results arrive in launch order (retries out of seed order) and are laid
// out in seed order at the end.
struct ProcAcc {
    const lcb_seed* seeds = nullptr; const uint32_t* view = nullptr; int64_t n = 0;
    int64_t* bestScore = nullptr; lcb_counters* ctr = nullptr; std::vector<lcb_counters>* perSeedCtr = nullptr;
    bool wantFp = false;
    std::vector<lcb_instance> flat;                     // instances in arrival order
    std::vector<uint64_t> flatOff;
    std::vector<uint32_t> cnt;
    std::vector<lcb_fp> fpFlat;                         // footprints in arrival order (only when they are wanted) ...
    std::vector<uint64_t> fpAt;                         // ... and where each seed's intervals start / how many there are
    std::vector<uint32_t> fpCnt;
    std::vector<int64_t> todo[4];                       // seeds still to run, per kernel variant
    std::vector<uint8_t> tried;                         // per seed: bit m = it overflowed variant m in this call
    std::vector<lcb_seed> ownSeeds;                     // (an asynchronous call keeps its own copy of the seeds)
    bool growCompactPath = false;                       // a compact seed overflowed the path set and the set can still grow
    int64_t arenaOvf = 0;                               // seeds of the current variant's launches that found the result arena full
    std::vector<uint8_t> start;                         // first variant of every seed of the call
    bool allBig = false;                                // every seed of the call starts in the big variant (accInit)
    int64_t neededBig = 0;                              // seeds that finished in the big / huge variant and could not have run in a smaller one
};

uint64_t hintKey(const lcb_seed& sd) { return ((uint64_t)(uint32_t)sd.vid << 8) | (uint64_t)(uint8_t)sd.ch; }

void setHint(lcb_device_impl* d, const lcb_seed& sd, uint8_t mode)
{
    const uint64_t key = hintKey(sd);
    const uint32_t hb = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 48);
    d->hintBits[hb >> 6] |= 1ull << (hb & 63);
    d->modeHint[key] = mode;
}

void accInit(lcb_device_impl* d, ProcAcc& A, const lcb_seed* seeds, int64_t n, const uint32_t* view, int64_t* bestScore, lcb_counters* ctr,
             std::vector<lcb_counters>* perSeedCtr, bool wantFp)
{
    A.seeds = seeds; A.view = view; A.n = n; A.bestScore = bestScore; A.ctr = ctr; A.perSeedCtr = perSeedCtr; A.wantFp = wantFp;
    if (perSeedCtr) perSeedCtr->assign((size_t)n, lcb_counters{});
    if (bestScore) memset(bestScore, 0, (size_t)n * sizeof(int64_t));
    A.flatOff.assign((size_t)n, 0); A.cnt.assign((size_t)n, 0); A.tried.assign((size_t)n, 0);
    if (wantFp) { A.fpAt.assign((size_t)n, 0); A.fpCnt.assign((size_t)n, 0); }
    // first variant of every seed: the caller's choice, else wide for calls with few seeds (as long as their longest seed)
    // and compact for calls with many (throughput); a seed that overflowed a variant earlier in this pass starts where it ended up
    const int base = d->o.start_mode ? (int)d->o.start_mode - 1 : (n <= (int64_t)d->o.wide_threshold ? 1 : 0);
    if (d->modeHint.empty() || d->o.start_mode) { A.todo[base].resize((size_t)n); for (int64_t s = 0; s < n; s++) A.todo[base][(size_t)s] = s; }
    else {
        A.start.assign((size_t)n, (uint8_t)base);
        int64_t nBig = 0;
        for (int64_t s = 0; s < n; s++) {
            const uint64_t key = hintKey(seeds[s]);
            const uint32_t hb = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 48);
            if ((d->hintBits[hb >> 6] >> (hb & 63)) & 1ull) { auto it = d->modeHint.find(key); if (it != d->modeHint.end()) A.start[(size_t)s] = it->second; }
            nBig += A.start[(size_t)s] >= 2;
        }
        // Variants run one after the other and a launch lasts as long as its longest seed. In the stretch of the seed order
        // where seeds outgrow the wide variant (config 3: the first ~12 000 seeds are paths of 3 000 - 7 000 vertices whose pools
        // reach 1 000 - 3 600 instances) a small call that has to visit the big variant anyway is as long as its big launch:
        // everything runs there, instead of a wide launch of tens of ms in front of it (73 such launches, 3.0 s of a 22-s
        // pass, in profiles/r02/launch_trace_config3.tsv). Two triggers: at least an eighth of the call's seeds are already known
        // to need the big variant, or the last sizeable call sent at least 5 % of its seeds there.
        const bool small = n <= 2 * (int64_t)d->ws[2].nSlots;
        const bool allBig = small && (nBig * 8 >= n || (n >= 64 && d->recentBigFrac >= 0.05));
        for (int64_t s = 0; s < n; s++) A.todo[allBig && A.start[(size_t)s] < 2 ? 2 : A.start[(size_t)s]].push_back(s);
        A.allBig = allBig;
        if (allBig) d->joinedBig += n - nBig;
    }
}

void fillSeeds(lcb_device_impl* d, const ProcAcc& A, const std::vector<int64_t>& list, size_t at, uint32_t m)
{
    for (uint32_t i = 0; i < m; i++) {
        const int64_t s = list[at + i];
        d->hSeeds[i].vid = A.seeds[s].vid; d->hSeeds[i].ch = A.seeds[s].ch; d->hSeeds[i].view = A.view ? A.view[s] : 0u; d->hSeeds[i].pad = 0;
    }
}

// Takes the results of one finished launch out of the host buffers; seeds that overflowed go to the next variant's list.
// Returns true if a seed overflowed the huge variant (its capacities must grow).
bool gatherBatch(lcb_device_impl* d, ProcAcc& A, const std::vector<int64_t>& list, size_t at, uint32_t m, bool screen, int mode)
{
    bool hugeOverflow = false;
    const uint32_t nLook = screen ? d->hLive[d->batchCap] : m;   // seeds the screening finalised (no instance, no footprint) keep their zeros
    if (screen) { d->screened += m; d->screenedDead += m - nLook; }
    for (uint32_t t = 0; t < nLook; t++) {
        const uint32_t i = screen ? d->hLive[t] : t;
        const LcbSeedOut& o = d->hOut[i];
        const int64_t s = list[at + i];
        if (o.status == LCB_ST_OK) {
            if (mode == 0 && o.nFp) d->compactDone++;
            // A block whose instances span thousands of junctions was found by a path of as many vertices: when this seed is computed again - as a stop's own job
            // or in a background batch, launches that begin in the wide variant - it would fill the wide variant's LDS path set (4 096 vertices) first and start
            // over elsewhere. It is remembered like an overflow: the next attempt begins in the compact variant (a lane: in the big one).
            if (mode == 0 && o.nInst && !d->o.start_mode) {
                // (the first instance stands for all: the instances of a block span about the same number of junctions, and this loop runs for every result)
                const uint4 r0 = d->hArena[o.arenaOff];
                const uint32_t span = r0.y > r0.z ? r0.y - r0.z : r0.z - r0.y;
                if ((uint64_t)span * 8 >= (uint64_t)d->ws[1].pathCap * 3) { setHint(d, A.seeds[s], 0); d->longPathHints++; }
            }
            if (mode >= 2 && (!(A.allBig && A.start[(size_t)s] < 2) || o.poolInst > LcbCfg<1>::IC)) A.neededBig++;
            A.cnt[(size_t)s] = o.nInst;
            A.flatOff[(size_t)s] = A.flat.size();
            if (o.nInst) {
                const uint4* src = d->hArena + o.arenaOff;
                const size_t f0 = A.flat.size();
                A.flat.resize(f0 + o.nInst);
                for (uint32_t e = 0; e < o.nInst; e++) A.flat[f0 + e] = lcb_instance{src[e].x, src[e].y, src[e].z, src[e].w};
            }
            if (A.bestScore) A.bestScore[s] = o.bestScore;
            if (A.wantFp) {
                A.fpAt[(size_t)s] = A.fpFlat.size(); A.fpCnt[(size_t)s] = o.nFp;
                const LcbFpOut* src = d->hFp + o.fpOff;
                const size_t f0 = A.fpFlat.size();
                A.fpFlat.resize(f0 + o.nFp);
                for (uint32_t e = 0; e < o.nFp; e++) A.fpFlat[f0 + e] = lcb_fp{d->plan.toHost(src[e].lo), d->plan.toHost(src[e].hi)};
            }
            if (A.perSeedCtr && d->stats) {
                const LcbSeedCtr& k = d->hCtr[i];
                lcb_counters& q = (*A.perSeedCtr)[(size_t)s];
                q.n_walk = k.c[0]; q.n_occ = k.c[1]; q.n_compat_call = k.c[2]; q.n_compat_step = k.c[3];
                q.n_inst_out = k.c[4]; q.n_vote = k.c[5]; q.n_push = k.c[6]; q.n_process = k.c[7];
            }
            if (A.ctr) {
                const LcbSeedCtr& k = d->hCtr[i];
                A.ctr->n_walk += k.c[0]; A.ctr->n_occ += k.c[1]; A.ctr->n_compat_call += k.c[2]; A.ctr->n_compat_step += k.c[3];
                A.ctr->n_inst_out += k.c[4]; A.ctr->n_vote += k.c[5]; A.ctr->n_push += k.c[6]; A.ctr->n_process += k.c[7];
            }
        } else if (o.status == LCB_ST_DIST_OVF) {
            throw LcbError("a path longer than 2^31 bp is not supported");
        } else if (o.status == LCB_ST_PENDING) {
            throw LcbError("device: a seed of the launch was not processed");
        } else if (o.status == LCB_ST_ARENA_OVF) {
            A.todo[mode].push_back(s);                          // same variant again: the arena is emptied between launches
            A.arenaOvf++;
        } else {
            if (o.status < 8) d->overflow[mode][o.status]++;
            if (mode == 0 && (o.status == LCB_ST_INST_OVF || o.status == LCB_ST_VOTE_OVF)) d->compactPoolOvf++;
            A.tried[(size_t)s] |= (uint8_t)(1u << mode);
            // The next variant. The ladder is compact -> wide -> big -> huge by capacity of instances and vote table, but the
            // PATH capacity is the other way round between the first two: the wide variant keeps its path set in LDS (4096
            // vertices), the compact one in HBM (65 536). A long path with few instances (long blocks of few genomes, k = 25)
            // that started in the wide variant therefore goes back to the compact one - unless it already holds more
            // instances than half the compact pool (the repeat-rich seeds of config 3: thousands of instances AND > 4096
            // pushes; they would crawl through the compact variant only to overflow its pool) - the rest goes on to big.
            int nextMode = mode < 3 ? mode + 1 : 3;
            if (mode == 1 && o.status == LCB_ST_PATH_OVF && !(A.tried[(size_t)s] & 1u) && o.poolInst * 2 <= d->compactIC()) nextMode = 0;
            else if (mode == 0 && o.status == LCB_ST_PATH_OVF && d->ws[0].pathCap < d->o.path_cap_max && o.poolInst * 2 <= d->compactIC()) {
                // a long path of few instances: the compact slots get a larger path set (runToCompletion) and the seed runs there again
                A.growCompactPath = true;
                A.tried[(size_t)s] &= (uint8_t)~1u;
                A.todo[0].push_back(s);
                continue;
            }
            else if (mode == 0 && o.status == LCB_ST_PATH_OVF) nextMode = 2;
            else if (mode == 0 && (A.tried[(size_t)s] & 2u)) nextMode = 2;
            if (mode == 3) hugeOverflow = true; else setHint(d, A.seeds[s], (uint8_t)nextMode);
            A.todo[nextMode].push_back(s);
        }
    }
    return hugeOverflow;
}

// Launches (and waits for) everything that is still to do, variant by variant, until no seed is left.
void runToCompletion(lcb_device_impl* d, ProcAcc& A)
{
    for (int round = 0; ; round++) {
        int mode = -1;
        for (int m = 0; m < 4; m++) if (!A.todo[m].empty()) { mode = m; break; }
        if (mode < 0) break;
        if (round > 20) throw LcbError("a seed keeps overflowing the device workspaces");
        WorkSet& ws = d->ws[mode];
        std::vector<int64_t> list;
        list.swap(A.todo[mode]);
        bool hugeOverflow = false;
        for (size_t at = 0; at < list.size(); at += d->batchCap) {
            const uint32_t m = (uint32_t)std::min<size_t>(list.size() - at, d->batchCap);
            fillSeeds(d, A, list, at, m);
            if (mode >= 2) d->bigRetries += m;
            const bool screen = !d->stats && m >= d->o.screen_min;
            d->launch(ws, m, screen);
            hugeOverflow = gatherBatch(d, A, list, at, m, screen, mode) || hugeOverflow;
        }
        if (mode == 0 && d->compactSmall && d->compactAuto && d->compactDone + d->compactPoolOvf >= 65536 && d->compactPoolOvf * 8 > d->compactDone + d->compactPoolOvf) {
            // a safeguard behind the choice by the input: more than an eighth of the live seeds outgrow the small pools (and run again in the wide variant, one
            // workgroup per CU): the large pools from here on. (The head of the seed order - the vertices with the most occurrences - outgrows either pool.)
            d->compactSmall = false; d->compactFellBack++;
            d->ws[0].nSlots = std::min(d->ws[0].nSlots, d->compactLargeSlots);
        }
        if (A.growCompactPath) {
            // The compact path set starts small on purpose (the sets of all slots together stay cache-resident: 1280 x 128 KB)
            // and grows only for workloads whose paths need it (k = 25, long blocks of few genomes)
            A.growCompactPath = false;
            WorkSet& c = d->ws[0];
            c.pathCap = (uint32_t)std::min<uint64_t>((uint64_t)c.pathCap * 4, d->o.path_cap_max); c.bodyCap = c.pathCap / 2;
            d->allocWork(c);
            d->compactPathGrown++;
        } else if (A.arenaOvf && d->arenaCap < (1ull << 26)) {
            // Seeds that found the result arena full were computed for nothing: it is enlarged at the first sign (config 3 lost
            // 10 % of its kernel time to second and third launches of the same seeds with a fixed arena of 2^20 instances)
            d->allocArena(d->arenaCap * 4);
            d->arenaGrown++;
        } else
        if (!A.todo[mode].empty() && mode < 3 && A.todo[mode].size() == list.size()) d->allocArena(d->arenaCap * 4);   // not even one batch fitted
        A.arenaOvf = 0;
        if (hugeOverflow) {
            // (statuses are per seed; growing everything keeps the logic simple and this path is rare)
            WorkSet& b = d->ws[3];
            // (the reference's vectors are unbounded, path.h:683-685: the huge variant's capacities double until the memory is gone -
            // pool indices are 32-bit there; from 2^16 instances on the number of slots halves with every doubling)
            if (b.instCap >= (1u << 26)) throw LcbError("a seed needs more than 2^26 path instances: not supported by the device");
            if (d->arenaCap < (1ull << 30)) d->allocArena(d->arenaCap * 4);
            b.pathCap *= 2; b.bodyCap *= 2; b.instCap *= 2; b.voteCap *= 2; b.bestCap *= 2;
            if (b.instCap > 65536 && b.nSlots > 4) b.nSlots /= 2;
            d->allocWork(b);
        } else if (mode == 3 && !A.todo[3].empty()) d->allocArena(d->arenaCap * 4);
    }
    if (A.n >= 64 && !d->o.start_mode) d->recentBigFrac = (double)A.neededBig / (double)A.n;
}

void accLayout(ProcAcc& A, std::vector<uint64_t>& offsets, std::vector<lcb_instance>& inst, std::vector<uint64_t>* fpOffsets, std::vector<lcb_fp>* fpOut)
{
    const int64_t n = A.n;
    offsets.assign((size_t)n + 1, 0);
    uint64_t total = 0;
    for (int64_t s = 0; s < n; s++) { offsets[(size_t)s] = total; total += A.cnt[(size_t)s]; }
    offsets[(size_t)n] = total;
    inst.resize((size_t)total);
    for (int64_t s = 0; s < n; s++)
        if (A.cnt[(size_t)s]) memcpy(inst.data() + offsets[(size_t)s], A.flat.data() + A.flatOff[(size_t)s], (size_t)A.cnt[(size_t)s] * sizeof(lcb_instance));
    if (A.wantFp) {
        fpOffsets->assign((size_t)n + 1, 0);
        uint64_t tf = 0;
        for (int64_t s = 0; s < n; s++) { (*fpOffsets)[(size_t)s] = tf; tf += A.fpCnt[(size_t)s]; }
        (*fpOffsets)[(size_t)n] = tf;
        fpOut->resize((size_t)tf);
        for (int64_t s = 0; s < n; s++)
            if (A.fpCnt[(size_t)s]) memcpy(fpOut->data() + (*fpOffsets)[(size_t)s], A.fpFlat.data() + A.fpAt[(size_t)s], (size_t)A.fpCnt[(size_t)s] * sizeof(lcb_fp));
    }
}

}  // namespace

void lcb_device_process_impl(lcb_device* h, const lcb_seed* seeds, int64_t n, std::vector<uint64_t>& offsets,
                             std::vector<lcb_instance>& inst, int64_t* bestScore, lcb_counters* ctr,
                             std::vector<uint64_t>* fpOffsets, std::vector<lcb_fp>* fpOut, const uint32_t* view, std::vector<lcb_counters>* perSeedCtr)
{
    lcb_device_impl* d = h->impl;
    d->use();
    inst.clear();
    d->wantFp = fpOffsets != nullptr && fpOut != nullptr;
    ProcAcc A;
    accInit(d, A, seeds, n, view, bestScore, ctr, perSeedCtr, d->wantFp);
    try { runToCompletion(d, A); } catch (...) {
        d->wantFp = false; throw; }
    accLayout(A, offsets, inst, fpOffsets, fpOut);
    d->wantFp = false;
}

// ---- a call whose first launch runs while the host is still busy with something else --------------------------------------
// (the engine's early critical launch: the results a stop cannot go on without are computed while the host plans the rest of the
// stop's jobs). begin: the first launch a synchronous call would make (compact, wide or big by the size of the call and what is
// known about its seeds) is enqueued on the second set of host buffers, nothing is waited for, against the live `used` state of
// this moment — later marks and launches are ordered behind it on the stream; end: waits for that launch, takes its results, then
// runs whatever is left (seeds with a hint for another variant, overflows) like a normal call.
struct lcb_async_call { ProcAcc A; std::vector<int64_t> list; uint32_t m = 0; int mode = 0; bool screen = false, launched = false; };

static void lcb_device_drop_async(lcb_device_impl* d)
{
    if (d->async && d->stream) (void)hipStreamSynchronize(d->stream);
    delete d->async;
    d->async = nullptr;
}

bool lcb_device_process_begin_impl(lcb_device* h, const lcb_seed* seeds, int64_t n)
{
    lcb_device_impl* d = h->impl;
    if (d->stats || d->async || n <= 0 || n > (int64_t)d->batchCap) return false;
    if (d->hDbg || d->seedTrace || d->forceProf) return false;     // (the instrumented variants share one profile buffer)
    d->use();
    std::unique_ptr<lcb_async_call> c(new lcb_async_call());
    c->A.ownSeeds.assign(seeds, seeds + n);
    accInit(d, c->A, c->A.ownSeeds.data(), n, nullptr, nullptr, nullptr, nullptr, true);
    for (int m = 3; m >= 0; m--) if (!c->A.todo[m].empty()) c->mode = m;       // the variant a synchronous call would launch first
    c->list.swap(c->A.todo[c->mode]);
    c->m = (uint32_t)c->list.size();
    if (c->m) {
        d->wantFp = true;
        d->swapBufs();
        fillSeeds(d, c->A, c->list, 0, c->m);
        c->screen = c->m >= d->o.screen_min;
        if (c->mode >= 2) d->bigRetries += c->m;
        try { d->launch(d->ws[c->mode], c->m, c->screen, false); } catch (...) { d->swapBufs(); d->wantFp = false; throw; }
        d->swapBufs();
        d->wantFp = false;
        c->launched = true;
    }
    d->async = c.release();
    return true;
}

void lcb_device_process_end_impl(lcb_device* h, std::vector<uint64_t>& offsets, std::vector<lcb_instance>& inst, std::vector<uint64_t>& fpOffsets,
                                 std::vector<lcb_fp>& fpOut)
{
    lcb_device_impl* d = h->impl;
    if (!d->async) throw LcbError("lcb_device_process_end without a begin");
    std::unique_ptr<lcb_async_call> c(d->async);
    d->async = nullptr;
    d->use();
    inst.clear();
    d->wantFp = true;
    if (c->launched) {
        d->swapBufs();
        try {
            WorkSet& ws = d->ws[c->mode];
            const uint32_t grid = c->m < ws.nSlots ? c->m : ws.nSlots;
            d->finishLaunch(ws, c->m, grid, false, d->ev2, d->ev3);
            gatherBatch(d, c->A, c->list, 0, c->m, c->screen, c->mode);
        } catch (...) { d->swapBufs(); d->wantFp = false; throw; }
        d->swapBufs();
    }
    runToCompletion(d, c->A);
    accLayout(c->A, offsets, inst, &fpOffsets, &fpOut);
    d->wantFp = false;
}


// ---- side lanes: asynchronous job batches ---------------------------------------------------------------------------------
// A batch = the speculative jobs of one stop of the ordered commit (engine.cpp). Its jobs run in the wide variant (16 wavefronts
// per seed), the ones known to need it in the big variant, both kernels at once on the lane's two streams, against the lane's own
// predicted views. The host does not wait for the kernels: it polls the status word of the one header it needs (the kernels
// publish a header with a system-scope release after everything else the seed wrote). A job that overflows its variant gets no
// result here - the engine computes it synchronously (through the whole variant ladder) if the commit turns out to need it.

// kernel time of a finished batch goes into the device's totals; the lane is idle afterwards
static void lcb_lane_retire(lcb_device_impl* d, SideLane& L)
{
    float ms = 0;
    // (a lane runs the wide kernel on one stream and the big kernel on the other)
    if (L.ranW) { HIP_CHECK(hipEventSynchronize(L.w1)); HIP_CHECK(hipEventElapsedTime(&ms, L.w0, L.w1)); d->kernelMs += ms; d->sideKernelMs += ms; d->launches++; d->noteSpan(L.w0, L.w1); d->modeMs[1] += ms; d->modeLaunches[1]++; }
    if (L.ranB) { HIP_CHECK(hipEventSynchronize(L.b1)); HIP_CHECK(hipEventElapsedTime(&ms, L.b0, L.b1)); d->kernelMs += ms; d->sideKernelMs += ms; d->launches++; d->noteSpan(L.b0, L.b1); d->modeMs[2] += ms; d->modeLaunches[2]++; }
    L.busy = L.released = L.ranW = L.ranB = false;
}

static bool lcb_lane_finished(SideLane& L)
{
    if (L.ranW && hipEventQuery(L.w1) != hipSuccess) return false;
    if (L.ranB && hipEventQuery(L.b1) != hipSuccess) return false;
    return true;
}

static void lcb_lane_stop(lcb_device_impl* d, SideLane& L)
{
    if (!L.busy || L.released) return;
    HIP_CHECK(hipMemsetAsync(L.dCtl + 8, 0xFF, 4, d->ctlStream));    // the jobs give up at their next vote (lcb_extend)
    L.released = true;
}

static void lcb_device_drain_lanes(lcb_device_impl* d)
{
    for (SideLane& L : d->lanes) if (L.busy) lcb_lane_stop(d, L);
    if (d->ctlStream) (void)hipStreamSynchronize(d->ctlStream);
    for (SideLane& L : d->lanes) if (L.busy) { (void)hipStreamSynchronize(L.sw); (void)hipStreamSynchronize(L.sb); lcb_lane_retire(d, L); }
}

int lcb_device_side_begin_impl(lcb_device* h, const lcb_seed* seeds, const uint32_t* view, int64_t n, int nViews, const LcbViewMark* marks, int64_t nMarks)
{
    lcb_device_impl* d = h->impl;
    if (d->lanes.empty() || d->stats || d->o.start_mode || n <= 0 || d->hDbg || d->seedTrace || d->forceProf) return -1;
    d->use();
    int lane = -1;
    for (size_t l = 0; l < d->lanes.size() && lane < 0; l++) if (!d->lanes[l].busy) lane = (int)l;
    for (size_t l = 0; l < d->lanes.size() && lane < 0; l++) if (d->lanes[l].released && lcb_lane_finished(d->lanes[l])) { HIP_CHECK(hipStreamSynchronize(d->ctlStream)); lcb_lane_retire(d, d->lanes[l]); lane = (int)l; }
    for (size_t l = 0; l < d->lanes.size() && lane < 0; l++)
        if (d->lanes[l].released) {       // told to stop: its jobs leave at their next vote
            HIP_CHECK(hipStreamSynchronize(d->ctlStream));
            HIP_CHECK(hipStreamSynchronize(d->lanes[l].sw)); HIP_CHECK(hipStreamSynchronize(d->lanes[l].sb));
            lcb_lane_retire(d, d->lanes[l]); lane = (int)l;
        }
    if (lane < 0) { d->sideNoLane++; return -1; }
    if (n > (int64_t)d->lanes[(size_t)lane].cap) { d->sideNoFit++; return -2; }      // (-2: no lane can take this batch, whatever gives way)
    SideLane& L = d->lanes[(size_t)lane];
    if (!lcb_build_views_into(d, L.views, L.sw, false, nViews, marks, nMarks)) { d->sideNoFit++; return -2; }
    // jobs by variant: big for the seeds known to need it, wide for the rest (ticket order = plan order within a kernel)
    uint32_t nW = 0, nB = 0;
    uint32_t* listW = L.hList; uint32_t* listB = L.hList + L.cap;
    L.seeds.assign(seeds, seeds + n);
    for (int64_t i = 0; i < n; i++) {
        L.hSeeds[i].vid = seeds[i].vid; L.hSeeds[i].ch = seeds[i].ch; L.hSeeds[i].view = view ? view[i] : 0u; L.hSeeds[i].pad = 0;
        L.hOut[i].status = LCB_ST_PENDING;
        uint8_t mode = 1;
        if (!d->modeHint.empty()) {
            const uint64_t key = hintKey(seeds[i]);
            const uint32_t hb = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 48);
            if ((d->hintBits[hb >> 6] >> (hb & 63)) & 1ull) { auto it = d->modeHint.find(key); if (it != d->modeHint.end()) mode = it->second; }
        }
        if (mode >= 3) L.hOut[i].status = LCB_ST_ABORTED;           // the huge variant does not run here: no result
        // (hint 0: a path too long for the wide variant's LDS path set - a lane has no compact kernel, so it runs in the big one)
        else if ((mode == 2 || mode == 0) && d->o.side_big_cap != 0xFFFFFFFFu && nB >= d->o.side_big_cap) L.hOut[i].status = LCB_ST_ABORTED;   // more heavy speculation than the lane's cap: no result (the plan's order is the order of need)
        else if (mode == 2 || mode == 0) listB[nB++] = (uint32_t)i;
        else listW[nW++] = (uint32_t)i;
    }
    memset(L.hCtl, 0, 64);
    L.hCtl[6] = nW; L.hCtl[7] = nB;
    HIP_CHECK(hipMemcpyAsync(L.dCtl, L.hCtl, 64, hipMemcpyHostToDevice, L.sw));
    HIP_CHECK(hipStreamSynchronize(L.sw));          // the control words are in place before either kernel starts
    LcbTables T = d->T;
    T.viewTab = L.views.tab;
    auto start = [&](WorkSet& w, hipStream_t q, hipEvent_t e0, hipEvent_t e1, uint32_t m, uint32_t* list, uint32_t ticketWord, uint32_t countWord) {
        LcbWork W;
        W.base = w.base; W.slotBytes = w.slotBytes; W.pathCap = w.pathCap; W.bodyCap = w.bodyCap; W.bestCap = w.bestCap; W.instCap = w.instCap; W.voteCap = w.voteCap;
        W.cursor = L.dCtl + ticketWord; W.cursorBase = 0; W.live = list; W.nLive = L.dCtl + countWord;
        W.arenaCursor = (unsigned long long*)(L.dCtl + 2); W.arenaBase = 0; W.fpCursor = (unsigned long long*)(L.dCtl + 4); W.fpBase = 0;
        W.ctr = nullptr; W.dbg = nullptr; W.abort = L.dCtl + 8;
        const uint32_t grid = m < w.nSlots ? m : w.nSlots;
        HIP_CHECK(hipEventRecord(e0, q));
        if (w.mode == 2 && d->seg) hipLaunchKernelGGL((lcb_process_kernel<2, false, LCB_NW_BIG, false, true>), dim3(grid), dim3(64 * LCB_NW_BIG), 0, q, T, d->KP, L.hSeeds, (uint32_t)n, W, L.hOut, L.hArena, L.arenaCap, L.hFp, L.arenaCap);
        else if (w.mode == 2) hipLaunchKernelGGL((lcb_process_kernel<2, false, LCB_NW_BIG, false, false>), dim3(grid), dim3(64 * LCB_NW_BIG), 0, q, T, d->KP, L.hSeeds, (uint32_t)n, W, L.hOut, L.hArena, L.arenaCap, L.hFp, L.arenaCap);
        else if (d->seg) hipLaunchKernelGGL((lcb_process_kernel<1, false, LCB_NW_WIDE, false, true>), dim3(grid), dim3(64 * LCB_NW_WIDE), 0, q, T, d->KP, L.hSeeds, (uint32_t)n, W, L.hOut, L.hArena, L.arenaCap, L.hFp, L.arenaCap);
        else hipLaunchKernelGGL((lcb_process_kernel<1, false, LCB_NW_WIDE, false, false>), dim3(grid), dim3(64 * LCB_NW_WIDE), 0, q, T, d->KP, L.hSeeds, (uint32_t)n, W, L.hOut, L.hArena, L.arenaCap, L.hFp, L.arenaCap);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(e1, q));
        d->modeSeeds[w.mode] += m;
    };
    L.ranW = nW > 0; L.ranB = nB > 0;
    if (nW) start(L.wide, L.sw, L.w0, L.w1, nW, listW, 0, 6);
    if (nB) start(L.big, L.sb, L.b0, L.b1, nB, listB, 1, 7);
    L.busy = true; L.released = false; L.n = n;
    d->sideBatches++; d->sideJobs += n;
    return lane;
}

int lcb_device_side_poll_impl(lcb_device* h, int lane, int64_t k, bool wait, std::vector<lcb_instance>& inst, std::vector<lcb_fp>& fp)
{
    lcb_device_impl* d = h->impl;
    SideLane& L = d->lanes[(size_t)lane];
    if (!L.busy || k < 0 || k >= L.n) return 2;
    volatile uint32_t* st = &L.hOut[k].status;
    if (*st == LCB_ST_PENDING) {
        if (!wait) return 0;
        d->use();
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t spin = 0; *st == LCB_ST_PENDING; spin++) {
            if ((spin & 0xFFF) != 0xFFF) continue;
            if (lcb_lane_finished(L) && *st == LCB_ST_PENDING) throw LcbError("side lane: the kernels finished without processing a job");
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (d->watchdogS > 0 && el > d->watchdogS) throw LcbError("kernel watchdog expired while waiting for a side-lane job (LCB_WATCHDOG_S)");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const LcbSeedOut o = L.hOut[k];
    if (o.status == LCB_ST_OK) {
        const uint4* src = L.hArena + o.arenaOff;
        for (uint32_t e = 0; e < o.nInst; e++) inst.push_back(lcb_instance{src[e].x, src[e].y, src[e].z, src[e].w});
        const LcbFpOut* fs = L.hFp + o.fpOff;
        for (uint32_t e = 0; e < o.nFp; e++) fp.push_back(lcb_fp{d->plan.toHost(fs[e].lo), d->plan.toHost(fs[e].hi)});
        return 1;
    }
    if (o.status == LCB_ST_DIST_OVF) throw LcbError("a path longer than 2^31 bp is not supported");
    // an overflow: the next attempt (synchronous, or a later batch) starts in the variant that holds it
    if (o.status >= LCB_ST_INST_OVF && o.status <= LCB_ST_BEST_OVF) {
        const bool wasBig = [&]() { for (uint32_t q = 0; q < L.hCtl[7]; q++) if (L.hList[L.cap + q] == (uint32_t)k) return true; return false; }();
        setHint(d, L.seeds[(size_t)k], (uint8_t)(wasBig ? 3 : 2));
        d->overflow[wasBig ? 2 : 1][o.status]++;
    }
    return 2;
}

void lcb_device_side_release_impl(lcb_device* h, int lane)
{
    lcb_device_impl* d = h->impl;
    d->use();
    lcb_lane_stop(d, d->lanes[(size_t)lane]);
}

int lcb_device_side_lanes_impl(lcb_device* h) { return (int)h->impl->lanes.size(); }

namespace {

// The product's per-rank engine: the HIP kernels on one MI355X.
struct DeviceProcessor : LcbProcessor {
    lcb_device* dev;
    explicit DeviceProcessor(lcb_device* d) : dev(d) {}
    void process(const lcb_seed* seeds, const uint32_t* view, int64_t n, std::vector<uint64_t>& off, std::vector<lcb_instance>& inst,
                 std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) override
    {
        lcb_device_process_impl(dev, seeds, n, off, inst, nullptr, nullptr, &fpOff, &fp, view, ctrSink);
    }
    int maxViews() const override { return lcb_device_max_views_impl(dev); }
    int concurrency() const override { return lcb_device_concurrency_impl(dev); }
    void buildViews(int nViews, const LcbViewMark* marks, int64_t nMarks) override { lcb_device_build_views_impl(dev, nViews, marks, nMarks); }
    bool processBegin(const lcb_seed* seeds, int64_t n) override { return lcb_device_process_begin_impl(dev, seeds, n); }
    void processEnd(std::vector<uint64_t>& off, std::vector<lcb_instance>& inst, std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) override
    {
        lcb_device_process_end_impl(dev, off, inst, fpOff, fp);
    }
    void mark(const uint64_t* ranges, int64_t n) override { lcb_device_mark_used_impl(dev, ranges, n); }
    void reset() override { lcb_device_reset_used_impl(dev); }
    int sideLanes() const override { return lcb_device_side_lanes_impl(dev); }
    int sideBegin(const lcb_seed* seeds, const uint32_t* view, int64_t n, int nViews, const LcbViewMark* marks, int64_t nMarks) override
    {
        return lcb_device_side_begin_impl(dev, seeds, view, n, nViews, marks, nMarks);
    }
    int sidePoll(int lane, int64_t k, bool wait, std::vector<lcb_instance>& inst, std::vector<lcb_fp>& fp) override { return lcb_device_side_poll_impl(dev, lane, k, wait, inst, fp); }
    void sideRelease(int lane) override { lcb_device_side_release_impl(dev, lane); }
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
    lcb_device_drain_lanes(dev->impl);      // (the last batches are released but not retired: their kernel time belongs to this pass)
    if (getenv("LCB_VERBOSE"))
        fprintf(stderr, "lcb engine: %.0f ms = processor %.0f + dry runs %.0f (marks to the device %.0f, simulation %.0f; %lld views) + rest %.0f (round setup %.0f, validation %.0f, commit %.0f, marks to the device %.0f)\n", es.wallMs, es.processMs, es.planMs, es.sectionMs[LCB_SEC_PLAN_FLUSH], es.sectionMs[LCB_SEC_PLAN_SIM], (long long)es.viewsBuilt, es.wallMs - es.processMs - es.planMs, es.sectionMs[LCB_SEC_SETUP],
                es.sectionMs[LCB_SEC_VALIDATE], es.sectionMs[LCB_SEC_COMMIT], es.sectionMs[LCB_SEC_FLUSH]);
    if (stats) {
        double ms = 0, busy = 0, side = 0; int64_t l = 0;
        lcb_device_kernel_time_impl(dev, &ms, &l, &busy, &side);
        stats->kernel_busy_ms = busy; stats->kernel_side_ms = side;
        stats->seeds = nSeeds; stats->blocks_found = es.blocksFound; stats->failures = es.failures;
        stats->launches = l; stats->kernel_ms = ms; stats->big_retries = lcb_device_big_retries_impl(dev) - retries0;
        stats->wall_ms = es.wallMs;
        stats->rounds = es.rounds; stats->recompute_launches = es.recomputeLaunches; stats->recomputed_seeds = es.recomputedSeeds;
        stats->conflict_launches = es.conflictLaunches; stats->conflict_seeds = es.conflictSeeds; stats->exchanges = es.exchanges;
        stats->jobs_used = es.jobsUsed; stats->views_built = es.viewsBuilt; stats->over_predicted = es.overPredicted;
        stats->process_ms = es.processMs; stats->plan_ms = es.planMs; stats->events = es.events;
        stats->side_batches = es.sideBatches; stats->side_jobs = es.sideJobs; stats->side_taken = es.sideTaken; stats->side_void = es.sideVoid; stats->side_failed = es.sideFailed;
        stats->early_critical = es.earlyCritical;
        stats->lazy_seeds = es.lazySeeds; stats->host_dead = es.hostDead; stats->collectives = es.collectives;
    }
}
