// lcb_kernel.h — gfx950 device code of the per-seed path-extension / bubble-scoring hot path.
//
// One seed (BlocksFinder::Bundle) per workgroup. Wavefront 0 runs the per-seed algorithm; the other wavefronts of the
// workgroup (NW - 1 of them, none in the compact mode) share the look-ahead vote: its voter walks, its arg-max and the
// clearing of the vote table. This replaces ProcessVertex::Process (blocksfinder.h:228-310) and everything under it:
// MostPopularVertex (blocksfinder.h:708-768), ExtendPathForward/Backward (:770-895), Path::Init,
// PointPushBack/Front + workers, Compatible, Score, Clear (path.h:33-46,380-677) and
// DistanceKeeper (distancekeeper.h:9-41). It is a new design, not a translation:
//
//  * tables are structure-of-arrays over a FLAT position index (chromosome after chromosome), so a
//    look-ahead walk is a coalesced read of consecutive posId/posPos words (lanes = walk steps). A position is a pair
//    (segment, g): a segment is a run of whole chromosomes with fewer than 2^32 positions, g a 32-bit offset inside it, and the
//    flat index is segBase[segment] + g. A chromosome never leaves its segment, so everything a path instance does - walks,
//    gaps, ordering inside a chromosome - is 32-bit arithmetic on g; 64 bits appear where a table or the bitmap is addressed.
//    Inputs of fewer than 2^32 positions are ONE segment with base 0: their kernels (SEG = false) have no segment code at all;
//  * the per-chromosome std::multiset<Instance> becomes one key-sorted index array over an
//    insertion-ordered instance pool (pool order IS allInstance_ order) in LDS;
//  * the dense per-thread vote array count[2V+1] becomes an LDS hash table filled with
//    ds atomics; the order-dependent running arg-max of the reference is restated order-free
//    (max count, then smallest origin of the last contributing instance in list order, then
//    earliest walk step — SURVEY.md Q7) so all (instance, step) pairs can vote concurrently; the voters of a vote
//    (instances that end at the path end) are found lane-parallel, 64 list entries per pass;
//  * the dense DistanceKeeper int[2V] becomes a per-wave open-addressing vertex set in global
//    memory behind an LDS Bloom filter; distances are carried by the instances (an instance's back/front
//    distance IS the path distance of its end vertex), so no distance lookups remain;
//  * the edges between the origin of a vote and the chosen vertex are fetched as one batch (lanes = edges: table rows
//    and CSR bounds of all of them at once), and the occurrence records of edge j+1 are in flight while edge j is pushed;
//  * a push evaluates all occurrences of the pushed vertex lane-parallel against the pre-push
//    state and resolves the (rare) occurrences that fall into the same gap between two
//    instances with a closed-form prefix rule that reproduces the sequential semantics;
//  * Compatible's unbounded `used` walk (path.h:387-393) becomes a masked bitmap range test
//    evaluated after the distance tests (it is a pure function);
//  * a seed reads the `used` VIEW its record names (the live bitmap or a predicted state built by the engine,
//    engine.cpp) and reports the FOOTPRINT of the bits it read as 0, which is what makes speculation exact;
//  * the replay of the forward extension (blocksfinder.h:271-284) restarts at a checkpoint taken at a best point.
//
// Wave-uniform values are kept in scalar registers explicitly (lcb_rfl / lcb_rl): what a lane loads from a uniform
// address is uniform, but the compiler cannot know that, and per-lane copies of uniform state turn every branch into
// exec-mask bookkeeping. All cross-lane operations sit in wave-uniform control flow.
// Integer arithmetic only; no MFMA — the work is indexing, not contraction.
#ifndef LCB_KERNEL_H
#define LCB_KERNEL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcb_kernel_limits.h"

#define LCB_EMPTY_KEY INT32_MIN
// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15), see lcb_vote_walk
#if defined(__HIP_DEVICE_COMPILE__)
#define LCB_NO_LOADS_IN_FLIGHT() __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define LCB_NO_LOADS_IN_FLIGHT() ((void)0)
#endif
// A pointer the compiler cannot prove to be a global one (it was merged with a null) is dereferenced with FLAT loads, which wait on
// two counters and are slower; this says what it is. (Device compiler only: the CPU emulator of the tests sees a plain pointer.)
#if defined(__HIP_DEVICE_COMPILE__)
#define LCB_GLOBAL_U32(p) ((const __attribute__((address_space(1))) uint32_t*)(p))
#else
#define LCB_GLOBAL_U32(p) (p)
#endif
// The flight-recorder sites (LCB_MARK) are compiled into every kernel variant and cost one predictable branch each
// when the recorder is off. The recorder is what localises a hang on the device (host watchdog, device.hip).
// Diagnostic build (-DLCB_PROF_PUSH=1, instrumented variant only): the vote-section slots of the per-seed profile carry the sections of a
// push instead - [8] until the vertex is in the path set, [9] search + classification of the occurrences, [10] cross-lane rule + apply,
// [11] rebuilding the ordered index.
#ifndef LCB_PROF_PUSH
#define LCB_PROF_PUSH 0
#endif

// Four kernel variants by where the per-path state lives and how many seeds share a CU. Seeds that overflow one are
// re-run by the host in the next:
//   mode 0 "compact": 256 instances / 1024 vote slots in 32 KB of LDS -> 5 two-wave workgroups per CU (throughput; with 512
//                     vote slots 5 % of the seeds of the 62-strain workload overflowed into the wide variant's 256 slots);
//                     path vertex set in the global workspace behind an LDS Bloom filter
//   mode 1 "wide":    1024 / 2048 and the path vertex set (8192 slots) in 140 KB of LDS, 16 wavefronts share the votes
//                     -> 1 workgroup per CU (latency)
//   mode 2 "big":     4096 / 4096: ordered index, lists and vote table in LDS, the instance fields in the global workspace
//   mode 3 "huge":    everything in the global workspace, capacities chosen (and grown) by the host
// IC instances, VC vote slots, BW Bloom words (0 = none), PC path-set slots in LDS (0 = global workspace)
template <int MODE> struct LcbCfg;
// Idx: a pool index / a vote-table slot (16 bits where the capacities are compile-time constants of a few thousand; 32 bits in the huge
// variant, whose capacities grow with the path - the reference's vectors are unbounded, path.h:683-685); VLast: (ordinal of the last
// contributing instance in the voting list, step) of a vote-table entry - 16 + 16 bits, or 32 + 32.
template <> struct LcbCfg<0> { static constexpr uint32_t IC = 256, VC = 1024, BW = 256, PC = 0; static constexpr bool INST_LDS = true, IDX_LDS = true; typedef uint16_t Idx; typedef uint32_t VLast; };
template <> struct LcbCfg<1> { static constexpr uint32_t IC = 1024, VC = 2048, BW = 0, PC = 8192; static constexpr bool INST_LDS = true, IDX_LDS = true; typedef uint16_t Idx; typedef uint32_t VLast; };
template <> struct LcbCfg<2> { static constexpr uint32_t IC = 4096, VC = 4096, BW = 2048, PC = 0; static constexpr bool INST_LDS = false, IDX_LDS = true; typedef uint16_t Idx; typedef uint32_t VLast; };
// 4 = the compact variant with half the pools (128 instances / 512 vote slots: 17 KB of LDS instead of 32 KB - 8 workgroups per CU, bound by
// registers, instead of 5): where the vertices have few occurrences (k = 25 inputs of 8-16 genomes: paths of ~16 instances) the round launches
// of a Gbp-scale input are bound by the number of seeds in flight (profiles/r06). For the host it is variant 0 (lcb_device_opts.compact_pools).
template <> struct LcbCfg<4> { static constexpr uint32_t IC = 128, VC = 512, BW = 256, PC = 0; static constexpr bool INST_LDS = true, IDX_LDS = true; typedef uint16_t Idx; typedef uint32_t VLast; };
template <> struct LcbCfg<3> { static constexpr uint32_t IC = 1, VC = 1, BW = 2048, PC = 0; static constexpr bool INST_LDS = false, IDX_LDS = false; typedef uint32_t Idx; typedef unsigned long long VLast; };

enum LcbStatus : uint32_t {
    LCB_ST_OK = 0,
    LCB_ST_INST_OVF = 1,   // instance pool full
    LCB_ST_VOTE_OVF = 2,   // vote table full
    LCB_ST_PATH_OVF = 3,   // path vertex set / body full
    LCB_ST_BEST_OVF = 4,   // result snapshot buffer full
    LCB_ST_ARENA_OVF = 5,  // batch result arena full
    LCB_ST_DIST_OVF = 6,   // path distance does not fit 32 bits (unsupported, > 2 Gbp paths)
    LCB_ST_ABORTED = 7,    // an asynchronous job batch was told to stop (LcbWork::abort): no result
    LCB_ST_PENDING = 0xFFFFFFFFu,   // header written by the screening kernel for a live seed (the process kernel overwrites it)
};

// Positions: a segment is a run of whole chromosomes with fewer than 2^32 junction occurrences in total; (segment, g) names the
// occurrence segBase[segment] + g of the flat tables. The reference bounds a CHROMOSOME by 2^32 (junctionstorage.h:120-151: uint32_t
// idx / pos per chromosome) and the number of chromosomes not at all; so do the tables here (up to LCB_MAX_SEG segments).
// A "chromosome word" cw = (segment << LCB_SEG_SHIFT) | chromosome travels with every occurrence record and every instance.
struct LcbTables {
    const uint2* chrLoHi;       // [nChr]  the chromosome's positions inside its segment: g in [x, y)
    const int32_t* posId;       // [nPos]  Position::id                  (flat index = segBase[segment] + g)
    const uint32_t* posPos;     // [nPos]  Position::pos
    const uint32_t* posWin;     // [nPos]  look-ahead window: steps forward (low 16 bits) / backward (high 16) within maxBranch bp inside the chromosome (lcb_window_table)
    const uint8_t* posCh;       // [nPos]  seq[pos + k]              (JunctionSequentialIterator::GetChar, + strand)
    const uint8_t* posRevCh;    // [nPos]  ReverseChar(seq[pos - 1]) or 'N' at pos 0   (- strand)
    const uint32_t* occStart32; // [nVertex+1] CSR over |vertex id| (one segment: fewer than 2^32 occurrences) ...
    const uint64_t* occStart64; // ... or 64-bit (SEG kernels); the other pointer is null
    const uint4* occRec;        // [occurrences]  per occurrence, ascending in (segment, g): {g, cw, Position::pos, Position::id} (one 16-B load)
    const uint64_t* segBase;    // [LCB_MAX_SEG] flat index of g = 0 of every segment (one segment: {0})
    const uint32_t* used;       // bitmap over the flat index: bit = Position::used of (chr, idx) — the LIVE state (view 0)
    // Predicted views 1.. of it (engine.cpp) are copy-on-write at a granularity of 4-KB pages (1024 words, 32768 positions).
    // The private pages live BEHIND the live bitmap in the same allocation, and viewTab[v * nPages + page] is the word offset
    // from a page of the live bitmap to view v's copy of it (0 if the view shares the live page): word w of view v is
    // used[w + viewTab[v * nPages + (w >> 10)]].
    const uint32_t* viewTab;
    uint32_t nPages;
    uint32_t nChr, nVertex, nSeg;
    uint64_t nPos;              // length of the flat tables
};

// The `used` state one seed reads: the live bitmap, or a predicted view of it through that view's page table.
struct LcbUsed { const uint32_t* live; const uint32_t* tab; };
#ifndef LCB_PAGE_SHIFT
#define LCB_PAGE_SHIFT 10u       // words per page = 1024 (the emulator tests also build with tiny pages, so that views span many)
#endif

struct LcbKParams { int32_t k, minBlock, maxBranch, maxFlank, depth; };
struct LcbKSeed { int32_t vid; int32_t ch; uint32_t view; uint32_t pad; };   // view: which `used` view this seed reads

struct LcbSeedOut {            // per-seed header written by the kernels (40 B)
    uint32_t nInst;
    uint32_t status;
    int64_t bestScore;
    uint64_t arenaOff;
    uint64_t fpOff;            // first footprint interval of this seed in the footprint arena
    uint32_t nFp;              // number of footprint intervals (= instances ever created)
    uint32_t poolInst;         // instances in the pool when the seed ended (on an overflow: what the next variant has to hold at least)
};
struct LcbFpOut { unsigned long long lo, hi; };   // one footprint interval: flat positions [lo, hi]
struct LcbSeedCtr { uint64_t c[16]; };  // [0..8): lcb_counters order in stats mode, a cheap profile in the instrumented variant; [8..16): vote sections (instrumented)

struct LcbWork {               // per-workgroup global-memory workspace slots + the work queue of one launch
    uint8_t* base;
    uint64_t slotBytes;
    uint32_t pathCap;          // power of two
    uint32_t bodyCap;
    uint32_t bestCap;
    uint32_t instCap;          // big mode only
    uint32_t voteCap;          // big mode only, power of two
    uint32_t* cursor;          // work-queue head: monotone ticket counter, never reset ...
    uint32_t cursorBase;       // ... tickets of this launch are [cursorBase, cursorBase + nTickets)
    const uint32_t* live;      // ticket -> seed index (written by lcb_screen_kernel), or null: ticket == seed index
    const uint32_t* nLive;     // number of tickets when `live` is set
    unsigned long long* arenaCursor;   // monotone result-arena allocator
    unsigned long long arenaBase;
    unsigned long long* fpCursor;      // monotone footprint-arena allocator
    unsigned long long fpBase;
    LcbSeedCtr* ctr;           // per-seed counters (stats / instrumented variants), or null
    uint32_t* dbg;             // optional flight recorder: 16 words per workgroup (host watchdog prints them), or null
    const uint32_t* abort;     // asynchronous job batches (device.hip, side lanes): when the word becomes non-zero the seeds give up at their next vote
};

// ---- workspace layout (shared by host and device) -------------------------------------------
struct LcbSlotLayout {
    uint64_t pKeys, pSlots, body, best, ck;                   // always (ck: forward-extension checkpoint, 6 words per instance)
    uint64_t inst, fp;                                          // big and huge modes
    uint64_t ordKey, ordIdx, good, goodPos, touch, vKey, vCount, vLast, vTouched;   // huge mode
    uint64_t total;
};
__host__ __device__ inline uint64_t lcb_align16(uint64_t x) { return (x + 15) & ~15ull; }
__host__ __device__ inline LcbSlotLayout lcb_slot_layout(uint32_t pathCap, uint32_t bodyCap, uint32_t bestCap,
                                                         uint32_t instCap, uint32_t voteCap)
{
    LcbSlotLayout L;
    uint64_t o = 0;
    L.pKeys = o; o = lcb_align16(o + 4ull * pathCap);
    L.pSlots = o; o = lcb_align16(o + 4ull * (pathCap / 2 + 1));
    L.body = o; o = lcb_align16(o + 8ull * bodyCap);
    L.best = o; o = lcb_align16(o + 16ull * bestCap);
    L.ck = o; o = lcb_align16(o + 6ull * 4 * bestCap + 4ull * (LCB_MAX_SEG + 1));   // (+ the per-segment cursors of the ordered index)
    L.inst = o; o = lcb_align16(o + 9ull * 4 * instCap);
    L.fp = o; o = lcb_align16(o + 3ull * 4 * instCap);                                // lo, hi, segment
    const uint32_t idxCap = voteCap ? instCap : 0;             // the index / list / vote arrays only exist in huge mode
    L.ordKey = o; o = lcb_align16(o + 2ull * 4 * idxCap);
    L.ordIdx = o; o = lcb_align16(o + 2ull * 4 * idxCap);          // (the huge variant's indices are 32-bit: LcbCfg<3>::Idx)
    L.good = o; o = lcb_align16(o + 4ull * idxCap);
    L.goodPos = o; o = lcb_align16(o + 4ull * idxCap);
    L.touch = o; o = lcb_align16(o + 4ull * idxCap);
    L.vKey = o; o = lcb_align16(o + 4ull * voteCap);
    L.vCount = o; o = lcb_align16(o + 4ull * voteCap);
    L.vLast = o; o = lcb_align16(o + 8ull * voteCap);
    L.vTouched = o; o = lcb_align16(o + 4ull * voteCap);
    L.total = lcb_align16(o);
    return L;
}

// ---- wave primitives ---------------------------------------------------------------------------
// Orders LDS/global accesses between the lanes of one wavefront.
#define LCB_WAVE_SYNC()                                           \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");    \
    } while (0)

// The same for data that lives in LDS and is only touched by this wavefront between two workgroup barriers: LDS operations
// of one wavefront execute in issue order, so only the compiler has to be kept from moving them; unlike the fence above
// this does not wait for the global loads in flight (prefetched occurrence records, `used` words), which is the point.
#define LCB_WAVE_SYNC_LDS()                                       \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)
// per kernel variant: the light form where the protected state is LDS-resident, the full fence otherwise
#define LCB_SYNC_IF(ldsResident) do { if (ldsResident) LCB_WAVE_SYNC_LDS(); else LCB_WAVE_SYNC(); } while (0)

// Flight recorder: lane 0 stores progress words the host watchdog can read while the kernel is running.
#define LCB_MARK(S, slot, value)                                                         \
    do {                                                                                 \
        if ((S).dbg && (S).lane == 0) __hip_atomic_store(&(S).dbg[(slot)], (uint32_t)(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
    } while (0)

// wave-uniform value -> scalar register (every lane must hold the same value; uniform control flow only)
__device__ __forceinline__ uint32_t lcb_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int32_t lcb_rfl(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t lcb_rfl(uint64_t v) { return ((uint64_t)lcb_rfl((uint32_t)(v >> 32)) << 32) | lcb_rfl((uint32_t)v); }
// value of lane `l` (l wave-uniform) -> scalar register
__device__ __forceinline__ uint32_t lcb_rl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t lcb_rl(uint64_t v, uint32_t l) { return ((uint64_t)lcb_rl((uint32_t)(v >> 32), l) << 32) | lcb_rl((uint32_t)v, l); }
__device__ __forceinline__ int32_t lcb_rl(int32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, (int)l); }

// Wave-wide max / min of a 32-bit value as a scalar: four row-shift steps inside the 16-lane rows, two row broadcasts,
// result in lane 63 (DPP, no LDS traffic).
__device__ __forceinline__ uint32_t lcb_wave_umax(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;   // row_shr:1
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;   // row_shr:2
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;   // row_shr:4
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;   // row_shr:8
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;   // row_bcast:15
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;   // row_bcast:31
    return lcb_rl(v, 63);
}
__device__ __forceinline__ uint32_t lcb_wave_umin(uint32_t v) { return ~lcb_wave_umax(~v); }

#define LCB_FLAG_POS 1u
#define LCB_FLAG_BACKFIN 2u
#define LCB_FLAG_FRONTFIN 4u
#define LCB_FLAG_BITS 3u       // iFlags = (chromosome word << 3) | flags   (chromosome word: segment << LCB_SEG_SHIFT | chromosome)
#define LCB_FLAG_SEG_SHIFT (LCB_SEG_SHIFT + LCB_FLAG_BITS)

template <bool WIDE> struct LcbWidth { typedef uint32_t T; };
template <> struct LcbWidth<true> { typedef uint64_t T; };

template <int MODE_, bool SEG_>
struct LcbStateT {
    static constexpr int MODE = MODE_;
    typedef typename LcbCfg<MODE_>::Idx Idx;
    typedef typename LcbCfg<MODE_>::VLast VLast;
    static constexpr uint32_t NONE = (uint32_t)(Idx)~(Idx)0;      // "no position" in goodPos
    static constexpr uint32_t LAST_SHIFT = 4u * sizeof(VLast);     // bits of the step in a VLast
    static constexpr bool SEG = SEG_;          // the input has more than one segment (>= 2^32 positions, or a test asked for it)
    typedef typename LcbWidth<SEG_>::T Flat;   // a flat position (segBase + g): 64-bit with several segments, else g itself
    typedef typename LcbWidth<SEG_>::T OccIdx; // an index into the occurrence records
    // (Keeping the fields of the first pool entries of the big variant in LDS, -DLCB_BIG_HOT of round 2, was measured in round 3:
    // -1 % - the HBM-resident fields are not what makes the big variant slow - and removed.)
    typedef uint32_t* FieldU;
    typedef int32_t* FieldI;
    LcbTables T;
    LcbUsed U;                 // the `used` state this seed reads
    LcbKParams P;
    uint32_t lane;
    const uint64_t* segBase;   // SEG: LDS copy of T.segBase
    uint32_t* segFirst;        // SEG: [nSeg + 1] first entry of every segment in the ordered index (LDS)
    // instance pool (SoA) — pool index order == allInstance_ order (path.h:684)
    FieldU iFrontG, iBackG, iFrontPos, iBackPos, iLo, iHi, iFlags;
    FieldI iFrontDist, iBackDist;
    uint32_t* ordKey;          // instance_ ordered sets flattened: keys (flat compare position) ... [2][instCap], half `cur` is live
    Idx* ordIdx;               // ... and pool indices, double buffered the same way
    Idx* good;                 // goodInstance_ (path.h:685), pool indices in append order
    Idx* goodPos;              // inverse of `good`: position of an instance in the good list, or NONE
    // The instances the last successful push extended or created (after Init: the initial instances). Exactly these end at
    // the path end (their end distance equals the flank, and path distances are strictly monotone), so they are the voters
    // of the next vote (blocksfinder.h:716-717) — no scan over the instance list is needed.
    Idx* touch;
    uint32_t nTouch, nInit;
    uint32_t endInst;          // instances in the pool when the seed ended (reported with an overflow status: the host picks the next variant by it)
    // Footprint: per instance ever created, the range of flat positions whose `used` bit was read as 0 and could
    // have changed the result (its span plus every look-ahead window walked from its ends). A result computed
    // against an older `used` snapshot is still exact iff no bit inside these ranges has been set since
    // (bits only go 0 -> 1 and the computation is deterministic). Survives the replay's Clear.
    FieldU fpLo, fpHi;
    uint8_t* fpSeg8;           // SEG: segment of every footprint slot (LDS-resident pools: bytes; else words behind fpHi)
    uint32_t* fpSeg32;
    uint32_t nFp;
    // The replay (blocksfinder.h:271-284) re-creates pool entries 0 .. n0-1 exactly as the forward extension had created them
    // (same occurrences, same order), so they keep their footprint slots. What the backward extension creates afterwards re-uses
    // the pool indices of the entries the forward extension had created beyond the best point - different instances: they get
    // slots of their own, pool index + fpShift for indices >= fpSplit (lcb_fp_slot). Sharing the old slot would still be a
    // superset (one hull over both instances), but such hulls span unrelated regions and void the result at almost any commit.
    uint32_t fpSplit, fpShift;
    uint32_t instCap;
    // vote table (open addressing): key, accumulated weight, (list ordinal << 16) | step of the last contribution
    int32_t* vKey;
    uint32_t* vCount;
    VLast* vLast;
    Idx* vTouched;             // slots claimed in the current vote, in claim order
    uint32_t* vNClaimed;       // LDS counter: number of them
    uint32_t* vTicket;         // LDS counter: next entry of the touch list to hand to a wavefront (votes shared by more than two wavefronts)
    uint32_t* vOvf;            // LDS flag: a walk of the current vote could not place a vertex
    uint32_t voteCap, voteShift;
    uint32_t* scr;             // LDS scratch, 4 * 64 words
    uint32_t* bloom;           // LDS Bloom filter over the path vertex set
    uint32_t bloomShift;
    uint32_t* mail;            // LDS mailbox to the helper wavefronts (LCB_MAIL_*)
    uint32_t* part;            // LDS: per-wave partial arg-max results, 8 words per wave
    unsigned long long* mailWalk;   // stats: walk steps counted by the helpers
    // path vertex set + bodies + result snapshot (global workspace)
    int32_t* pKeys;
    uint32_t* pSlots;
    uint32_t pathCap, pathShift;
    unsigned long long* body;  // right body: (strand << 40) | (segment << 32) | g of the iterator whose outgoing edge was pushed
    uint32_t bodyCap;
    uint4* best;
    uint32_t bestCap;
    // checkpoint of the forward extension at (or shortly before) the best-scoring point, so that the "replay" after the
    // forward extension (blocksfinder.h:271-284) restarts there instead of at Init: 6 arrays of bestCap words in the slot
    uint32_t* ck;
    uint32_t ckN, ckInst, ckGood, ckPath;   // pushes / instances / good instances / path vertices at the checkpoint (ckN 0 = none)
    int32_t ckFlank;
    // wave-uniform scalars
    uint32_t nInst, nGood, cur, nPath, nRight, nLeft, nBest, status;
    int32_t rightFlank, leftFlank;   // rightBodyFlank_, leftBodyFlank_ (path.h:692-693)
    uint32_t* dbg;             // flight recorder of this workgroup (may be null)
    const uint32_t* abort;     // stop flag of an asynchronous batch (null: none)
    uint32_t pfPush, pfVote, pfMaxProbe, pfMaxInst;   // cheap per-seed profile of the instrumented variant (wave-uniform)
    uint64_t pfTVote, pfTPush, pfTScore;              // 10 ns ticks spent in the vote / push / score+snapshot sections
    uint64_t pfTWalk, pfTWaitB, pfTReduce, pfTScan;   // vote sections of wave 0 (10 ns ticks): own walks, wait for the other waves' walks, arg-max + clearing, voter scans
    uint64_t pfVoters, pfChunks, pfTouchSum;          // voters / 64-step chunks walked by wave 0, sum of the touch-list lengths over the votes
    // per-lane event counters (stats mode)
    uint64_t cWalk, cOcc, cCompatCall, cCompatStep, cVote, cPush;
};

// footprint slot of pool entry i (see LcbStateT::fpSplit)
// (-DLCB_TEST_FP_SLOT_DIV=n, emulator tests only: a pool with 1/n of its footprint slots, so that small inputs reach the shared slots)
#ifndef LCB_TEST_FP_SLOT_DIV
#define LCB_TEST_FP_SLOT_DIV 1u
#endif
template <class ST>
__device__ __forceinline__ uint32_t lcb_fp_slot(const ST& S, uint32_t i)
{
    if (i < S.fpSplit) return i;
    const uint32_t f = i + S.fpShift;
    // no room for a slot of its own: it shares the old one - one hull over both instances, a superset: still exact, provided both lie in
    // the slot's segment (a slot is 32-bit offsets plus ONE segment; lcb_push ends the seed with LCB_ST_INST_OVF otherwise and the next
    // variant, whose pool is larger, gives the instance a slot of its own)
    return f < S.instCap / LCB_TEST_FP_SLOT_DIV ? f : i;
}

// segment of footprint slot fs (SEG kernels)
template <class ST>
__device__ __forceinline__ void lcb_fp_set_seg(ST& S, uint32_t fs, uint32_t seg)
{
    if (ST::SEG) { if (LcbCfg<ST::MODE>::INST_LDS) S.fpSeg8[fs] = (uint8_t)seg; else S.fpSeg32[fs] = seg; }
}
template <class ST>
__device__ __forceinline__ uint32_t lcb_fp_get_seg(const ST& S, uint32_t fs)
{
    return ST::SEG ? (LcbCfg<ST::MODE>::INST_LDS ? (uint32_t)S.fpSeg8[fs] : S.fpSeg32[fs]) : 0u;
}

// All fields of instance i, one per lane (lane f < 9 holds field f), with ONE load: the field arrays are equally spaced (lcb_process_body).
// A wave-uniform consumer that read them one by one - `lcb_rfl(S.iLo[i])`, ... - got six loads that the compiler serialises (load, wait,
// readfirstlane, next load: the scalarised value is needed at once and the variants at their register budget have no second
// register to load into): six round trips to the workspace in HBM per voter drawn in the big variant, four per extension step.
enum { LCB_F_FRONTG = 0, LCB_F_BACKG, LCB_F_FRONTPOS, LCB_F_BACKPOS, LCB_F_LO, LCB_F_HI, LCB_F_FLAGS, LCB_F_FRONTDIST, LCB_F_BACKDIST, LCB_F_COUNT };
template <class ST>
__device__ __forceinline__ uint32_t lcb_inst_fields(const ST& S, uint32_t i)
{
    const uint32_t stride = (uint32_t)(S.iBackG - S.iFrontG);
    uint32_t f = 0;
    if (S.lane < (uint32_t)LCB_F_COUNT) f = S.iFrontG[S.lane * stride + i];
    return f;
}

__device__ __forceinline__ uint32_t lcb_hash(int32_t vid, uint32_t shift)
{
    return ((uint32_t)vid * 2654435761u) >> shift;
}

__device__ __forceinline__ LcbUsed lcb_used_of(const LcbTables& T, uint32_t view)
{
    LcbUsed U;
    U.live = T.used;
    U.tab = view ? T.viewTab + (size_t)view * T.nPages : nullptr;
    return U;
}

// word w of the state (a view costs one more load, of a table entry that stays in the L1/L2 of the CU)
__device__ __forceinline__ uint32_t lcb_uword(const LcbUsed& U, uint32_t w)
{
    if (U.tab) w += LCB_GLOBAL_U32(U.tab)[w >> LCB_PAGE_SHIFT];
    return U.live[w];
}

// Flat positions (segBase + g) are 64-bit; the bitmap has fewer than 2^32 words (2^37 positions: 4 TB of tables), so a word index
// is 32-bit. With one segment the base is the constant 0 and everything below folds back to 32-bit arithmetic on g.
template <class ST>
__device__ __forceinline__ typename ST::Flat lcb_seg_base(const ST& S, uint32_t seg) { return ST::SEG ? (typename ST::Flat)S.segBase[seg] : (typename ST::Flat)0; }
// segment of a chromosome word / of an instance's flag word
__device__ __forceinline__ uint32_t lcb_cw_seg(uint32_t cw) { return cw >> LCB_SEG_SHIFT; }
__device__ __forceinline__ uint32_t lcb_fl_seg(uint32_t fl) { return fl >> LCB_FLAG_SEG_SHIFT; }

// occurrences [o0, o1) of vertex |id| = av
template <bool SEG>
__device__ __forceinline__ typename LcbWidth<SEG>::T lcb_occ_start(const LcbTables& T, uint32_t av) { return SEG ? (typename LcbWidth<SEG>::T)T.occStart64[av] : (typename LcbWidth<SEG>::T)T.occStart32[av]; }

// (F: uint32_t or uint64_t)
template <class F>
__device__ __forceinline__ bool lcb_used_bit(const LcbUsed& U, F f)
{
    return (lcb_uword(U, (uint32_t)(f >> 5)) >> ((uint32_t)f & 31u)) & 1u;
}

// JunctionSequentialIterator::IsUsed (junctionstorage.h:270-283): `used` marks the edge idx -> idx+1. (sb: base of the segment of g)
template <class F>
__device__ __forceinline__ bool lcb_it_used(const LcbUsed& U, F sb, uint32_t g, bool positive, uint32_t lo)
{
    if (positive) return lcb_used_bit(U, (F)(sb + g));
    return g > lo ? lcb_used_bit(U, (F)(sb + g - 1)) : false;
}

// JunctionSequentialIterator::GetChar (junctionstorage.h:234-243), f = flat index
template <class F>
__device__ __forceinline__ uint8_t lcb_it_char(const LcbTables& T, F f, bool positive)
{
    return positive ? T.posCh[f] : T.posRevCh[f];
}

// Any used bit in [a, b)?  This is the `used` walk of Path::Compatible (path.h:387-393) for both strands:
// + strand visits bits a..b-1 going up, - strand visits bits b-1..a going down. (flat indices)
template <class F>
__device__ inline bool lcb_range_any_used(const LcbUsed& U, F a, F b)
{
    if (a >= b) return false;
    const uint32_t wa = (uint32_t)(a >> 5), wb = (uint32_t)((b - 1) >> 5);
    const uint32_t ma = 0xFFFFFFFFu << ((uint32_t)a & 31), mb = 0xFFFFFFFFu >> (31 - ((uint32_t)(b - 1) & 31));
    if (wa == wb) return (lcb_uword(U, wa) & ma & mb) != 0;
    if (lcb_uword(U, wa) & ma) return true;
    for (uint32_t w = wa + 1; w < wb; w++)
        if (lcb_uword(U, w)) return true;
    return (lcb_uword(U, wb) & mb) != 0;
}

// Number of iterations the reference's walk makes over [a, b) (stats mode only).
template <class F>
__device__ inline uint32_t lcb_range_walk_steps(const LcbUsed& used, F a, F b, bool up)
{
    uint32_t steps = 0;
    if (up) { for (F g = a; g < b; g++) { steps++; if (lcb_used_bit(used, g)) break; } }
    else { for (F g = b; g > a; g--) { steps++; if (lcb_used_bit(used, (F)(g - 1))) break; } }
    return steps;
}

__device__ __forceinline__ uint32_t lcb_bcast(uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)src); }

// Wave-wide sum of a 64-bit value (same DPP ladder as lcb_wave_umax, both halves moved together); wave-uniform result.
#define LCB_DPP_ADD64(v, ctrl, rowMask)                                                                             \
    do {                                                                                                            \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v), ctrl, rowMask, 0xf, false); \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((v) >> 32), ctrl, rowMask, 0xf, false); \
        (v) += ((uint64_t)hi_ << 32) | lo_;                                                                         \
    } while (0)
__device__ __forceinline__ int64_t lcb_wave_sum(int64_t x)
{
    uint64_t v = (uint64_t)x;
    LCB_DPP_ADD64(v, 0x111, 0xf); LCB_DPP_ADD64(v, 0x112, 0xf); LCB_DPP_ADD64(v, 0x114, 0xf); LCB_DPP_ADD64(v, 0x118, 0xf);
    LCB_DPP_ADD64(v, 0x142, 0xa); LCB_DPP_ADD64(v, 0x143, 0xc);
    return (int64_t)(((uint64_t)lcb_rl((uint32_t)(v >> 32), 63) << 32) | lcb_rl((uint32_t)v, 63));
}

// ---- path vertex set (DistanceKeeper::IsSet / Set / Unset, distancekeeper.h:17-35) --------------
// Exact open-addressing set in the workgroup's global workspace, fronted by an LDS Bloom filter: the common
// answer during a look-ahead walk is "not in the path", which the filter gives from LDS without touching
// global memory; only filter hits probe the exact table.
__device__ __forceinline__ uint32_t lcb_bloom1(int32_t vid, uint32_t sh) { return ((uint32_t)vid * 2654435761u) >> sh; }
__device__ __forceinline__ uint32_t lcb_bloom2(int32_t vid, uint32_t sh) { return ((uint32_t)vid * 0xC2B2AE35u + 0x27D4EB2Fu) >> sh; }

template <class ST>
__device__ __forceinline__ bool lcb_bloom_maybe(const ST& S, int32_t vid)
{
    const uint32_t a = lcb_bloom1(vid, S.bloomShift), b = lcb_bloom2(vid, S.bloomShift);
    return ((S.bloom[a >> 5] >> (a & 31)) & (S.bloom[b >> 5] >> (b & 31)) & 1u) != 0;
}

template <class ST>
__device__ inline bool lcb_path_probe(const ST& S, int32_t vid, uint32_t& probes)
{
    uint32_t h = lcb_hash(vid, S.pathShift);
    const uint32_t mask = S.pathCap - 1;
    uint32_t probe = 0;
    int32_t k = S.pKeys[h];
    while (k != vid && k != LCB_EMPTY_KEY && probe < S.pathCap) {   // the set is at most half full; the bound only guards a corrupted table
        h = (h + 1) & mask; probe++;
        k = S.pKeys[h];
    }
    probes = probe;
    return k == vid;
}

template <class ST>
__device__ __forceinline__ bool lcb_path_contains(const ST& S, int32_t vid)
{
    if (LcbCfg<ST::MODE>::BW && !lcb_bloom_maybe(S, vid)) return false;    // (no filter in front of an LDS-resident set)
    uint32_t probes;
    return lcb_path_probe(S, vid, probes);
}

// Wave-uniform: inserts vid (not present). Lane 0 writes.
template <class ST>
__device__ inline void lcb_path_insert(ST& S, int32_t vid)
{
    if ((S.nPath + 1) * 2 > S.pathCap) { S.status = LCB_ST_PATH_OVF; return; }
    uint32_t h = lcb_hash(vid, S.pathShift);
    const uint32_t mask = S.pathCap - 1;
    uint32_t probe = 0;
    while (S.pKeys[h] != LCB_EMPTY_KEY && probe < S.pathCap) { h = (h + 1) & mask; probe++; }
    if (probe == S.pathCap) { S.status = LCB_ST_PATH_OVF; return; }
    constexpr bool PL = LcbCfg<ST::MODE>::PC != 0;
    LCB_SYNC_IF(PL);               // every lane has finished probing before lane 0 publishes the key
    if (S.lane == 0) {
        S.pKeys[h] = vid; S.pSlots[S.nPath] = h;
        if (LcbCfg<ST::MODE>::BW) {
            const uint32_t a = lcb_bloom1(vid, S.bloomShift), b = lcb_bloom2(vid, S.bloomShift);
            S.bloom[a >> 5] |= 1u << (a & 31);
            S.bloom[b >> 5] |= 1u << (b & 31);
        }
    }
    S.nPath++;
    LCB_SYNC_IF(PL);
}

// Path::Clear (path.h:650-677): wave-uniform. The right-body list is kept (the replay reads it).
template <class ST>
__device__ inline void lcb_path_clear(ST& S)
{
    constexpr uint32_t BW = LcbCfg<ST::MODE>::BW;
    for (uint32_t i = S.lane; i < S.nPath; i += 64) S.pKeys[S.pSlots[i]] = LCB_EMPTY_KEY;
    if (BW && S.nPath) for (uint32_t i = S.lane; i < (uint32_t)BW; i += 64) S.bloom[i] = 0;
    S.nPath = 0; S.nRight = 0; S.nLeft = 0; S.nInst = 0; S.nGood = 0; S.cur = 0; S.nTouch = 0;
    S.rightFlank = 0; S.leftFlank = 0;
    if (ST::SEG && S.lane <= LCB_MAX_SEG) S.segFirst[S.lane] = 0;
    LCB_WAVE_SYNC();
}

// ---- instance helpers ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lcb_absdiff(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

// Instance::RealLength (path.h:165-168): |front.GetPosition() - back.GetPosition()| (k cancels).
template <class ST>
__device__ __forceinline__ int64_t lcb_real_length(const ST& S, uint32_t i)
{
    return (int64_t)lcb_absdiff(S.iFrontPos[i], S.iBackPos[i]);
}

// The ordered index is sorted by (segment, g): the entries of segment s are [segFirst[s], segFirst[s + 1]) and their keys are plain
// 32-bit g (an occurrence is only ever compared with instances of its own chromosome, hence of its own segment). One segment:
// the whole index, no cursors.
// Rebuilds the ordered index after m inserts of one chunk. The insert list (position in the OLD order,
// key, pool index; SEG: segment) is in scr[0..m), scr[64..64+m), scr[128..128+m), scr[192..192+m), ascending by position then key.
template <class ST>
__device__ inline void lcb_order_merge(ST& S, uint32_t m)
{
    const uint32_t n = S.nInst - m;     // old element count (nInst already includes the new ones)
    const uint32_t c = S.cur, d = c ^ 1;
    const uint32_t* ck = S.ordKey + c * S.instCap; const typename ST::Idx* ci = S.ordIdx + c * S.instCap;
    uint32_t* dk = S.ordKey + d * S.instCap; typename ST::Idx* di = S.ordIdx + d * S.instCap;
    for (uint32_t i = S.lane; i < n; i += 64) {
        uint32_t cnt = 0;
        for (uint32_t r = 0; r < m; r++) cnt += (S.scr[r] <= i) ? 1u : 0u;
        dk[i + cnt] = ck[i];
        di[i + cnt] = ci[i];
    }
    if (S.lane < m) {
        const uint32_t at = S.scr[S.lane] + S.lane;
        dk[at] = S.scr[64 + S.lane];
        di[at] = (typename ST::Idx)S.scr[128 + S.lane];
    }
    if (ST::SEG && S.lane <= LCB_MAX_SEG) {          // cursor t moves up by the inserts into the segments below t
        uint32_t cnt = 0;
        for (uint32_t r = 0; r < m; r++) cnt += (S.scr[192 + r] < S.lane) ? 1u : 0u;
        S.segFirst[S.lane] += cnt;
    }
    S.cur = d;
    LCB_SYNC_IF(LcbCfg<ST::MODE>::IDX_LDS);
}

// Path::Init (path.h:33-46): one instance per unused occurrence of vid whose next character is ch.
template <bool STATS, class ST>
__device__ inline void lcb_path_init(ST& S, int32_t vid, int32_t ch)
{
    const LcbTables& T = S.T;
    lcb_path_insert(S, vid);          // distanceKeeper_.Set(vid, 0)
    if (S.status) return;
    const uint32_t av = (uint32_t)(vid < 0 ? -vid : vid);
    typedef typename ST::Flat Flat;
    typedef typename ST::OccIdx OccIdx;
    const OccIdx o0 = lcb_rfl(lcb_occ_start<ST::SEG>(T, av)), o1 = lcb_rfl(lcb_occ_start<ST::SEG>(T, av + 1));
    for (OccIdx base = o0; base < o1; base += 64) {
        const OccIdx j = base + S.lane;
        bool ok = false;
        uint32_t g = 0, chr = 0, lo = 0, hi = 0, pos = 0;       // chr: the chromosome word (segment | chromosome)
        bool positive = false;
        if (j < o1) {
            if (STATS) S.cOcc++;
            const uint4 rec = T.occRec[j];
            g = rec.x; chr = rec.y; pos = rec.z;
            const uint2 lh = T.chrLoHi[chr & LCB_CHR_MASK];
            lo = lh.x; hi = lh.y;
            positive = ((int32_t)rec.w == vid);
            const Flat sb = lcb_seg_base(S, lcb_cw_seg(chr));
            ok = !lcb_it_used(S.U, sb, g, positive, lo) && (int32_t)lcb_it_char(T, (Flat)(sb + g), positive) == ch;
        }
        const unsigned long long m = __ballot(ok);
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (S.nInst + cnt > S.instCap) { S.status = LCB_ST_INST_OVF; return; }
        if (ok) {
            const uint32_t i = S.nInst + (uint32_t)__popcll(m & ((1ull << S.lane) - 1));
            S.iFrontG[i] = g; S.iBackG[i] = g; S.iFrontPos[i] = pos; S.iBackPos[i] = pos;
            S.iFrontDist[i] = 0; S.iBackDist[i] = 0; S.iLo[i] = lo; S.iHi[i] = hi;
            S.iFlags[i] = (chr << LCB_FLAG_BITS) | (positive ? LCB_FLAG_POS : 0u);
            if (i >= S.nFp) { S.fpLo[i] = g; S.fpHi[i] = g; lcb_fp_set_seg(S, i, lcb_cw_seg(chr)); }   // the replay re-creates instance i at the same occurrence
            S.ordKey[S.cur * S.instCap + i] = g;   // occurrences ascend in (segment, g), so pool order == key order here
            S.ordIdx[S.cur * S.instCap + i] = (typename ST::Idx)i;
            S.goodPos[i] = ST::NONE;
            S.touch[i] = (typename ST::Idx)i;              // every initial instance ends at the seed vertex: they vote first
        }
        S.nInst += cnt;
        if (S.nInst > S.nFp) S.nFp = S.nInst;
    }
    S.nTouch = S.nInst; S.nInit = S.nInst;
    LCB_WAVE_SYNC();
    if (ST::SEG) {
        // cursor t = number of instances in the segments below t (the pool is sorted by segment here)
        if (S.lane <= LCB_MAX_SEG) {
            uint32_t a = 0, b = S.nInst;
            while (a < b) { const uint32_t mid = (a + b) >> 1; if (lcb_fl_seg(S.iFlags[mid]) < S.lane) a = mid + 1; else b = mid; }
            S.segFirst[S.lane] = a;
        }
        LCB_WAVE_SYNC();
    }
}

// ---- the vote: MostPopularVertex (blocksfinder.h:708-768) --------------------------------------
// One voter = one instance whose end vertex is the path end (blocksfinder.h:716-717). Its scalars live in SGPRs.

// The voter walks of one vote. The voters are the instances of the touch list that are in the voting list (the good list
// if it has two entries, else all instances; blocksfinder.h:713). With two wavefronts wave w takes the voters with ordinal == w
// (mod 2); with more (wide, big, huge: votes of dozens of voters whose windows take one to six 64-step chunks) the wavefronts draw
// the entries of the touch list one by one from an LDS ticket counter - a wavefront that is done with a short window takes the
// next voter instead of waiting at the barrier for the one that drew three long ones (measured with the static deal: the walking
// wavefront waits 2.0 / 3.6 us of a 8.3 / 13.3 us vote in the wide / big variant). All waves accumulate into the shared vote table
// with atomics, so the split needs no merging - and no order: the arg-max is order-free.
// `exact`: the walks stop at vertices of the path (blocksfinder.h:736-741). Where the path set lives in the HBM slot behind an LDS
// Bloom filter (compact, big, huge) a per-step membership test costs a dependent global round trip for every chunk in which the
// filter says "maybe" for one lane - with a path of thousands of vertices that is almost every chunk. A walk meets a path vertex
// only at repeats and rearrangements, so the first pass of a vote walks WITHOUT the test (exact = false: only `used` positions and
// the window bound stop a walk) and the distinct vertices it touched - they are all in the vote table - are checked afterwards,
// lane-parallel, once (lcb_vote_any_in_path). No hit: no walk can have met a path vertex, the pass is what the reference computes.
// A hit: the table is cleared and the vote repeated with exact = true. (Footprints of a discarded pass stay: a superset.) Variants
// with the path set in LDS always walk exactly.
// (Tried and measured slower on the MI355X, profiles/r03: requesting the next chunk of a voter ahead of time with unconditional,
// straight-line loads - the main wavefront is bound by instruction issue, not by the round trips the prefetch hides.)
// ---- The round-5 walk - one voter after the other, chunk by chunk, voters drawn from LDS tickets. Which kernel variant walks which way is a
// bit mask over the variants (bit m set: variant m takes the window-table walk below); decided by same-box A/B on the MI355X (profiles/r06).
#ifndef LCB_WALK_V2_MODES
#define LCB_WALK_V2_MODES 0x11u
#endif
#define LCB_WALK_V2_OF(ST) (((LCB_WALK_V2_MODES) >> ST::MODE) & 1u)
template <class F> struct LcbVoterT { uint32_t e, i, g0, pos0, lo, rem, weight; int32_t dir; bool positive; F sb; };   // sb: base of the voter's segment
struct LcbWalk { uint32_t g, pos; int32_t id; uint32_t uw; bool valid; };
template <bool STATS, bool PROF = false, class ST>
__device__ inline void lcb_vote_walk_v1(ST& S, bool forward, bool tryUsed, bool useGood, uint32_t waveId, uint32_t nWaves, bool exact)
{
    const LcbTables& T = S.T;
    typedef typename ST::Flat Flat;
    typedef LcbVoterT<Flat> LcbVoter;
    const uint32_t vmask = S.voteCap - 1;
    const uint32_t claimCap = S.voteCap - (S.voteCap >> 2);
    const uint32_t depth = (uint32_t)S.P.depth, maxBranch = (uint32_t)S.P.maxBranch;
    // the current chunk of 64 touch-list entries (per-lane fields) and the voters still to hand out from it
    const uint32_t nTouch = S.nTouch;
    uint32_t chunkBase = 0, ordinal = 0;
    unsigned long long pend = 0;
    uint32_t fE = 0, fI = 0, fG = 0, fPos = 0, fLo = 0, fHi = 0, fW = 0, fFl = 0;
    bool scanned = false;
    const bool tickets = nWaves > 2;
    auto nextVoter = [&](LcbVoter& v) -> bool {
        if (tickets) {
            // one entry of the touch list per draw; its fields are wave-uniform loads (the next voter is drawn while the current one
            // is walked, so their latency hides behind a walk)
            for (;;) {
                uint32_t t = 0;
                if (S.lane == 0) t = atomicAdd(S.vTicket, 1u);
                t = lcb_rfl(t);
                if (t >= nTouch) return false;
                const uint32_t i = lcb_rfl((uint32_t)S.touch[t]);
                const uint32_t e = useGood ? lcb_rfl((uint32_t)S.goodPos[i]) : i;      // position in the voting list (its order breaks ties)
                if (e == ST::NONE) continue;
                const uint32_t f = lcb_inst_fields(S, i);
                const uint32_t fl = lcb_rl(f, LCB_F_FLAGS), fp = lcb_rl(f, LCB_F_FRONTPOS), bp = lcb_rl(f, LCB_F_BACKPOS);
                v.e = e; v.i = i;
                v.g0 = forward ? lcb_rl(f, LCB_F_BACKG) : lcb_rl(f, LCB_F_FRONTG);
                v.pos0 = forward ? bp : fp;
                v.lo = lcb_rl(f, LCB_F_LO);
                const uint32_t hi = lcb_rl(f, LCB_F_HI);
                v.weight = lcb_absdiff(fp, bp) + 1u;                                   // blocksfinder.h:719
                v.positive = (fl & LCB_FLAG_POS) != 0;
                v.dir = (forward == v.positive) ? 1 : -1;
                v.rem = v.dir > 0 ? hi - 1u - v.g0 : v.g0 - v.lo;                      // steps for which it.Valid() holds
                v.sb = ST::SEG ? (Flat)lcb_rfl(S.segBase[lcb_fl_seg(fl)]) : (Flat)0;
                return true;
            }
        }
        for (;;) {
            while (pend == 0) {
                if (scanned) chunkBase += 64;
                scanned = true;
                if (chunkBase >= nTouch) return false;
                const uint32_t t = chunkBase + S.lane;
                bool is = false;
                if (t < nTouch) {
                    fI = S.touch[t];
                    fE = useGood ? (uint32_t)S.goodPos[fI] : fI;          // position in the voting list (its order breaks ties)
                    is = fE != ST::NONE;
                    if (is) {
                        fFl = S.iFlags[fI];
                        const uint32_t fp = S.iFrontPos[fI], bp = S.iBackPos[fI];
                        fG = forward ? S.iBackG[fI] : S.iFrontG[fI];
                        fPos = forward ? bp : fp;
                        fLo = S.iLo[fI]; fHi = S.iHi[fI];
                        fW = lcb_absdiff(fp, bp) + 1u;                                       // blocksfinder.h:719
                    }
                }
                pend = __ballot(is);
            }
            const uint32_t b = (uint32_t)__ffsll((long long)pend) - 1u;
            pend &= pend - 1;
            const uint32_t o = ordinal++;
            if (nWaves > 1 && (o % nWaves) != waveId) continue;
            v.e = lcb_rl(fE, b);
            v.i = lcb_rl(fI, b); v.g0 = lcb_rl(fG, b); v.pos0 = lcb_rl(fPos, b); v.lo = lcb_rl(fLo, b);
            const uint32_t hi = lcb_rl(fHi, b);
            v.weight = lcb_rl(fW, b);
            const uint32_t vfl = lcb_rl(fFl, b);
            v.positive = (vfl & LCB_FLAG_POS) != 0;
            v.dir = (forward == v.positive) ? 1 : -1;
            v.rem = v.dir > 0 ? hi - 1u - v.g0 : v.g0 - v.lo;                               // steps for which it.Valid() holds
            v.sb = ST::SEG ? (Flat)lcb_rfl(S.segBase[lcb_fl_seg(vfl)]) : (Flat)0;
            return true;
        }
    };
    // The table reads of a pass (pos, id, used word) are independent and issued together; the first pass of the NEXT
    // voter is issued before the current one is consumed, so its latency hides behind the LDS work of this one.
    auto issue = [&](const LcbVoter& v, uint32_t c) -> LcbWalk {
        LcbWalk w;
        const uint32_t d = c * 64 + S.lane + 1;
        w.valid = d <= v.rem;                                                               // it.Valid()
        w.g = v.dir > 0 ? v.g0 + d : v.g0 - d;
        w.pos = 0; w.id = 0; w.uw = 0;
        if (w.valid) {
            // (wave-uniform table base + 32-bit lane offset)
            w.pos = (T.posPos + v.sb)[w.g];
            w.id = (T.posId + v.sb)[w.g];
            // IsUsed: + strand bit g, - strand bit g-1 (none at the chromosome start)
            const Flat ub = v.sb + (w.g - (v.positive ? 0u : 1u));
            if (!tryUsed && (v.positive || w.g > v.lo)) w.uw = lcb_uword(S.U, (uint32_t)(ub >> 5)) >> ((uint32_t)ub & 31u);
        }
        return w;
    };
    LcbVoter cur, nxt;
    LcbWalk wcur, wnxt;
    bool have = nextVoter(cur);
    if (have) wcur = issue(cur, 0);
    while (have) {
        const bool haveNext = nextVoter(nxt);
        if (haveNext) wnxt = issue(nxt, 0);
        for (uint32_t c = 0;; c++) {
            const LcbWalk w = c == 0 ? wcur : issue(cur, c);
            if (PROF) S.pfChunks++;
            const uint32_t d = c * 64 + S.lane + 1;
            const bool cond = w.valid && (d < depth || lcb_absdiff(w.pos, cur.pos0) <= maxBranch);
            const int32_t vid = cur.positive ? w.id : -w.id;
            const bool stop = cond && ((w.uw & 1u) != 0 || (exact && lcb_path_contains(S, vid)));
            const unsigned long long failM = __ballot(!cond);
            const unsigned long long stopM = __ballot(stop);
            const unsigned long long endM = failM | stopM;
            const uint32_t first = endM ? (uint32_t)(__ffsll((long long)endM) - 1) : 64u;
            if (STATS) {
                // loop iterations entered: contributing steps plus the breaking one (not the failed loop test)
                const bool breaking = stopM && (uint32_t)(__ffsll((long long)stopM) - 1) == first;
                if (S.lane < first || (S.lane == first && breaking)) S.cWalk++;
            }
            if (S.lane < first) {
                uint32_t h = lcb_hash(vid, S.voteShift);
                int32_t old = atomicCAS(&S.vKey[h], LCB_EMPTY_KEY, vid);
                uint32_t probe = 0;
                while (old != LCB_EMPTY_KEY && old != vid && probe < S.voteCap) {
                    h = (h + 1) & vmask; probe++;
                    old = atomicCAS(&S.vKey[h], LCB_EMPTY_KEY, vid);
                }
                if (old == LCB_EMPTY_KEY) {
                    const uint32_t t = atomicAdd(S.vNClaimed, 1u);
                    if (t < claimCap) S.vTouched[t] = (typename ST::Idx)h; else *S.vOvf = 1u;
                }
                if (old == LCB_EMPTY_KEY || old == vid) {
                    atomicAdd(&S.vCount[h], cur.weight);
                    atomicMax(&S.vLast[h], ((typename ST::VLast)cur.e << ST::LAST_SHIFT) | d);
                } else *S.vOvf = 1u;
            }
            if (first < 64) {
                if (!tryUsed && S.lane == 0) {
                    // steps 1 .. c*64+first-1 read used == 0 (one step of slack keeps the - strand's bit g-1 inside)
                    const uint32_t st = c * 64 + first;
                    const uint32_t ge = st > cur.rem ? (cur.dir > 0 ? cur.g0 + cur.rem : cur.lo) : (cur.dir > 0 ? cur.g0 + st : cur.g0 - st);
                    const uint32_t fs = lcb_fp_slot(S, cur.i);
                    atomicMin(&S.fpLo[fs], ge);
                    atomicMax(&S.fpHi[fs], ge);
                }
                break;
            }
        }
        if (PROF) S.pfVoters++;
        have = haveNext; cur = nxt; wcur = wnxt;
    }
}


// Round 6: the walk knows a voter's window before it reads anything of it. LcbTables::posWin holds, per position, the number of steps in
// either direction that stay within maxBranch bp (a prefix, because positions ascend strictly inside a chromosome), so the reference's
// loop test `step < lookingDepth || |pos - pos0| <= maxBranch` (blocksfinder.h:722-727) is `step <= L` with
// L = min(steps to the chromosome end, max(depth - 1, win)), and the only other thing that ends a walk early - a used position
// (blocksfinder.h:729-735) - is found lane-parallel over the VOTERS from the few bitmap words their windows cover. After that stage a
// voter is (origin, direction, number of contributing steps), and the rest of the vote is a flat gather of vertex ids + hash inserts:
//   stage A  lanes = this wavefront's voters (entry t of the touch list goes to wavefront t mod nWaves): instance fields, window word,
//            8 bitmap words in walking order - all voters' loads in flight together, ONE round trip - then the contributing steps nv,
//            the footprint update and the event counter, per lane;
//   stage B  items (voter, 64-step chunk), LCB_VB of them per round: their id loads are issued back to back (the first LCB_VB voters'
//            first chunks speculatively, together with the loads of stage A), then consumed one after the other.
// Until round 5 a wavefront walked one voter after the other, chunk by chunk, each chunk waiting for its own loads (the compiler's
// s_waitcnt vmcnt(0) at the head of the chunk loop also swallowed the "prefetch" of the next voter): ~1 us per chunk in every
// variant (profiles/r04/ab_third.txt), i.e. the latency of one global round trip per 64 steps. The path-membership stop
// (blocksfinder.h:736-741; `exact`) is evaluated per item and shortens the voter's nv for its later items.
#ifndef LCB_VB
#define LCB_VB 4
#endif
__host__ __device__ inline uint32_t lcb_brev32(uint32_t x)
{
#if defined(__clang__)
    return __builtin_bitreverse32(x);       // v_bfrev_b32
#else                                       // (g++: the CPU emulator of the tests)
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) if (x & (1u << i)) r |= 1u << (31 - i);
    return r;
#endif
}
#define LCB_BREV32(x) lcb_brev32(x)

template <bool STATS, bool PROF = false, class ST>
__device__ inline void lcb_vote_walk(ST& S, bool forward, bool tryUsed, bool useGood, uint32_t waveId, uint32_t nWaves, bool exact)
{
    if (!LCB_WALK_V2_OF(ST)) { lcb_vote_walk_v1<STATS, PROF>(S, forward, tryUsed, useGood, waveId, nWaves, exact); return; }
    const LcbTables& T = S.T;
    typedef typename ST::Flat Flat;
    const uint32_t vmask = S.voteCap - 1;
    const uint32_t claimCap = S.voteCap - (S.voteCap >> 2);
    const uint32_t depth = (uint32_t)S.P.depth;
    const uint32_t dm = depth ? depth - 1u : 0u;                  // steps allowed by `step < lookingDepth` alone
    const uint32_t nTouch = S.nTouch;
    const uint32_t maxW = (uint32_t)(T.nPos >> 5);               // (the bitmap has at least nPos / 32 + 2 words)
    for (uint32_t p0 = 0; p0 * nWaves + waveId < nTouch; p0 += 64) {
        // (Nothing is in flight here, but the compiler cannot know - its wait-count analysis is conservative around the loops below and
        // would otherwise wait for "possibly pending" loads between the requests of stage A, one round trip each. Said once, it is free.)
        LCB_NO_LOADS_IN_FLIGHT();
        // ---- stage A: one voter per lane
        const uint32_t t = (p0 + S.lane) * nWaves + waveId;
        bool isV = false;
        uint32_t vI = 0, vE = 0, g0 = 0, rem = 0, weight = 0, fl = 0;
        if (t < nTouch) {
            vI = S.touch[t];
            vE = useGood ? (uint32_t)S.goodPos[vI] : vI;          // position in the voting list (its order breaks ties)
            isV = vE != ST::NONE;
        }
        if (isV) {
            fl = S.iFlags[vI];
            const uint32_t fp = S.iFrontPos[vI], bp = S.iBackPos[vI];
            g0 = forward ? S.iBackG[vI] : S.iFrontG[vI];
            const uint32_t lo = S.iLo[vI], hi = S.iHi[vI];
            weight = lcb_absdiff(fp, bp) + 1u;                    // blocksfinder.h:719
            rem = (forward == ((fl & LCB_FLAG_POS) != 0)) ? hi - 1u - g0 : g0 - lo;   // steps for which it.Valid() holds
        }
        const bool positive = (fl & LCB_FLAG_POS) != 0, up = forward == positive;
        Flat sb = 0;
        if (ST::SEG) sb = (Flat)S.segBase[lcb_fl_seg(fl)];
        const Flat f0 = sb + g0;
        const bool walks = isV && rem != 0;
        uint32_t win = 0;
        if (walks) win = T.posWin[f0];
        // IsUsed of step d: + strand bit g, - strand bit g - 1 (none at the chromosome start). The bits of steps 1, 2, ... in walking
        // order: flat bit fb, then fb + 1, ... (up) or fb - 1, ... (down); 8 words of it cover at least 225 steps.
        const uint32_t delta = positive ? 0u : 1u;
        const bool scan = walks && !tryUsed && !(!up && f0 < (Flat)(1u + delta));   // (down from flat position 0 / 1 on the - strand: no step has a bit)
        const Flat fb = scan ? (up ? f0 + 1u - delta : f0 - 1u - delta) : (Flat)0;
        uint32_t uw0 = 0, uw1 = 0, uw2 = 0, uw3 = 0, uw4 = 0, uw5 = 0, uw6 = 0, uw7 = 0;
        if (scan) {
            const uint32_t W0 = (uint32_t)(fb >> 5);
            const uint32_t um = up ? 0xFFFFFFFFu : 0u;       // (selects by mask, not by branch: the eight addresses are straight-line code)
#define LCB_WIDX(k) ((((W0 + (k)) < maxW ? (W0 + (k)) : maxW) & um) | ((W0 - (W0 < (k) ? W0 : (k))) & ~um))
            uint32_t a0 = LCB_WIDX(0u), a1 = LCB_WIDX(1u), a2 = LCB_WIDX(2u), a3 = LCB_WIDX(3u), a4 = LCB_WIDX(4u), a5 = LCB_WIDX(5u), a6 = LCB_WIDX(6u), a7 = LCB_WIDX(7u);
#undef LCB_WIDX
            if (S.U.tab) {
                const auto tab = LCB_GLOBAL_U32(S.U.tab);
                const uint32_t o0 = tab[a0 >> LCB_PAGE_SHIFT], o1 = tab[a1 >> LCB_PAGE_SHIFT], o2 = tab[a2 >> LCB_PAGE_SHIFT], o3 = tab[a3 >> LCB_PAGE_SHIFT];
                const uint32_t o4 = tab[a4 >> LCB_PAGE_SHIFT], o5 = tab[a5 >> LCB_PAGE_SHIFT], o6 = tab[a6 >> LCB_PAGE_SHIFT], o7 = tab[a7 >> LCB_PAGE_SHIFT];
                a0 += o0; a1 += o1; a2 += o2; a3 += o3; a4 += o4; a5 += o5; a6 += o6; a7 += o7;
            }
            uw0 = S.U.live[a0]; uw1 = S.U.live[a1]; uw2 = S.U.live[a2]; uw3 = S.U.live[a3];
            uw4 = S.U.live[a4]; uw5 = S.U.live[a5]; uw6 = S.U.live[a6]; uw7 = S.U.live[a7];
        }
        // ---- the first chunks of the first LCB_VB voters are requested now, before their windows are known
        unsigned long long pend = __ballot(walks);
        const unsigned long long walkM = pend;
        if (PROF) S.pfVoters += (uint32_t)__popcll(walkM);
        uint32_t itB[LCB_VB], itC[LCB_VB];
        int32_t itId[LCB_VB];
        bool itV[LCB_VB];
        // id of step c * 64 + lane + 1 of voter b (any position of the chromosome stands in for steps beyond its end)
        auto request = [&](uint32_t b, uint32_t c) -> int32_t {
            const uint32_t bg = lcb_rl(g0, b), br = lcb_rl(rem, b), bf = lcb_rl(fl, b);
            const bool bup = forward == ((bf & LCB_FLAG_POS) != 0);
            const uint32_t d = c * 64 + S.lane + 1;
            const uint32_t g = d <= br ? (bup ? bg + d : bg - d) : bg;
            if (ST::SEG) { const Flat bsb = (Flat)lcb_rfl(S.segBase[lcb_fl_seg(bf)]); return (T.posId + bsb)[g]; }
            return T.posId[g];
        };
#pragma unroll
        for (int q = 0; q < LCB_VB; q++) {
            itV[q] = pend != 0; itB[q] = 0; itC[q] = 0; itId[q] = 0;
            if (itV[q]) { itB[q] = (uint32_t)__ffsll((long long)pend) - 1u; pend &= pend - 1; itId[q] = request(itB[q], 0); }
        }
        // ---- stage A, second half: contributing steps of every voter
        uint32_t nv = 0, brk = 0;                                  // steps that vote; 1 if a breaking step (used / in the path) follows them
        if (walks) {
            const uint32_t wv = up ? (win & 0xFFFFu) : (win >> 16);
            uint32_t L = wv > dm ? wv : dm;
            if (L > rem) L = rem;
            nv = L;
            if (scan) {
                const uint32_t Lu = (!positive && !up && L == rem) ? L - 1u : L;   // the step onto the chromosome's first position reads no bit on the - strand
                const uint32_t sh = up ? ((uint32_t)fb & 31u) : 31u - ((uint32_t)fb & 31u);
                if (!up) { uw0 = LCB_BREV32(uw0); uw1 = LCB_BREV32(uw1); uw2 = LCB_BREV32(uw2); uw3 = LCB_BREV32(uw3); uw4 = LCB_BREV32(uw4); uw5 = LCB_BREV32(uw5); uw6 = LCB_BREV32(uw6); uw7 = LCB_BREV32(uw7); }
                const uint64_t s0 = (uint64_t)uw0 | ((uint64_t)uw1 << 32), s1 = (uint64_t)uw2 | ((uint64_t)uw3 << 32), s2 = (uint64_t)uw4 | ((uint64_t)uw5 << 32), s3 = (uint64_t)uw6 | ((uint64_t)uw7 << 32);
                const uint64_t x0 = (s0 >> sh) | ((s1 << 1) << (63u - sh)), x1 = (s1 >> sh) | ((s2 << 1) << (63u - sh)), x2 = (s2 >> sh) | ((s3 << 1) << (63u - sh)), x3 = s3 >> sh;
                const uint32_t have = 256u - sh;                   // steps whose bit is in x0 .. x3
                const uint32_t n = Lu < have ? Lu : have;
                uint32_t first = 0;                                // first used step (1-based), 0 = none
#define LCB_SCAN64(x, base) if (!first && n > (base)) { const uint32_t c_ = n - (base) < 64u ? n - (base) : 64u; const uint64_t m_ = (x) & (c_ == 64u ? ~0ull : ((1ull << c_) - 1ull)); if (m_) first = (base) + (uint32_t)__ffsll((long long)m_); }
                LCB_SCAN64(x0, 0u) LCB_SCAN64(x1, 64u) LCB_SCAN64(x2, 128u) LCB_SCAN64(x3, 192u)
#undef LCB_SCAN64
                for (uint32_t d = have + 1; !first && d <= Lu; d++)    // (windows of more than 225 steps: -b beyond 225)
                    if (lcb_used_bit(S.U, (Flat)(up ? fb + (d - 1u) : fb - (d - 1u)))) first = d;
                if (first) { nv = first - 1u; brk = 1u; }
            }
        }
        if (isV && !tryUsed) {
            // steps 1 .. nv read used == 0 (the read-out widens by one position for the - strand's bit g - 1)
            const uint32_t ge = up ? g0 + nv : g0 - nv;
            const uint32_t fs = lcb_fp_slot(S, vI);
            atomicMin(&S.fpLo[fs], ge);
            atomicMax(&S.fpHi[fs], ge);
        }
        if (STATS && isV) S.cWalk += nv + brk;                      // loop iterations entered: contributing steps plus the breaking one
        // ---- stage B: items
        // one item: steps c * 64 + 1 ... of voter b into the vote table
        auto consume = [&](uint32_t b, uint32_t c, int32_t id) {
            if (PROF) S.pfChunks++;
            const uint32_t bn = lcb_rl(nv, b), bf = lcb_rl(fl, b);
            const uint32_t d = c * 64 + S.lane + 1;
            bool act = d <= bn;
            const int32_t vid = (bf & LCB_FLAG_POS) ? id : -id;
            if (exact) {
                const unsigned long long hitM = __ballot(act && lcb_path_contains(S, vid));
                if (hitM) {
                    // the walk of this voter ends at its first vertex that is in the path: the steps before it vote, it is the breaking step
                    const uint32_t fh = (uint32_t)__ffsll((long long)hitM) - 1u;
                    act = act && S.lane < fh;
                    if (S.lane == b) {
                        const uint32_t nn = c * 64 + fh;
                        if (STATS) S.cWalk += (uint64_t)(nn + 1u) - (uint64_t)(nv + brk);
                        nv = nn; brk = 1u;
                    }
                }
            }
            const uint32_t wgt = lcb_rl(weight, b), e = lcb_rl(vE, b);
            if (act) {
                uint32_t h = lcb_hash(vid, S.voteShift);
                int32_t old = atomicCAS(&S.vKey[h], LCB_EMPTY_KEY, vid);
                uint32_t probe = 0;
                while (old != LCB_EMPTY_KEY && old != vid && probe < S.voteCap) {
                    h = (h + 1) & vmask; probe++;
                    old = atomicCAS(&S.vKey[h], LCB_EMPTY_KEY, vid);
                }
                if (old == LCB_EMPTY_KEY) {
                    const uint32_t tt = atomicAdd(S.vNClaimed, 1u);
                    if (tt < claimCap) S.vTouched[tt] = (typename ST::Idx)h; else *S.vOvf = 1u;
                }
                if (old == LCB_EMPTY_KEY || old == vid) {
                    atomicAdd(&S.vCount[h], wgt);
                    atomicMax(&S.vLast[h], ((typename ST::VLast)e << ST::LAST_SHIFT) | d);
                } else *S.vOvf = 1u;
            }
        };
        // the items behind the speculative ones, in voter order: chunk 1 ... of the first LCB_VB voters, every chunk of the others
        unsigned long long rest = walkM;
        uint32_t curB = 0, curC = 0, curN = 0, ord = 0;
        bool curOn = false;
        auto nextItem = [&](uint32_t& b, uint32_t& c) -> bool {
            for (;;) {
                if (curOn) {
                    curN = lcb_rl(nv, curB);                    // (an exact pass may have shortened it)
                    if (curC * 64 < curN) { b = curB; c = curC++; return true; }
                    curOn = false;
                }
                if (!rest) return false;
                curB = (uint32_t)__ffsll((long long)rest) - 1u; rest &= rest - 1;
                curC = ord < (uint32_t)LCB_VB ? 1u : 0u; ord++;
                curOn = true;
            }
        };
        for (;;) {
#pragma unroll
            for (int q = 0; q < LCB_VB; q++) if (itV[q]) consume(itB[q], itC[q], itId[q]);
#pragma unroll
            for (int q = 0; q < LCB_VB; q++) {
                itV[q] = nextItem(itB[q], itC[q]);
                itId[q] = 0;
                if (itV[q]) itId[q] = request(itB[q], itC[q]);
            }
            if (!itV[0]) break;
        }
    }
}

// Is one of the vertices named by the entries [s0, s1) of the touched list of the vote table in the path? (after a pass with
// exact = false, see lcb_vote_walk; wave-uniform result)
template <class ST>
__device__ inline bool lcb_vote_any_in_path(const ST& S, uint32_t s0, uint32_t s1)
{
    bool hit = false;
    for (uint32_t q = s0 + S.lane; q < s1; q += 64) if (lcb_path_contains(S, S.vKey[S.vTouched[q]])) hit = true;
    return __ballot(hit) != 0;
}

// Arg-max over the entries [s0, s1) of the touched list of the vote table, wave-wide: max count; ties -> smallest origin (strand, g) of the
// last contributing instance in list order; ties -> earliest step. Equal counts are the NORMAL case (collinear voters
// give every vertex of their common window the same total), so the tie key is always computed.
struct LcbBest { uint32_t cnt, keyHi, keyLo; int32_t vid; uint32_t e; };

template <class ST>
__device__ inline LcbBest lcb_vote_argmax(const ST& S, bool forward, bool useGood, uint32_t s0, uint32_t s1)
{
    uint32_t bCnt = 0, bHi = 0xFFFFFFFFu, bLo = 0xFFFFFFFFu, bE = 0;
    int32_t bVid = 0;
    for (uint32_t q = s0 + S.lane; q < s1; q += 64) {
        const uint32_t t = S.vTouched[q];
        const int32_t key = S.vKey[t];
        const uint32_t cnt = S.vCount[t];
        const typename ST::VLast last = S.vLast[t];
        const uint32_t e = (uint32_t)(last >> ST::LAST_SHIFT), d = (uint32_t)(last & (((typename ST::VLast)1 << ST::LAST_SHIFT) - 1));
        const uint32_t i = useGood ? S.good[e] : e;
        const uint32_t g0 = forward ? S.iBackG[i] : S.iFrontG[i];
        // JunctionSequentialIterator::operator< (junctionstorage.h:349-362): negative strand first, then chr, idx = (segment, g);
        // the step d has 16 bits (lcb_device_create checks -b)
        const uint32_t afl = S.iFlags[i];
        const uint32_t hi = ((afl & LCB_FLAG_POS) << 31) | (ST::SEG ? lcb_fl_seg(afl) << 26 : 0u) | (g0 >> 6), lo = (g0 << 26) | d;
        if (cnt > bCnt || (cnt == bCnt && (hi < bHi || (hi == bHi && lo < bLo)))) { bCnt = cnt; bHi = hi; bLo = lo; bVid = key; bE = e; }
    }
    LcbBest r;
    r.cnt = lcb_wave_umax(bCnt);
    bool c = bCnt == r.cnt && r.cnt != 0;
    r.keyHi = lcb_wave_umin(c ? bHi : 0xFFFFFFFFu);
    c = c && bHi == r.keyHi;
    r.keyLo = lcb_wave_umin(c ? bLo : 0xFFFFFFFFu);
    c = c && bLo == r.keyLo;
    const unsigned long long m = __ballot(c);
    const uint32_t w = m ? (uint32_t)__ffsll((long long)m) - 1u : 0u;
    r.vid = m ? lcb_rl(bVid, w) : 0;
    r.e = lcb_rl(bE, w);
    return r;
}

// Mailbox words through which wave 0 hands a vote to the helper wavefronts of its workgroup.
enum { LCB_MAIL_CMD = 0, LCB_MAIL_FLAGS, LCB_MAIL_NLIST, LCB_MAIL_FLANK, LCB_MAIL_NTOUCH, LCB_MAIL_FPSPLIT, LCB_MAIL_FPSHIFT, LCB_MAIL_WORDS = 8 };
enum { LCB_CMD_VOTE = 1, LCB_CMD_EXIT = 2 };

// Clears the vote-table slots named by the entries [s0, s1) of the touched list (blocksfinder.h:761-766).
template <class ST>
__device__ inline void lcb_vote_clear(ST& S, uint32_t s0, uint32_t s1)
{
    for (uint32_t q = s0 + S.lane; q < s1; q += 64) { const uint32_t t = S.vTouched[q]; S.vKey[t] = LCB_EMPTY_KEY; S.vCount[t] = 0; S.vLast[t] = 0; }
}

// Votes with at least this many voters are also reduced and cleared by all wavefronts of the workgroup (two more
// barriers); smaller ones by wave 0 alone while the helpers already wait for the next vote. Decided before the walks, so
// that every wavefront knows which protocol the vote follows.
#ifndef LCB_VOTE_SHARE_MIN
#define LCB_VOTE_SHARE_MIN 24u         // (the emulator tests also build with a tiny threshold to walk the shared path)
#endif

// One wave's share of a large vote after the walks: partial arg-max over its slice of the touched list, then (after
// everyone has read) the clearing of that slice.
template <int NW, class ST>
__device__ inline void lcb_vote_reduce_slice(ST& S, bool forward, bool useGood, uint32_t waveId, uint32_t nTouched, bool exact)
{
    const uint32_t per = (nTouched + NW - 1) / NW;
    const uint32_t s0 = waveId * per < nTouched ? waveId * per : nTouched, s1 = s0 + per < nTouched ? s0 + per : nTouched;
    const bool hit = !exact && lcb_vote_any_in_path(S, s0, s1);      // a pass without path stops: is a vertex of this slice in the path?
    const LcbBest b = lcb_vote_argmax(S, forward, useGood, s0, s1);
    if (S.lane == 0) {
        uint32_t* p = S.part + 8 * waveId;
        p[0] = b.cnt; p[1] = b.keyHi; p[2] = b.keyLo; p[3] = (uint32_t)b.vid; p[4] = b.e; p[5] = hit ? 1u : 0u;
    }
    __syncthreads();                                           // C: every slice has been read, partials are visible
    lcb_vote_clear(S, s0, s1);
    __syncthreads();                                           // D: the table is clean before wave 0 votes again (possibly on its own)
}

// One pass of a vote over the voters of the touch list: walks, arg-max, clearing of the table. `hit`: the pass walked without
// path stops and one of the vertices it touched is in the path (or the touched list is incomplete) - its result is void.
// (One call site of the walk and one of the arg-max: the compiler inlines everything into the kernel, and until round 5 a vote's code
// existed thirteen times per instantiation.)
template <bool STATS, int NW, bool PROF = false, class ST>
__device__ inline LcbBest lcb_vote_pass(ST& S, bool forward, bool tryUsed, bool useGood, bool exact, bool& ovfAny, bool& hit)
{
    const uint32_t claimCap = S.voteCap - (S.voteCap >> 2);
    LcbBest b;
    hit = false;
    const bool multi = NW > 1 && S.nTouch > 1;                     // the helper wavefronts take part
    const bool share = multi && S.nTouch >= LCB_VOTE_SHARE_MIN;    // ... also in the reduction and the clearing
    if (multi) {
        // wake the helper wavefronts: every wave walks its share of the voters
        if (S.lane == 0) {
            S.mail[LCB_MAIL_FLAGS] = (forward ? 1u : 0u) | (tryUsed ? 2u : 0u) | (useGood ? 4u : 0u) | (share ? 8u : 0u) | (exact ? 16u : 0u);
            S.mail[LCB_MAIL_NTOUCH] = S.nTouch;
            S.mail[LCB_MAIL_FPSPLIT] = S.fpSplit; S.mail[LCB_MAIL_FPSHIFT] = S.fpShift;
            if (!LCB_WALK_V2_OF(ST)) *S.vTicket = 0;
            S.mail[LCB_MAIL_CMD] = LCB_CMD_VOTE;
        }
        __syncthreads();                                           // A
    }
    const uint64_t tw0 = PROF ? wall_clock64() : 0;
    lcb_vote_walk<STATS, PROF>(S, forward, tryUsed, useGood, 0, multi ? (uint32_t)NW : 1u, exact);
    const uint64_t tw1 = PROF ? wall_clock64() : 0;
    if (multi) __syncthreads();                                    // B: all walks done
    else LCB_SYNC_IF(LcbCfg<ST::MODE>::IDX_LDS);
    const uint64_t tw2 = PROF ? wall_clock64() : 0;
    if (PROF && !LCB_PROF_PUSH) { S.pfTWalk += tw1 - tw0; S.pfTWaitB += tw2 - tw1; }
    uint32_t nTouched = lcb_rfl(*S.vNClaimed);
    if (nTouched > claimCap) nTouched = claimCap;
    if (STATS && multi && S.lane == 0) { S.cWalk += *S.mailWalk; *S.mailWalk = 0; }
    // wave 0's slice of the touched list: all of it unless the reduction is shared
    const uint32_t per = share ? (nTouched + NW - 1) / NW : nTouched;
    const uint32_t s1 = per < nTouched ? per : nTouched;
    hit = !exact && lcb_vote_any_in_path(S, 0, s1);               // a pass without path stops: is a vertex of this slice in the path?
    b = lcb_vote_argmax(S, forward, useGood, 0, s1);
    if (share) {
        if (S.lane == 0) {
            uint32_t* p = S.part;
            p[0] = b.cnt; p[1] = b.keyHi; p[2] = b.keyLo; p[3] = (uint32_t)b.vid; p[4] = b.e; p[5] = hit ? 1u : 0u;
        }
        __syncthreads();                                           // C: every slice has been read, partials are visible
        lcb_vote_clear(S, 0, s1);
        __syncthreads();                                           // D: the table is clean before wave 0 votes again (possibly on its own)
        // final reduction over the NW partial results
        uint32_t pc = 0, ph = 0xFFFFFFFFu, pl = 0xFFFFFFFFu, pv = 0, pe = 0, pHit = 0;
        if (S.lane < (uint32_t)NW) { const uint32_t* p = S.part + 8 * S.lane; pc = p[0]; ph = p[1]; pl = p[2]; pv = p[3]; pe = p[4]; pHit = p[5]; }
        hit = __ballot(pHit != 0) != 0;
        b.cnt = lcb_wave_umax(pc);
        bool c = pc == b.cnt && b.cnt != 0;
        b.keyHi = lcb_wave_umin(c ? ph : 0xFFFFFFFFu); c = c && ph == b.keyHi;
        b.keyLo = lcb_wave_umin(c ? pl : 0xFFFFFFFFu); c = c && pl == b.keyLo;
        const unsigned long long m = __ballot(c);
        const uint32_t w = m ? (uint32_t)__ffsll((long long)m) - 1u : 0u;
        b.vid = m ? (int32_t)lcb_rl(pv, w) : 0;
        b.e = lcb_rl(pe, w);
    } else {
        LCB_SYNC_IF(LcbCfg<ST::MODE>::IDX_LDS);
        lcb_vote_clear(S, 0, s1);
    }
    if (PROF && !LCB_PROF_PUSH) S.pfTReduce += wall_clock64() - tw2;
    ovfAny = lcb_rfl(*S.vOvf) != 0;
    if (S.lane == 0) { *S.vNClaimed = 0; *S.vOvf = 0; }
    LCB_SYNC_IF(LcbCfg<ST::MODE>::IDX_LDS);
    return b;
}

// MostPopularVertex (blocksfinder.h:708-768); the second attempt of the forward extension with `used` positions allowed
// (blocksfinder.h:782-785, forward only, Q2) is the same code run again: attempts = 2.
template <bool STATS, bool PROF, int NW, class ST>
__device__ inline int32_t lcb_vote(ST& S, bool forward, uint32_t attempts, uint32_t& originInst)
{
    const bool useGood = S.nGood >= 2;                             // blocksfinder.h:713
    const uint32_t nList = useGood ? S.nGood : S.nInst;
    originInst = 0;
    // path set behind a Bloom filter: the first pass walks without path stops and is verified afterwards (lcb_vote_walk)
    constexpr bool DEFER = LcbCfg<ST::MODE>::BW != 0;
    // (Tried and removed, profiles/r03: handing paths with dozens of voters from the compact to the wide variant through the overflow
    // ladder. With low thresholds 10^5-10^6 seeds of a config-3 pass leave for the one-seed-per-CU wide variant (17-47 % slower); with
    // 32 voters past 1 024 pushes config 3 gains 1 % and the k = 25 workloads, whose repeat families have a hundred voters, lose 8-28 %.)
    for (uint32_t attempt = 0; attempt < attempts; attempt++) {
        const bool tryUsed = attempt != 0;
        if (STATS && S.lane == 0) S.cVote++;
        if (PROF) S.pfVote++;
        if (nList == 0 || S.nTouch == 0) continue;                 // nobody votes: nothing to walk, nothing to clear
        const uint64_t cWalk0 = S.cWalk;
        bool ovfAny = false, hit = false;
        if (PROF) S.pfTouchSum += S.nTouch;
        LcbBest b;
        for (uint32_t pass = 0;; pass++) {
            b = lcb_vote_pass<STATS, NW, PROF>(S, forward, tryUsed, useGood, pass != 0 || !DEFER, ovfAny, hit);
            if (pass != 0 || !DEFER || !(hit || ovfAny)) break;
            // a touched vertex is in the path (or the table overflowed, so that the touched list is incomplete - a pass with path stops
            // may touch fewer vertices): the vote as the reference walks it. After an overflow the table may hold stale keys.
            if (ovfAny) {
                // (no helper is inside the table: they all wait at barrier A for the next vote)
                for (uint32_t h = S.lane; h < S.voteCap; h += 64) { S.vKey[h] = LCB_EMPTY_KEY; S.vCount[h] = 0; S.vLast[h] = 0; }
                LCB_SYNC_IF(LcbCfg<ST::MODE>::IDX_LDS);
            }
            if (STATS) S.cWalk = cWalk0;
            if (PROF) S.pfMaxProbe++;             // (instrumented variant: number of repeated votes)
        }
        if (ovfAny) { S.status = LCB_ST_VOTE_OVF; return 0; }
        if (b.cnt == 0) continue;
        originInst = useGood ? lcb_rfl((uint32_t)S.good[b.e]) : b.e;
        return b.vid;
    }
    return 0;
}

// One occurrence of the pushed vertex: its record plus the three `used` words around it, so that IsUsed and (almost
// always) the Compatible gap test need no further global loads.
// g0: the g of bit 0 of uw0 (modulo 2^32: only differences g - g0 < 96 are used); sb: base of the occurrence's segment
template <class F> struct LcbOccT { uint4 rec; uint32_t lo, hi, g0, uw0, uw1, uw2; F sb; };

template <class J>
__device__ __forceinline__ uint4 lcb_load_rec(const LcbTables& T, J j, bool active)
{
    uint4 r = uint4{0u, 0u, 0u, 0u};
    if (active) r = T.occRec[j];
    return r;
}

template <class ST>
__device__ __forceinline__ LcbOccT<typename ST::Flat> lcb_finish_occ(const ST& S, const uint4& rec, bool active)
{
    const LcbTables& T = S.T;
    const LcbUsed& U = S.U;
    typedef typename ST::Flat Flat;
    LcbOccT<Flat> o;
    o.rec = rec; o.lo = 0; o.hi = 0; o.g0 = 0; o.uw0 = o.uw1 = o.uw2 = 0; o.sb = 0;
    if (active) {
        const uint2 lh = T.chrLoHi[rec.y & LCB_CHR_MASK];
        o.lo = lh.x; o.hi = lh.y;
        o.sb = lcb_seg_base(S, lcb_cw_seg(rec.y));
        const Flat f = o.sb + rec.x;
        const uint32_t wi = (uint32_t)(f >> 5);
        const uint32_t wbase = wi ? wi - 1 : 0;
        o.g0 = rec.x - ((uint32_t)f & 31u) - (wi ? 32u : 0u);
        if (U.tab) {
            // the three words around the occurrence almost always share its page: one table entry serves them
            const uint32_t pg = wi >> LCB_PAGE_SHIFT, d = U.tab[pg];
            const uint32_t w1 = wbase + 1, w2 = wbase + 2;
            o.uw0 = U.live[wbase + ((wbase >> LCB_PAGE_SHIFT) == pg ? d : U.tab[wbase >> LCB_PAGE_SHIFT])];
            o.uw1 = U.live[w1 + ((w1 >> LCB_PAGE_SHIFT) == pg ? d : U.tab[w1 >> LCB_PAGE_SHIFT])];
            o.uw2 = U.live[w2 + ((w2 >> LCB_PAGE_SHIFT) == pg ? d : U.tab[w2 >> LCB_PAGE_SHIFT])];
        } else { o.uw0 = U.live[wbase]; o.uw1 = U.live[wbase + 1]; o.uw2 = U.live[wbase + 2]; }
    }
    return o;
}

// (The three words are combined with shifts, never selected by a computed index: a select among members of the struct turns into a
// dynamically indexed load, which keeps the whole LcbOcc in scratch memory - 64 B per lane written and read back in every push.)
template <class F>
__device__ __forceinline__ bool lcb_occ_bit(const LcbOccT<F>& o, uint32_t g)
{
    const uint32_t off = g - o.g0;                          // 0..95 for g and g-1
    const uint64_t lo = (uint64_t)o.uw0 | ((uint64_t)o.uw1 << 32);
    return off < 64u ? ((lo >> off) & 1ull) != 0 : ((o.uw2 >> (off - 64u)) & 1u) != 0;
}

// lcb_range_any_used over [a, b), served from the cached words when the range lies inside them.
template <class F>
__device__ __forceinline__ bool lcb_range_any_used_c(const LcbUsed& U, const LcbOccT<F>& o, uint32_t a, uint32_t b)
{
    if (a >= b) return false;
    const uint32_t oa = a - o.g0, ob = b - o.g0;                              // bit offsets into the 96 cached bits: oa < ob <= 96 if the range lies inside
    if (oa >= 96u || ob > 96u || ob < oa) return lcb_range_any_used(U, (F)(o.sb + a), (F)(o.sb + b));
    const uint64_t lo = (uint64_t)o.uw0 | ((uint64_t)o.uw1 << 32);
    bool any = false;
    if (oa < 64u) {
        const uint32_t e = ob < 64u ? ob : 64u;                               // bits [oa, e) of the low 64
        const uint64_t m = (e == 64u ? ~0ull : ((1ull << e) - 1ull)) & ~((1ull << oa) - 1ull);
        any = (lo & m) != 0;
    }
    if (ob > 64u) {
        const uint32_t s = oa > 64u ? oa - 64u : 0u, e = ob - 64u;            // bits [s, e) of the third word, e <= 32
        const uint32_t m = (e == 32u ? 0xFFFFFFFFu : ((1u << e) - 1u)) & ~((1u << s) - 1u);
        any = any || (o.uw2 & m) != 0;
    }
    return any;
}

// What a push needs to know about the walked edge (wave-uniform, in scalar registers).
template <class J> struct LcbEdgeT { uint32_t gIt, segIt; bool itPositive; int32_t idIt, idN; uint32_t posIt, posN; int32_t ech; J o0; uint32_t nOcc; };   // occurrences [o0, o0 + nOcc) of the pushed vertex

// ---- a push: PointPushBack / PointPushFront with their workers (path.h:430-602) ------------------
// Per-occurrence outcomes.
#define LCB_ACT_NONE 0u
#define LCB_ACT_SKIP 1u      // Within() the upper-bound instance -> `continue`
#define LCB_ACT_EXT_P 2u     // extend the predecessor instance
#define LCB_ACT_EXT_X 3u     // extend the upper-bound instance
#define LCB_ACT_INSERT 4u    // new single-point instance

// BACK=true:  PointPushBack(e), e = OutgoingEdge of iterator (gIt, itPositive): vertex = end vertex.
// BACK=false: PointPushFront(e), e = IngoingEdge of iterator (gIt, itPositive): vertex = start vertex.
// rec0: the occurrence records o0 + lane of the pushed vertex, already loaded by the caller.
// Returns false iff the vertex is already in the path (path.h:571-574,589-592).
template <bool BACK, bool STATS, bool PROF, class ST>
__device__ inline bool lcb_push(ST& S, const LcbEdgeT<typename ST::OccIdx>& E, bool record, const uint4& rec0)
{
    const LcbTables& T = S.T;
    typedef typename ST::Flat Flat;
    typedef typename ST::OccIdx OccIdx;
    constexpr bool HOIST = !LcbCfg<ST::MODE>::INST_LDS;
    const uint64_t tq0 = (PROF && LCB_PROF_PUSH) ? wall_clock64() : 0;
    const int32_t vertex = E.itPositive ? E.idN : -E.idN;            // pushed vertex
    const int32_t otherVertex = E.itPositive ? E.idIt : -E.idIt;     // e.GetEndVertex() for a front push
    const OccIdx o0 = E.o0, o1 = E.o0 + E.nOcc;
    // the dependent loads of the first chunk (chromosome start, `used` words) fly while the path set is updated
    LcbOccT<Flat> occ = lcb_finish_occ(S, rec0, S.lane < E.nOcc);
    bool inPath = LcbCfg<ST::MODE>::BW ? lcb_bloom_maybe(S, vertex) : true;   // (an LDS-resident set is probed directly)
    if (inPath) { uint32_t probes = 0; inPath = lcb_path_probe(S, vertex, probes); if (PROF && probes > S.pfMaxProbe) S.pfMaxProbe = probes; }
    if (inPath) return false;
    const uint32_t length = lcb_absdiff(E.posN, E.posIt);
    const int32_t ech = E.ech;
    const int64_t dist64 = BACK ? (int64_t)S.rightFlank + length : (int64_t)S.leftFlank - (int64_t)length;
    if (dist64 > INT32_MAX || dist64 < -(int64_t)INT32_MAX) { S.status = LCB_ST_DIST_OVF; return false; }
    const int32_t distance = (int32_t)dist64;
    lcb_path_insert(S, vertex);
    if (S.status) return false;

    const int64_t B = S.P.maxBranch;
    uint32_t nTouch = 0;                                             // instances this push extends or creates
    if (PROF && LCB_PROF_PUSH) S.pfTWalk += wall_clock64() - tq0;
    for (OccIdx base = o0; base < o1; base += 64) {
        const uint64_t tq1 = (PROF && LCB_PROF_PUSH) ? wall_clock64() : 0;
        const OccIdx j = base + S.lane;
        const bool active = j < o1;
        const uint32_t n = S.nInst;
        const uint32_t* oKey = S.ordKey + S.cur * S.instCap;
        const typename ST::Idx* oIdx = S.ordIdx + S.cur * S.instCap;
        uint32_t g = 0, chr = 0, lo = 0, pos = 0, u = 0, cand = 0, act = LCB_ACT_NONE;
        uint32_t stCall = 0, stStep = 0;                             // stats: Compatible calls / walk steps of this occurrence
        uint32_t lenBefore = 0, lenAfter = 0;
        bool positive = false, usedS = false, usesP = false;
        if (base != o0) occ = lcb_finish_occ(S, lcb_load_rec(T, j, active), active);
        if (active) {
            if (STATS) S.cOcc++;
            g = occ.rec.x; chr = occ.rec.y; pos = occ.rec.z;
            positive = ((int32_t)occ.rec.w == vertex);               // JunctionIterator::IsPositiveStrand
            // instanceSet.upper_bound(Instance(seqIt, 0)): first key > g (LDS work first: the chromosome bounds and `used`
            // words of the occurrence are still in flight).
            // (Measured and removed, profiles/r04/ab_fourth.txt: a wave-uniform binary descent with both neighbours and the candidate
            // read unconditionally - three rounds of independent reads instead of five dependent ones - was 1 % SLOWER on every
            // workload: a step is bound by the instructions the seed's wavefront issues, not by its LDS round trips.)
            uint32_t a = 0, b = n;
            if (ST::SEG) { const uint32_t sg = lcb_cw_seg(chr); a = S.segFirst[sg]; b = S.segFirst[sg + 1]; }   // the entries of the occurrence's segment
            while (a < b) { const uint32_t mid = (a + b) >> 1; if (g < oKey[mid]) b = mid; else a = mid + 1; }
            u = a;
            uint32_t x = 0, p = 0, xFl = 0, pFl = 0;
            if (u < n) x = oIdx[u];
            if (u > 0) p = oIdx[u - 1];
            usesP = BACK ? positive : !positive;
            // Instance fields in the HBM workspace (big, huge): everything the classification and the extension may read of the two
            // neighbours is requested at once - one round trip instead of the three to five dependent ones of the step-by-step code
            // below (flags -> Within -> candidate -> length before -> length after), which is what an LDS-resident pool wants.
            uint32_t hXF = 0, hXB = 0, hCG = 0, hCP = 0, hCO = 0;
            int32_t hCD = 0;
            if (HOIST) {
                const uint32_t c0 = usesP ? p : x;
                const bool hasC = usesP ? u > 0 : u < n;
                if (u < n) { xFl = S.iFlags[x]; hXF = S.iFrontG[x]; hXB = S.iBackG[x]; }
                if (u > 0) pFl = S.iFlags[p];
                if (hasC) {
                    hCG = BACK ? S.iBackG[c0] : S.iFrontG[c0]; hCP = BACK ? S.iBackPos[c0] : S.iFrontPos[c0];
                    hCD = BACK ? S.iBackDist[c0] : S.iFrontDist[c0]; hCO = BACK ? S.iFrontPos[c0] : S.iBackPos[c0];
                }
            } else {
                if (u < n) xFl = S.iFlags[x];
                if (u > 0) pFl = S.iFlags[p];
            }
            const bool hasX = u < n && (xFl >> LCB_FLAG_BITS) == chr;
            const bool hasP = u > 0 && (pFl >> LCB_FLAG_BITS) == chr;
            bool skip = false;
            if (hasX) {                                              // Instance::Within (path.h:170-175)
                const uint32_t f = HOIST ? hXF : S.iFrontG[x], bk = HOIST ? hXB : S.iBackG[x];
                skip = g >= (f < bk ? f : bk) && g <= (f < bk ? bk : f);
            }
            if (skip) act = LCB_ACT_SKIP;
            else {
                const bool has = usesP ? hasP : hasX;
                bool compat = false;
                uint32_t cFl = 0;
                if (has) {
                    cand = usesP ? p : x;
                    cFl = usesP ? pFl : xFl;
                    if (STATS) stCall = 1;
                    const bool cpos = (cFl & LCB_FLAG_POS) != 0;
                    if (cpos == positive) {                          // path.h:382-385
                        const uint32_t cg = HOIST ? hCG : (BACK ? S.iBackG[cand] : S.iFrontG[cand]);
                        const uint32_t cp = HOIST ? hCP : (BACK ? S.iBackPos[cand] : S.iFrontPos[cand]);
                        const int32_t cd = HOIST ? hCD : (BACK ? S.iBackDist[cand] : S.iFrontDist[cand]);
                        lenBefore = lcb_absdiff(hCO, hCP); lenAfter = lcb_absdiff(hCO, pos);   // (HOIST: RealLength of the candidate, and once extended)
                        // Compatible(start, end, e): BACK: start = cand.Back(), end = seqIt; FRONT: start = seqIt, end = cand.Front()
                        const int64_t startPos = BACK ? cp : pos, endPos = BACK ? pos : cp;
                        const int64_t realDiff = positive ? endPos - startPos : startPos - endPos;
                        const int64_t ancestralDiff = BACK ? (int64_t)distance - cd : (int64_t)cd - distance;
                        const uint32_t ga = g < cg ? g : cg, gb = g < cg ? cg : g;
                        if (STATS) stStep = lcb_range_walk_steps(S.U, (Flat)(occ.sb + ga), (Flat)(occ.sb + gb), positive);
                        bool okDist = realDiff >= 0;
                        if (okDist && (realDiff > B || ancestralDiff > B)) {
                            // only an exact next-edge continuation is accepted (path.h:407-411,420-424)
                            const uint32_t gs = BACK ? cg : g, ge = BACK ? g : cg;      // start, end
                            const bool adjacent = positive ? (ge == gs + 1) : (gs == ge + 1);
                            okDist = adjacent && (int32_t)lcb_it_char(T, (Flat)(occ.sb + gs), positive) == ech;
                            if (okDist && !BACK) {
                                const int32_t idE = T.posId[(Flat)(occ.sb + ge)];
                                okDist = (positive ? idE : -idE) == otherVertex;       // start1.GetVertexId() == e.GetEndVertex()
                            }
                        }
                        compat = okDist && !lcb_range_any_used_c(S.U, occ, ga, gb);
                        // `inst->Back().GetVertexId() != vertex` (path.h:541; :472 for the front): a candidate that already ends at
                        // the pushed vertex — it was inserted or extended by an occurrence of an EARLIER 64-lane chunk of this very
                        // push (within a chunk the prefix rule below does the same) — is not extended again: else branch.
                        // Equal path distance <=> same vertex (distances are strictly monotone along the path).
                        if (cd == distance) compat = false;
                    }
                }
                lo = occ.lo;
                usedS = positive ? lcb_occ_bit(occ, g) : (g > lo ? lcb_occ_bit(occ, g - 1) : false);   // JunctionSequentialIterator::IsUsed
                if (compat) {
                    const bool fin = (cFl & (BACK ? LCB_FLAG_BACKFIN : LCB_FLAG_FRONTFIN)) != 0;
                    act = fin ? LCB_ACT_NONE : (usesP ? LCB_ACT_EXT_P : LCB_ACT_EXT_X);
                } else act = usedS ? LCB_ACT_NONE : LCB_ACT_INSERT;
            }
        }
        const uint64_t tq2 = (PROF && LCB_PROF_PUSH) ? wall_clock64() : 0;
        if (PROF && LCB_PROF_PUSH) S.pfTWaitB += tq2 - tq1;
        // Occurrences that fall into the same gap (same chromosome, same upper bound u) interact sequentially in the
        // reference; lanes are ordered by g, so a gap is a contiguous lane segment:
        //  * after the first EXT_X in the gap every later occurrence is Within() that instance -> SKIP;
        //  * after the first INSERT / EXT_P in the gap the predecessor of later predecessor-using
        //    occurrences ends at the pushed vertex -> they take the else branch (insert if unused).
        const uint32_t uPrev = lcb_bcast(u, S.lane ? S.lane - 1 : 0);
        const uint32_t chrPrev = lcb_bcast(chr, S.lane ? S.lane - 1 : 0);
        const bool segStart = active && (S.lane == 0 || u != uPrev || chr != chrPrev);   // a gap belongs to one chromosome's set
        const unsigned long long startM = __ballot(segStart);
        const unsigned long long touchM = __ballot(active && (act == LCB_ACT_INSERT || act == LCB_ACT_EXT_P));
        const unsigned long long extXM = __ballot(active && act == LCB_ACT_EXT_X);
        if (active) {
            const unsigned long long below = (1ull << S.lane) - 1;
            const unsigned long long mineStart = startM & (below | (1ull << S.lane));
            const uint32_t s0 = 63u - (uint32_t)__clzll((long long)mineStart);   // my segment's first lane
            const unsigned long long seg = below & ~((1ull << s0) - 1);           // earlier lanes of my segment
            if (extXM & seg) { act = LCB_ACT_SKIP; stCall = 0; stStep = 0; }    // `continue` before Compatible
            else if (usesP && act != LCB_ACT_SKIP && (touchM & seg)) act = usedS ? LCB_ACT_NONE : LCB_ACT_INSERT;
        }
        if (STATS) {
            // the reference evaluates Compatible against the instance the latest earlier occurrence of the
            // gap inserted or extended (it ends at the pushed vertex, so the outcome is the else branch)
            const unsigned long long finalTouchM = __ballot(active && (act == LCB_ACT_INSERT || act == LCB_ACT_EXT_P));
            const unsigned long long below = (1ull << S.lane) - 1;
            const unsigned long long mineStart = startM & (below | (1ull << S.lane));
            const uint32_t s0 = mineStart ? 63u - (uint32_t)__clzll((long long)mineStart) : 0u;
            const unsigned long long prior = active ? (finalTouchM & below & ~((1ull << s0) - 1)) : 0ull;
            const uint32_t t = prior ? 63u - (uint32_t)__clzll((long long)prior) : 0u;
            const uint32_t gT = lcb_bcast(g, t);
            const uint32_t posT = lcb_bcast(positive ? 1u : 0u, t);
            if (active && usesP && prior && !(extXM & below & ~((1ull << s0) - 1)) && act != LCB_ACT_SKIP) {
                stCall = 1;
                stStep = (posT != 0) == positive ? lcb_range_walk_steps(S.U, (Flat)(occ.sb + gT), (Flat)(occ.sb + g), positive) : 0u;
            }
            S.cCompatCall += stCall; S.cCompatStep += stStep;
        }
        // apply
        const bool ins = active && act == LCB_ACT_INSERT;
        const bool ext = active && (act == LCB_ACT_EXT_P || act == LCB_ACT_EXT_X);
        bool becameGood = false;
        if (ext) {
            const int64_t before = HOIST ? (int64_t)lenBefore : lcb_real_length(S, cand);
            if (BACK) {                                              // Instance::ChangeBack (path.h:124-133)
                S.iBackG[cand] = g; S.iBackPos[cand] = pos; S.iBackDist[cand] = distance;
                if (positive) S.ordKey[S.cur * S.instCap + u - 1] = g;   // compareIdx_ follows the + strand back
                if (usedS) S.iFlags[cand] |= LCB_FLAG_BACKFIN;
            } else {                                                 // Instance::ChangeFront (path.h:113-122)
                S.iFrontG[cand] = g; S.iFrontPos[cand] = pos; S.iFrontDist[cand] = distance;
                if (!positive) S.ordKey[S.cur * S.instCap + u - 1] = g;  // compareIdx_ follows the - strand front
                if (usedS) S.iFlags[cand] |= LCB_FLAG_FRONTFIN;
            }
            const uint32_t fs = lcb_fp_slot(S, cand);
            if (HOIST) { atomicMin(&S.fpLo[fs], g); atomicMax(&S.fpHi[fs], g); }      // (no result: nothing waits for the workspace)
            else { if (g < S.fpLo[fs]) S.fpLo[fs] = g; if (g > S.fpHi[fs]) S.fpHi[fs] = g; }
            becameGood = before < (int64_t)S.P.minBlock && (HOIST ? (int64_t)lenAfter : lcb_real_length(S, cand)) >= (int64_t)S.P.minBlock;
        }
        if (ST::SEG) {
            // a new instance that has to share the footprint slot of an older pool entry (no slot of its own left) must lie in that slot's
            // segment: the slot's hull is (segment, 32-bit offsets), and a hull of another segment would not cover this instance's reads
            bool clash = false;
            const unsigned long long im = __ballot(ins);
            if (ins) {
                const uint32_t i = S.nInst + (uint32_t)__popcll(im & ((1ull << S.lane) - 1));
                const uint32_t fs = lcb_fp_slot(S, i);
                clash = fs < S.nFp && lcb_fp_get_seg(S, fs) != lcb_cw_seg(chr);
            }
            if (__ballot(clash)) { S.status = LCB_ST_INST_OVF; return true; }
        }
        const unsigned long long insM = __ballot(ins);
        const unsigned long long extM = __ballot(ext);
        const unsigned long long goodM = __ballot(becameGood);
        const uint32_t m = (uint32_t)__popcll(insM);
        if (S.nInst + m > S.instCap) { S.status = LCB_ST_INST_OVF; return true; }
        if (becameGood) {
            const uint32_t at = S.nGood + (uint32_t)__popcll(goodM & ((1ull << S.lane) - 1));
            S.good[at] = (typename ST::Idx)cand; S.goodPos[cand] = (typename ST::Idx)at;
        }
        S.nGood += (uint32_t)__popcll(goodM);
        if (ext) S.touch[nTouch + (uint32_t)__popcll(extM & ((1ull << S.lane) - 1))] = (typename ST::Idx)cand;
        nTouch += (uint32_t)__popcll(extM);
        if (ins) {
            const uint32_t r = (uint32_t)__popcll(insM & ((1ull << S.lane) - 1));
            const uint32_t i = S.nInst + r;
            S.iFrontG[i] = g; S.iBackG[i] = g; S.iFrontPos[i] = pos; S.iBackPos[i] = pos;
            S.iFrontDist[i] = distance; S.iBackDist[i] = distance; S.iLo[i] = lo;
            S.iHi[i] = occ.hi;
            S.iFlags[i] = (chr << LCB_FLAG_BITS) | (positive ? LCB_FLAG_POS : 0u);
            {   // the creating read (the occurrence's own `used` bit was 0)
                const uint32_t fs = lcb_fp_slot(S, i);
                if (fs >= S.nFp) { S.fpLo[fs] = g; S.fpHi[fs] = g; lcb_fp_set_seg(S, fs, lcb_cw_seg(chr)); }
#ifndef LCB_TEST_PLAIN_FP_READ
                else if (HOIST) { atomicMin(&S.fpLo[fs], g); atomicMax(&S.fpHi[fs], g); }     // (slots in the HBM workspace are only ever UPDATED with atomics: the walks of the other wavefronts update them at the L2)
#endif
                else { if (g < S.fpLo[fs]) S.fpLo[fs] = g; if (g > S.fpHi[fs]) S.fpHi[fs] = g; }
            }
            S.scr[r] = u; S.scr[64 + r] = g; S.scr[128 + r] = i;
            if (ST::SEG) S.scr[192 + r] = lcb_cw_seg(chr);
            S.goodPos[i] = ST::NONE;
            S.touch[nTouch + r] = (typename ST::Idx)i;
        }
        nTouch += m;
        S.nInst += m;
        if (m) { const uint32_t top = lcb_fp_slot(S, S.nInst - 1) + 1; if (top > S.nFp) S.nFp = top; if (S.nInst > S.nFp) S.nFp = S.nInst; }
        LCB_SYNC_IF(LcbCfg<ST::MODE>::INST_LDS && LcbCfg<ST::MODE>::IDX_LDS);
        const uint64_t tq3 = (PROF && LCB_PROF_PUSH) ? wall_clock64() : 0;
        if (PROF && LCB_PROF_PUSH) S.pfTReduce += tq3 - tq2;
        if (m) lcb_order_merge(S, m);
        if (PROF && LCB_PROF_PUSH) S.pfTScan += wall_clock64() - tq3;
    }
    if (BACK) {
        if (record) {
            if (S.nRight >= S.bodyCap) { S.status = LCB_ST_PATH_OVF; return true; }
            if (S.lane == 0) S.body[S.nRight] = ((unsigned long long)(E.itPositive ? 1u : 0u) << 40) | ((unsigned long long)E.segIt << 32) | E.gIt;
        }
        S.nRight++;
        S.rightFlank = distance;
    } else {
        S.nLeft++;
        S.leftFlank = distance;
    }
    S.nTouch = nTouch;
    if (STATS && S.lane == 0) S.cPush++;
    if (PROF) { S.pfPush++; if (S.nInst > S.pfMaxInst) S.pfMaxInst = S.nInst; }
    return true;
}

// Path::Score (path.h:604-628)
template <class ST>
__device__ inline int64_t lcb_score(const ST& S)
{
    int64_t sum = 0;
    bool bad = false;
    for (uint32_t e = S.lane; e < S.nGood; e += 64) {
        const uint32_t i = S.good[e];
        const int64_t rightPenalty = (int64_t)S.rightFlank - S.iBackDist[i];
        const int64_t leftPenalty = -(int64_t)S.leftFlank + S.iFrontDist[i];
        if (leftPenalty >= S.P.maxFlank || rightPenalty >= S.P.maxFlank) bad = true;
        else sum += lcb_real_length(S, i) - (rightPenalty + leftPenalty) * (rightPenalty + leftPenalty);
    }
    const bool anyBad = __ballot(bad) != 0;
    if (anyBad) return -(int64_t)INT32_MAX;
    if (S.nGood == 0) return 0;
    return lcb_wave_sum(sum);
}

// bestInstance <- goodInstance_ (blocksfinder.h:820-824,883-887)
template <class ST>
__device__ inline void lcb_snapshot(ST& S)
{
    if (S.nGood > S.bestCap) { S.status = LCB_ST_BEST_OVF; return; }
    for (uint32_t e = S.lane; e < S.nGood; e += 64) {
        const uint32_t i = S.good[e];
        const uint32_t fl = S.iFlags[i];
        uint4 r;
        r.x = (fl >> LCB_FLAG_BITS) & LCB_CHR_MASK; r.y = S.iFrontG[i] - S.iLo[i]; r.z = S.iBackG[i] - S.iLo[i]; r.w = (fl & LCB_FLAG_POS);
        S.best[e] = r;
    }
    S.nBest = S.nGood;
}

// Forward pushes only append instances, good-list entries and path vertices and only change the Back fields, the flags and
// the ordered index of existing instances; the best-scoring point only moves forward. A checkpoint taken at a best point
// therefore stays a valid restart for the replay whatever follows.
#define LCB_CK_EVERY 32u
template <class ST>
__device__ inline void lcb_checkpoint(ST& S)
{
    const uint32_t n = S.nInst, cap = S.bestCap;
    if (n > cap) return;
    uint32_t* c = S.ck;
    const uint32_t* oKey = S.ordKey + S.cur * S.instCap;
    const typename ST::Idx* oIdx = S.ordIdx + S.cur * S.instCap;
    for (uint32_t i = S.lane; i < n; i += 64) {
        c[i] = S.iBackG[i]; c[cap + i] = S.iBackPos[i]; c[2 * cap + i] = (uint32_t)S.iBackDist[i]; c[3 * cap + i] = S.iFlags[i];
        c[4 * cap + i] = oKey[i]; c[5 * cap + i] = oIdx[i];
    }
    if (ST::SEG && S.lane <= LCB_MAX_SEG) c[6 * cap + S.lane] = S.segFirst[S.lane];
    S.ckN = S.nRight; S.ckInst = n; S.ckGood = S.nGood; S.ckPath = S.nPath; S.ckFlank = S.rightFlank;
}

// Back to the checkpoint: the state after ckN forward pushes (Front fields, chromosome data and the right-body list are
// untouched by forward pushes; the Bloom filter and the footprints keep their supersets).
template <class ST>
__device__ inline void lcb_restore_checkpoint(ST& S)
{
    LCB_WAVE_SYNC();
    for (uint32_t i = S.ckPath + S.lane; i < S.nPath; i += 64) S.pKeys[S.pSlots[i]] = LCB_EMPTY_KEY;
    for (uint32_t e = S.ckGood + S.lane; e < S.nGood; e += 64) S.goodPos[S.good[e]] = ST::NONE;   // they became good after the checkpoint
    const uint32_t n = S.ckInst, cap = S.bestCap;
    const uint32_t* c = S.ck;
    uint32_t* oKey = S.ordKey + S.cur * S.instCap;
    typename ST::Idx* oIdx = S.ordIdx + S.cur * S.instCap;
    for (uint32_t i = S.lane; i < n; i += 64) {
        S.iBackG[i] = c[i]; S.iBackPos[i] = c[cap + i]; S.iBackDist[i] = (int32_t)c[2 * cap + i]; S.iFlags[i] = c[3 * cap + i];
        oKey[i] = c[4 * cap + i]; oIdx[i] = (typename ST::Idx)c[5 * cap + i];
    }
    if (ST::SEG && S.lane <= LCB_MAX_SEG) S.segFirst[S.lane] = c[6 * cap + S.lane];
    S.nPath = S.ckPath; S.nInst = n; S.nGood = S.ckGood; S.nRight = S.ckN; S.nLeft = 0;
    S.rightFlank = S.ckFlank; S.leftFlank = 0; S.nTouch = 0;        // (no vote follows before the next push or the backward phase)
    LCB_WAVE_SYNC();
}

// ---- edge batches ---------------------------------------------------------------------------------
// Up to 64 consecutive edges to push, one per lane: the table rows of both ends and the CSR bounds of the pushed vertex
// are fetched for all of them at once (two dependent global round trips per batch instead of three per push).
template <class J> struct LcbEdgeBatchT { uint32_t gIt, segIt, itPos, posIt, posN, nOcc; J o0; int32_t idIt, idN, ech; };

// Edges along a chromosome walk: edge l goes from position g + dir*l to g + dir*(l+1) (ExtendPathForward/Backward,
// blocksfinder.h:789-802,852-865). Returns the number of edges up to the first position whose vertex is `next`
// (64 if it is further away; the caller then asks for the following batch).
template <bool FORWARD, class ST>
__device__ inline uint32_t lcb_batch_from_walk(const ST& S, uint32_t seg, uint32_t g, int dir, bool positive, uint32_t lo, uint32_t hi, int32_t next, LcbEdgeBatchT<typename ST::OccIdx>& b)
{
    const LcbTables& T = S.T;
    typedef typename ST::Flat Flat;
    const Flat sb = ST::SEG ? (Flat)lcb_rfl(S.segBase[seg]) : (Flat)0;       // (seg is wave-uniform: the chromosome of the origin instance)
    const int64_t q0 = (int64_t)g + (int64_t)dir * (int64_t)S.lane, q1 = q0 + dir;
    const bool v0 = q0 >= (int64_t)lo && q0 < (int64_t)hi, v1 = q1 >= (int64_t)lo && q1 < (int64_t)hi;
    b.gIt = (uint32_t)q0; b.segIt = seg; b.itPos = positive ? 1u : 0u;
    b.idIt = 0; b.idN = 0; b.posIt = 0; b.posN = 0; b.ech = 0; b.o0 = 0; b.nOcc = 0;
    int32_t ch0 = 0, ch1 = 0;
    if (v0) { b.idIt = (T.posId + sb)[(uint32_t)q0]; b.posIt = (T.posPos + sb)[(uint32_t)q0]; ch0 = (int32_t)lcb_it_char(T, (Flat)(sb + (uint32_t)q0), positive); }
    if (v1) { b.idN = (T.posId + sb)[(uint32_t)q1]; b.posN = (T.posPos + sb)[(uint32_t)q1]; ch1 = (int32_t)lcb_it_char(T, (Flat)(sb + (uint32_t)q1), positive); }
    // e.GetChar(): outgoing -> char at the iterator, ingoing -> char at the previous position (junctionstorage.h:191-227)
    b.ech = FORWARD ? ch0 : ch1;
    // the walk stops at the first position whose vertex is `next`; positions past the chromosome end never match
    const unsigned long long hit = __ballot(v1 && (positive ? b.idN : -b.idN) == next);
    const unsigned long long out = __ballot(!v1);
    const unsigned long long endM = hit | out;
    const uint32_t firstEnd = endM ? (uint32_t)__ffsll((long long)endM) - 1u : 64u;
    uint32_t n = 64;
    if (hit && (uint32_t)__ffsll((long long)hit) - 1u == firstEnd) n = firstEnd + 1;   // edges 0 .. firstEnd end at `next`
    else if (firstEnd < 64) n = firstEnd;                                               // ran off the chromosome (cannot happen for a voted vertex)
    if (S.lane < n) {
        const uint32_t av = (uint32_t)(b.idN < 0 ? -b.idN : b.idN);
        b.o0 = lcb_occ_start<ST::SEG>(T, av); b.nOcc = (uint32_t)(lcb_occ_start<ST::SEG>(T, av + 1) - b.o0);
    }
    return n;
}

// Edges of the recorded right body [from, from + 64) for the replay (blocksfinder.h:271-284).
template <class ST>
__device__ inline void lcb_batch_from_body(const ST& S, uint32_t from, uint32_t n, LcbEdgeBatchT<typename ST::OccIdx>& b)
{
    const LcbTables& T = S.T;
    typedef typename ST::Flat Flat;
    b.gIt = 0; b.segIt = 0; b.itPos = 0; b.idIt = 0; b.idN = 0; b.posIt = 0; b.posN = 0; b.ech = 0; b.o0 = 0; b.nOcc = 0;
    if (S.lane < n) {
        const unsigned long long r = S.body[from + S.lane];
        b.gIt = (uint32_t)r; b.segIt = ST::SEG ? ((uint32_t)(r >> 32) & 0xFFu) : 0u; b.itPos = (uint32_t)(r >> 40);
        const Flat sb = lcb_seg_base(S, b.segIt);
        const Flat fIt = sb + b.gIt, fN = b.itPos ? fIt + 1 : fIt - 1;
        b.idIt = T.posId[fIt]; b.idN = T.posId[fN]; b.posIt = T.posPos[fIt]; b.posN = T.posPos[fN];
        b.ech = (int32_t)lcb_it_char(T, fIt, b.itPos != 0);
        const uint32_t av = (uint32_t)(b.idN < 0 ? -b.idN : b.idN);
        b.o0 = lcb_occ_start<ST::SEG>(T, av); b.nOcc = (uint32_t)(lcb_occ_start<ST::SEG>(T, av + 1) - b.o0);
    }
}

template <class J>
__device__ __forceinline__ LcbEdgeT<J> lcb_edge_of(const LcbEdgeBatchT<J>& b, uint32_t l)
{
    LcbEdgeT<J> e;
    e.gIt = lcb_rl(b.gIt, l); e.segIt = lcb_rl(b.segIt, l); e.itPositive = lcb_rl(b.itPos, l) != 0; e.idIt = lcb_rl(b.idIt, l); e.idN = lcb_rl(b.idN, l);
    e.posIt = lcb_rl(b.posIt, l); e.posN = lcb_rl(b.posN, l); e.ech = lcb_rl(b.ech, l); e.o0 = lcb_rl(b.o0, l); e.nOcc = lcb_rl(b.nOcc, l);
    return e;
}

// ExtendPathForward / ExtendPathBackward (blocksfinder.h:770-895)
template <bool FORWARD, bool STATS, bool PROF, int NW, class ST>
__device__ inline bool lcb_extend(ST& S, uint32_t& bestRightSize, int64_t& bestScore, int64_t& nowScore)
{
    const LcbTables& T = S.T;
    uint32_t oi = 0;
    LCB_MARK(S, 4, S.nRight); LCB_MARK(S, 5, S.nLeft); LCB_MARK(S, 6, 1);
    const uint64_t tv0 = PROF ? wall_clock64() : 0;
    // an asynchronous batch nobody waits for any more gives up here (the word is read past the L1: another stream writes it)
    if (S.abort && lcb_rfl(__hip_atomic_load(S.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) { S.status = LCB_ST_ABORTED; return false; }
    const int32_t next = lcb_vote<STATS, PROF, NW>(S, FORWARD, FORWARD ? 2u : 1u, oi);    // (forward: a second attempt over used positions, blocksfinder.h:782-785, Q2)
    LCB_MARK(S, 6, 2); LCB_MARK(S, 7, (uint32_t)next);
    if (S.status) return false;
    if (PROF) S.pfTVote += wall_clock64() - tv0;
    bool success = false;
    if (next != 0) {
        // (pool in the HBM workspace: one load for the four fields, see lcb_inst_fields; an LDS-resident pool is read field by field -
        // there the one-load form made the register allocator of the compact instantiation spill whole 16-register tuples)
        constexpr bool ONE = !LcbCfg<ST::MODE>::INST_LDS;
        const uint32_t f = ONE ? lcb_inst_fields(S, oi) : 0u;
        const uint32_t ofl = ONE ? lcb_rl(f, LCB_F_FLAGS) : lcb_rfl(S.iFlags[oi]);
        const bool positive = (ofl & LCB_FLAG_POS) != 0;
        const uint32_t oseg = ST::SEG ? lcb_fl_seg(ofl) : 0u;
        uint32_t g = ONE ? (FORWARD ? lcb_rl(f, LCB_F_BACKG) : lcb_rl(f, LCB_F_FRONTG)) : lcb_rfl(FORWARD ? S.iBackG[oi] : S.iFrontG[oi]);
        const int dir = (FORWARD == positive) ? 1 : -1;
        const uint32_t lo = ONE ? lcb_rl(f, LCB_F_LO) : lcb_rfl(S.iLo[oi]), hi = ONE ? lcb_rl(f, LCB_F_HI) : lcb_rfl(S.iHi[oi]);
        for (;;) {
            LcbEdgeBatchT<typename ST::OccIdx> bt;
            const uint32_t nE = lcb_batch_from_walk<FORWARD>(S, oseg, g, dir, positive, lo, hi, next, bt);
            if (nE == 0) break;                                      // defensive: a voted vertex is always reached
            // occurrence records of the first edge; those of edge l+1 are requested before edge l is pushed
            uint4 rec = lcb_load_rec(T, lcb_rl(bt.o0, 0) + S.lane, S.lane < lcb_rl(bt.nOcc, 0));
            for (uint32_t l = 0; l < nE; l++) {
                const LcbEdgeT<typename ST::OccIdx> E = lcb_edge_of(bt, l);
                uint4 recN = uint4{0u, 0u, 0u, 0u};
                if (l + 1 < nE) recN = lcb_load_rec(T, lcb_rl(bt.o0, l + 1) + S.lane, S.lane < lcb_rl(bt.nOcc, l + 1));
                LCB_MARK(S, 6, 3); LCB_MARK(S, 8, E.gIt);
                const uint64_t tp0 = PROF ? wall_clock64() : 0;
                success = lcb_push<FORWARD, STATS, PROF>(S, E, true, rec);
                const uint64_t tp1 = PROF ? wall_clock64() : 0;
                if (PROF) S.pfTPush += tp1 - tp0;
                LCB_MARK(S, 6, 4);
                if (S.status) return false;
                if (success) {
                    nowScore = lcb_score(S);
                    if (nowScore > bestScore) {
                        bestScore = nowScore;
                        if (FORWARD) bestRightSize = S.nRight + 1;
                        // bestInstance <- goodInstance_ (blocksfinder.h:820-824,883-887). Forward: the replay after the forward
                        // extension rebuilds exactly the state of the best point, so the copy is taken once, there
                        // (lcb_process_seed), not at every improvement on the way — a growing block improves at almost every push.
                        if (!FORWARD && nowScore > 0) { lcb_snapshot(S); if (S.status) return false; }
                        if (FORWARD && !STATS && S.nRight >= S.ckN + LCB_CK_EVERY) lcb_checkpoint(S);
                    }
                    if (PROF) S.pfTScore += wall_clock64() - tp1;
                }
                rec = recN;
            }
            if (nE < 64 || (positive ? lcb_rl(bt.idN, 63) : -lcb_rl(bt.idN, 63)) == next) break;
            g = (uint32_t)((int64_t)g + 64 * dir);
        }
    }
    return success;
}

// ProcessVertex::Process (blocksfinder.h:228-310)
template <int MODE, bool STATS, bool PROF, int NW, class ST>
__device__ inline void lcb_process_seed(ST& S, int32_t vid, int32_t ch, int64_t& bestScoreOut)
{
    int64_t score = 0, bestScore = 0;
    S.nBest = 0; S.status = LCB_ST_OK; S.ckN = 0; S.endInst = 0; S.fpSplit = 0xFFFFFFFFu; S.fpShift = 0;
    LCB_MARK(S, 2, 1);
    lcb_path_init<STATS>(S, vid, ch);
    LCB_MARK(S, 2, 2); LCB_MARK(S, 3, S.nInst);
    // No unused occurrence carries the seed's character: nothing can vote, nothing is pushed, the result is empty
    // (blocksfinder.h:781-786 finds no vertex in either direction). Stats mode walks through the motions for the counters.
    const bool dead = !STATS && !S.status && S.nInst == 0;
    uint32_t bestRightSize = 1;
    const int64_t minRun = 2 * (int64_t)S.P.maxBranch;
    if (!S.status && !dead) {
        for (;;) {                                                   // blocksfinder.h:255-269
            bool ret = true, positive = false;
            const int64_t prevLength = (int64_t)S.rightFlank - S.leftFlank;
            while ((ret = lcb_extend<true, STATS, PROF, NW>(S, bestRightSize, bestScore, score)) &&
                   ((int64_t)S.rightFlank - S.leftFlank) - prevLength <= minRun)
                positive = positive || (score > 0);
            if (!ret || !positive || S.status) break;
        }
    }
    LCB_MARK(S, 2, 3);
    if (!S.status && !dead) {                                        // replay, blocksfinder.h:271-284
        const uint32_t nEdge = bestRightSize - 1;
        uint32_t from = 0;
        if (!STATS && S.ckN && S.ckN <= nEdge) {
            lcb_restore_checkpoint(S);    // the state after ckN pushes; stats mode replays from Init like the reference (event counts)
            from = S.ckN;
        } else {
            lcb_path_clear(S);        // keeps the body list, resets everything else
            lcb_path_init<STATS>(S, vid, ch);
        }
        while (from < nEdge && !S.status) {
            const uint32_t nE = nEdge - from < 64 ? nEdge - from : 64;
            LcbEdgeBatchT<typename ST::OccIdx> bt;
            lcb_batch_from_body(S, from, nE, bt);
            uint4 rec = lcb_load_rec(S.T, lcb_rl(bt.o0, 0) + S.lane, S.lane < lcb_rl(bt.nOcc, 0));
            for (uint32_t l = 0; l < nE && !S.status; l++) {
                const LcbEdgeT<typename ST::OccIdx> E = lcb_edge_of(bt, l);
                uint4 recN = uint4{0u, 0u, 0u, 0u};
                if (l + 1 < nE) recN = lcb_load_rec(S.T, lcb_rl(bt.o0, l + 1) + S.lane, S.lane < lcb_rl(bt.nOcc, l + 1));
                lcb_push<true, STATS, PROF>(S, E, false, rec);
                rec = recN;
            }
            from += nE;
        }
        // the state is the one of the best forward point: its good instances are the best instances so far
        if (!S.status && bestScore > 0) lcb_snapshot(S);
        // pool entries created from here on are new instances, not re-creations: footprint slots of their own (lcb_fp_slot)
        S.fpSplit = S.nInst; S.fpShift = S.nFp - S.nInst;
    }
    LCB_MARK(S, 2, 4);
    if (!S.status && !dead) {
        // The first backward vote: the instances whose FRONT is the seed vertex (front distance 0 = left flank) are the
        // initial ones, pool entries [0, nInit) — back pushes leave fronts alone and create instances at distance > 0.
        for (uint32_t i = S.lane; i < S.nInit; i += 64) S.touch[i] = (typename ST::Idx)i;
        S.nTouch = S.nInit;
        LCB_WAVE_SYNC();
        for (;;) {                                                   // blocksfinder.h:292-306 (stray ';' at :297, Q1)
            bool ret = true;
            const int64_t prevLength = (int64_t)S.rightFlank - S.leftFlank;
            while ((ret = lcb_extend<false, STATS, PROF, NW>(S, bestRightSize, bestScore, score)) &&
                   ((int64_t)S.rightFlank - S.leftFlank) - prevLength <= minRun)
                ;
            const bool positive = score > 0;
            if (!ret || !positive || S.status) break;
        }
    }
    LCB_MARK(S, 2, 5);
    S.endInst = S.nInst;
    lcb_path_clear(S);            // Path::Clear (blocksfinder.h:308)
    bestScoreOut = bestScore;
}

// ---- the kernel --------------------------------------------------------------------------------
// Per-launch arguments that are only touched between seeds (work queue, result arenas). They are parked in LDS so that
// they do not occupy scalar registers for the whole kernel.
struct LcbLaunchArgs {
    const LcbKSeed* seeds;
    LcbSeedOut* out;
    LcbSeedCtr* ctr;
    uint4* arena;
    LcbFpOut* fpArena;
    unsigned long long arenaCap, fpCap, arenaBase, fpBase;
    unsigned long long* arenaCursor;
    unsigned long long* fpCursor;
    uint32_t* cursor;
    const uint32_t* live;
    uint32_t cursorBase, nSeeds;
    const uint32_t* usedTab;       // page table of the `used` view of the seed wave 0 is working on (read by the helpers at each vote)
};

// NW = wavefronts per workgroup: wave 0 runs the per-seed algorithm, waves 1..NW-1 are vote helpers.
template <int MODE, bool STATS, int NW, bool PROF, bool SEG = false>
__device__ inline void lcb_process_body(const LcbTables& T, const LcbKParams& P, const LcbKSeed* seeds, uint32_t nSeeds,
                                        const LcbWork& W, LcbSeedOut* out, uint4* arena, unsigned long long arenaCap,
                                        LcbFpOut* fpArena, unsigned long long fpCap)
{
    typedef LcbCfg<MODE> Cfg;
    constexpr bool INST_LDS = Cfg::INST_LDS, IDX_LDS = Cfg::IDX_LDS;
    constexpr uint32_t IC = Cfg::IC, VC = Cfg::VC, BW = Cfg::BW, PC = Cfg::PC;
    __shared__ uint32_t sInst[INST_LDS ? 9 * IC : 1];
    __shared__ uint32_t sFp[INST_LDS ? 2 * IC : 1];
    __shared__ uint32_t sOrdKey[IDX_LDS ? 2 * IC : 1];
    typedef typename Cfg::Idx Idx;
    typedef typename Cfg::VLast VLast;
    __shared__ Idx sOrdIdx[IDX_LDS ? 2 * IC : 1];
    __shared__ Idx sGood[IDX_LDS ? IC : 1];
    __shared__ Idx sGoodPos[IDX_LDS ? IC : 1];
    __shared__ Idx sTouch[IDX_LDS ? IC : 1];
    __shared__ int32_t sVKey[IDX_LDS ? VC : 1];
    __shared__ uint32_t sVCount[IDX_LDS ? VC : 1];
    __shared__ VLast sVLast[IDX_LDS ? VC : 1];
    __shared__ Idx sVTouched[IDX_LDS ? VC : 1];
    __shared__ uint32_t sBloom[BW ? BW : 1];
    __shared__ int32_t sPath[PC ? PC : 1];
    __shared__ uint32_t sScr[(SEG ? 4 : 3) * 64];
    __shared__ uint64_t sSegBase[SEG ? LCB_MAX_SEG : 1];
    __shared__ uint32_t sSegFirst[SEG ? LCB_MAX_SEG + 2 : 1];
    __shared__ uint8_t sFpSeg[(SEG && INST_LDS) ? IC : 1];
    __shared__ uint32_t sMisc[4];
    __shared__ uint32_t sMail[LCB_MAIL_WORDS];
    __shared__ uint32_t sPart[8 * NW];
    __shared__ unsigned long long sMailWalk[1];
    __shared__ LcbLaunchArgs sArgs;
    if (threadIdx.x == 0) {
        sArgs.seeds = seeds; sArgs.out = out; sArgs.ctr = W.ctr; sArgs.arena = arena; sArgs.fpArena = fpArena; sArgs.arenaCap = arenaCap; sArgs.fpCap = fpCap;
        sArgs.arenaBase = W.arenaBase; sArgs.fpBase = W.fpBase; sArgs.arenaCursor = W.arenaCursor; sArgs.fpCursor = W.fpCursor;
        sArgs.cursor = W.cursor; sArgs.cursorBase = W.cursorBase; sArgs.live = W.live; sArgs.nSeeds = W.live ? *W.nLive : nSeeds;
    }

    LcbStateT<MODE, SEG> S;
    S.T = T; S.P = P; S.U = lcb_used_of(T, 0);
    S.lane = threadIdx.x & 63u;
    S.segBase = sSegBase; S.segFirst = sSegFirst;
    if (SEG) {
        if (threadIdx.x < LCB_MAX_SEG) sSegBase[threadIdx.x] = threadIdx.x < T.nSeg ? T.segBase[threadIdx.x] : 0ull;
        if (threadIdx.x < LCB_MAX_SEG + 2) sSegFirst[threadIdx.x] = 0;
    }
    uint8_t* slot = W.base + (uint64_t)blockIdx.x * W.slotBytes;
    const LcbSlotLayout L = lcb_slot_layout(W.pathCap, W.bodyCap, W.bestCap, INST_LDS ? 0 : W.instCap, IDX_LDS ? 0 : W.voteCap);
    S.pSlots = (uint32_t*)(slot + L.pSlots);
    if (PC) { S.pKeys = sPath; S.pathCap = W.pathCap < PC ? W.pathCap : PC; for (uint32_t h = threadIdx.x; h < PC; h += 64 * NW) sPath[h] = LCB_EMPTY_KEY; }   // (a smaller capacity only in tests)
    else { S.pKeys = (int32_t*)(slot + L.pKeys); S.pathCap = W.pathCap; }
    S.pathShift = 32u - (uint32_t)__ffs((int)S.pathCap) + 1u;
    S.body = (unsigned long long*)(slot + L.body); S.bodyCap = W.bodyCap;
    S.best = (uint4*)(slot + L.best); S.bestCap = W.bestCap;
    S.ck = (uint32_t*)(slot + L.ck); S.ckN = S.ckInst = S.ckGood = S.ckPath = 0; S.ckFlank = 0;
    uint32_t* instBase;
    uint32_t instStride;                                           // words between the instance field arrays
    uint32_t *fpBase;
    if (INST_LDS) { instBase = sInst; instStride = IC; fpBase = sFp; }
    else { instBase = (uint32_t*)(slot + L.inst); instStride = W.instCap; fpBase = (uint32_t*)(slot + L.fp); }
    S.fpLo = fpBase; S.fpHi = fpBase + instStride;
    S.fpSeg8 = sFpSeg; S.fpSeg32 = INST_LDS ? nullptr : fpBase + 2 * instStride;
    if (IDX_LDS) {
        S.instCap = IC;
        S.ordKey = sOrdKey; S.ordIdx = sOrdIdx; S.good = sGood; S.goodPos = sGoodPos; S.touch = sTouch;
        S.vKey = sVKey; S.vCount = sVCount; S.vLast = sVLast; S.vTouched = sVTouched;
        S.voteCap = VC;
        for (uint32_t h = threadIdx.x; h < VC; h += 64 * NW) { sVKey[h] = LCB_EMPTY_KEY; sVCount[h] = 0; sVLast[h] = 0; }
    } else {
        S.instCap = W.instCap;
        S.ordKey = (uint32_t*)(slot + L.ordKey); S.ordIdx = (Idx*)(slot + L.ordIdx);
        S.good = (Idx*)(slot + L.good); S.goodPos = (Idx*)(slot + L.goodPos); S.touch = (Idx*)(slot + L.touch);
        S.vKey = (int32_t*)(slot + L.vKey); S.vCount = (uint32_t*)(slot + L.vCount);
        S.vLast = (VLast*)(slot + L.vLast); S.vTouched = (Idx*)(slot + L.vTouched);
        S.voteCap = W.voteCap;
    }
    S.bloom = sBloom;
    S.bloomShift = BW ? 32u - (uint32_t)__ffs((int)(BW * 32u)) + 1u : 0u;
    for (uint32_t h = threadIdx.x; h < BW; h += 64 * NW) sBloom[h] = 0;
    S.voteShift = 32u - (uint32_t)__ffs((int)S.voteCap) + 1u;
    // (field f of the pool lives at instBase + f * instStride, in the order of the LCB_F_* enumeration: lcb_inst_fields relies on it)
    S.iFrontG = instBase; S.iBackG = instBase + instStride;
    S.iFrontPos = instBase + 2 * instStride; S.iBackPos = instBase + 3 * instStride;
    S.iLo = instBase + 4 * instStride; S.iHi = instBase + 5 * instStride;
    S.iFlags = instBase + 6 * instStride;
    S.iFrontDist = (int32_t*)(instBase + 7 * instStride);
    S.iBackDist = (int32_t*)(instBase + 8 * instStride);
    S.scr = sScr; S.vNClaimed = &sMisc[0]; S.vOvf = &sMisc[1]; S.vTicket = &sMisc[2];
    S.mail = sMail; S.mailWalk = sMailWalk; S.part = sPart;
    const uint32_t waveId = lcb_rfl(threadIdx.x >> 6);
    S.dbg = (W.dbg && waveId == 0) ? W.dbg + 16u * blockIdx.x : nullptr;
    S.abort = W.abort;
    LCB_MARK(S, 0, 1);
    if (threadIdx.x == 0) { sMisc[0] = 0; sMisc[1] = 0; sMisc[2] = 0; sMail[LCB_MAIL_CMD] = 0; sMailWalk[0] = 0; }
    S.nInst = S.nGood = S.cur = S.nPath = S.nRight = S.nLeft = S.nBest = 0; S.status = 0; S.nTouch = S.nInit = 0;
    S.rightFlank = S.leftFlank = 0;
    S.cWalk = S.cOcc = S.cCompatCall = S.cCompatStep = S.cVote = S.cPush = 0;
    S.pfPush = S.pfVote = S.pfMaxProbe = S.pfMaxInst = 0; S.pfTVote = S.pfTPush = S.pfTScore = 0; S.nFp = 0;
    S.pfTWalk = S.pfTWaitB = S.pfTReduce = S.pfTScan = 0; S.pfVoters = S.pfChunks = S.pfTouchSum = 0;
    LCB_WAVE_SYNC();
    if (NW > 1) {
        __syncthreads();           // the LDS tables and sArgs above are initialised
        if (waveId != 0) {
            // helper wavefront: sleeps at the barrier until wave 0 posts a vote; walks its share of the voters, reduces and
            // clears its slice of the vote table
            for (;;) {
                __syncthreads();                                   // A
                if (lcb_rfl(S.mail[LCB_MAIL_CMD]) == LCB_CMD_EXIT) return;
                S.U.tab = sArgs.usedTab;
                const uint32_t flags = lcb_rfl(S.mail[LCB_MAIL_FLAGS]);
                S.nTouch = lcb_rfl(S.mail[LCB_MAIL_NTOUCH]);
                S.fpSplit = lcb_rfl(S.mail[LCB_MAIL_FPSPLIT]); S.fpShift = lcb_rfl(S.mail[LCB_MAIL_FPSHIFT]);
                S.cWalk = 0;
                lcb_vote_walk<STATS>(S, (flags & 1u) != 0, (flags & 2u) != 0, (flags & 4u) != 0, waveId, NW, (flags & 16u) != 0);
                if (STATS) {
                    const unsigned long long w = (unsigned long long)lcb_wave_sum((int64_t)S.cWalk);
                    if (S.lane == 0 && w) atomicAdd(S.mailWalk, w);
                }
                __syncthreads();                                   // B
                if (flags & 8u) {
                    // a vote with many voters is reduced and cleared by everyone (wave 0 resets the counter only after barrier D)
                    uint32_t nTouched = lcb_rfl(*S.vNClaimed);
                    const uint32_t claimCap = S.voteCap - (S.voteCap >> 2);
                    if (nTouched > claimCap) nTouched = claimCap;
                    lcb_vote_reduce_slice<NW>(S, (flags & 1u) != 0, (flags & 4u) != 0, waveId, nTouched, (flags & 16u) != 0);   // contains barriers C and D
                }
            }
        }
    } else LCB_WAVE_SYNC();

    for (;;) {
        uint32_t tk = 0;
        if (S.lane == 0) tk = atomicAdd(sArgs.cursor, 1u) - sArgs.cursorBase;   // every workgroup overshoots by exactly one ticket
        tk = lcb_rfl(tk);
        if (tk >= sArgs.nSeeds) break;
        const uint32_t s = sArgs.live ? lcb_rfl(sArgs.live[tk]) : tk;
        LCB_MARK(S, 1, s + 1);
        S.cWalk = S.cOcc = S.cCompatCall = S.cCompatStep = S.cVote = S.cPush = 0;
        S.nInst = S.nGood = S.cur = S.nRight = S.nLeft = 0; S.rightFlank = S.leftFlank = 0;
        S.nFp = 0;
        int64_t bestScore = 0;
        S.pfPush = S.pfVote = S.pfMaxProbe = S.pfMaxInst = 0; S.pfTVote = S.pfTPush = S.pfTScore = 0;
        S.pfTWalk = S.pfTWaitB = S.pfTReduce = S.pfTScan = 0; S.pfVoters = S.pfChunks = S.pfTouchSum = 0;
        const uint64_t tick0 = PROF ? wall_clock64() : 0;
        const LcbKSeed sd = sArgs.seeds[s];
        const int32_t vid = lcb_rfl(sd.vid), ch = lcb_rfl(sd.ch);
        S.U = lcb_used_of(T, lcb_rfl(sd.view));
        if (NW > 1 && S.lane == 0) sArgs.usedTab = S.U.tab;      // published to the helpers by the vote's first barrier
        lcb_process_seed<MODE, STATS, PROF, NW>(S, vid, ch, bestScore);
        if (S.status == LCB_ST_VOTE_OVF) {
            // the vote table may hold stale keys after an overflow (the helpers have cleared their slices; wave 0 wipes all)
            LCB_WAVE_SYNC();
            for (uint32_t h = S.lane; h < S.voteCap; h += 64) { S.vKey[h] = LCB_EMPTY_KEY; S.vCount[h] = 0; S.vLast[h] = 0; }
            if (S.lane == 0) { *S.vNClaimed = 0; *S.vOvf = 0; }
            LCB_WAVE_SYNC();
        }
        const uint64_t ticks = PROF ? wall_clock64() - tick0 : 0;
        const uint32_t n = S.status ? 0u : S.nBest;
        unsigned long long off = 0;
        if (n) {
            uint32_t olo = 0, ohi = 0;
            if (S.lane == 0) { off = atomicAdd(sArgs.arenaCursor, (unsigned long long)n) - sArgs.arenaBase; olo = (uint32_t)off; ohi = (uint32_t)(off >> 32); }
            olo = lcb_rfl(olo); ohi = lcb_rfl(ohi);
            off = ((unsigned long long)ohi << 32) | olo;
            if (off + n > sArgs.arenaCap) S.status = LCB_ST_ARENA_OVF;
            else { uint4* ar = sArgs.arena; for (uint32_t e = S.lane; e < n; e += 64) ar[off + e] = S.best[e]; }
        }
        // footprint intervals (one per instance ever created), widened by one position on the low side
        // because the - strand reads bit g-1
        LcbFpOut* fpa = sArgs.fpArena;
        const uint32_t nfp = (S.status == LCB_ST_OK && fpa) ? S.nFp : 0u;
        unsigned long long fpo = 0;
        if (nfp) {
            uint32_t olo = 0, ohi = 0;
            if (S.lane == 0) { fpo = atomicAdd(sArgs.fpCursor, (unsigned long long)nfp) - sArgs.fpBase; olo = (uint32_t)fpo; ohi = (uint32_t)(fpo >> 32); }
            olo = lcb_rfl(olo); ohi = lcb_rfl(ohi);
            fpo = ((unsigned long long)ohi << 32) | olo;
            if (fpo + nfp > sArgs.fpCap) S.status = LCB_ST_ARENA_OVF;
            else for (uint32_t e = S.lane; e < nfp; e += 64) {
                // (slots in the HBM workspace were updated with atomics at the L2 by every wavefront of the workgroup: read them there, not
                // through a line this wavefront's L1 may still hold from an earlier plain read)
#ifdef LCB_TEST_PLAIN_FP_READ      // (experiment build only: the read-out as it was before commit 0d6481f - can tests/test_gpu_segments.py see the stale line?)
                const uint32_t lo = S.fpLo[e], hi = S.fpHi[e];
#else
                const uint32_t lo = INST_LDS ? S.fpLo[e] : __hip_atomic_load(&S.fpLo[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t hi = INST_LDS ? S.fpHi[e] : __hip_atomic_load(&S.fpHi[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                const uint64_t sb = lcb_seg_base(S, lcb_fp_get_seg(S, e));
                LcbFpOut r; r.lo = sb + (lo ? lo - 1 : 0u); r.hi = sb + hi; fpa[fpo + e] = r;
            }
        }
        // (no local arrays here: the compiler would move them to LDS, 48 B x every lane of the workgroup)
        uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
        if (STATS) {
            c0 = (uint64_t)lcb_wave_sum((int64_t)S.cWalk); c1 = (uint64_t)lcb_wave_sum((int64_t)S.cOcc);
            c2 = (uint64_t)lcb_wave_sum((int64_t)S.cCompatCall); c3 = (uint64_t)lcb_wave_sum((int64_t)S.cCompatStep);
            c4 = (uint64_t)lcb_wave_sum((int64_t)S.cVote); c5 = (uint64_t)lcb_wave_sum((int64_t)S.cPush);
        }
        // The header is what the host polls while the kernel is still running (asynchronous batches): everything the seed wrote
        // - instances, footprints, the other header fields - is made visible system-wide before the status word is.
        if (S.lane == 0) {
            LcbSeedOut* o = sArgs.out + s;
            o->nInst = n;   // kept on ARENA_OVF so the host can track the allocator
            o->bestScore = bestScore; o->arenaOff = off;
            o->fpOff = fpo; o->nFp = nfp; o->poolInst = S.endInst;
        }
        // (only in launches whose headers the host polls: the release writes back the L2 of the whole XCD, 8 % of a config-3 pass
        // when every seed of every launch did it; a synchronous launch is read after its stream has drained)
        if (S.abort) __threadfence_system();
        if (S.lane == 0) {
            LcbSeedOut* o = sArgs.out + s;
            if (S.abort) __hip_atomic_store(&o->status, (uint32_t)S.status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            else o->status = S.status;
            if ((STATS || PROF) && sArgs.ctr) {
                uint64_t* k = sArgs.ctr[s].c;
                if (STATS) { k[0] = c0; k[1] = c1; k[2] = c2; k[3] = c3; k[4] = n; k[5] = c4; k[6] = c5; k[7] = 1; }
                else {
                    k[0] = ticks; k[1] = S.pfPush; k[2] = S.pfVote; k[3] = S.pfMaxProbe; k[4] = S.pfMaxInst; k[5] = S.pfTVote; k[6] = S.pfTPush; k[7] = S.pfTScore;
                    k[8] = S.pfTWalk; k[9] = S.pfTWaitB; k[10] = S.pfTReduce; k[11] = S.pfTScan; k[12] = S.pfVoters; k[13] = S.pfChunks; k[14] = S.pfTouchSum; k[15] = 0;
                }
            }
        }
        LCB_MARK(S, 2, 6);
    }
    if (NW > 1) {                  // release the helper wavefronts
        if (S.lane == 0) S.mail[LCB_MAIL_CMD] = LCB_CMD_EXIT;
        __syncthreads();
    }
    LCB_MARK(S, 0, 2);
}

// ---- screening --------------------------------------------------------------------------------------
// A seed none of whose occurrences is unused with the seed's character has an empty Path::Init (path.h:33-46), so its
// Process() returns nothing and reads no bit as 0: its header is final here. Later rounds consist almost entirely of such
// seeds (their neighbourhood is covered by committed blocks); the others are queued for the process kernel, and the host
// reads the headers of the queued seeds only. One thread per seed.
__device__ inline void lcb_screen_body(const LcbTables& T, const LcbKSeed* seeds, uint32_t nSeeds, LcbSeedOut* out, uint32_t* live, uint32_t* nLive)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    bool alive = false;
    if (s < nSeeds) {
        const LcbKSeed sd = seeds[s];
        const LcbUsed used = lcb_used_of(T, sd.view);
        const uint32_t av = (uint32_t)(sd.vid < 0 ? -sd.vid : sd.vid);
        const uint64_t o1 = T.occStart64 ? T.occStart64[av + 1] : (uint64_t)T.occStart32[av + 1];
        for (uint64_t j = T.occStart64 ? T.occStart64[av] : (uint64_t)T.occStart32[av]; j < o1 && !alive; j++) {
            const uint4 rec = T.occRec[j];
            const bool positive = (int32_t)rec.w == sd.vid;
            const uint64_t f = T.segBase[lcb_cw_seg(rec.y)] + rec.x;       // (a 32-entry table: cache-resident)
            bool isUsed;
            if (positive) isUsed = lcb_used_bit(used, f);
            else isUsed = rec.x > T.chrLoHi[rec.y & LCB_CHR_MASK].x ? lcb_used_bit(used, f - 1) : false;
            alive = !isUsed && (int32_t)(positive ? T.posCh[f] : T.posRevCh[f]) == sd.ch;
        }
        (void)out;      // a dead seed needs no header: the host looks at the live list only (its result is empty by definition)
    }
    // compact the live seeds in seed order within the wave (heavy seeds come first in the sorted seed list)
    const unsigned long long m = __ballot(alive);
    if (m) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(nLive, (uint32_t)__popcll(m));
        base = lcb_rfl(base);
        if (alive) live[base + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = s;
    }
}

#endif
