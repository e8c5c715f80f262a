// lcb_kernel.h — gfx950 device code of the per-seed path-extension / bubble-scoring hot path.
//
// One seed (BlocksFinder::Bundle) per workgroup: wavefront 0 runs the per-seed algorithm, the other wavefronts of the
// workgroup are helpers that share the look-ahead vote.
// This replaces ProcessVertex::Process (blocksfinder.h:228-310) and everything under it:
// MostPopularVertex (blocksfinder.h:708-768), ExtendPathForward/Backward (:770-895), Path::Init,
// PointPushBack/Front + workers, Compatible, Score, Clear (path.h:33-46,380-677) and
// DistanceKeeper (distancekeeper.h:9-41). It is a new design, not a translation:
//
//  * tables are structure-of-arrays over a FLAT position index g = chrStart[chr] + idx, so a
//    look-ahead walk is a coalesced read of consecutive posId/posPos words (lanes = walk steps);
//  * the per-chromosome std::multiset<Instance> becomes one key-sorted index array over an
//    insertion-ordered instance pool (pool order IS allInstance_ order) in LDS;
//  * the dense per-thread vote array count[2V+1] becomes an LDS hash table filled with
//    ds atomics; the order-dependent running arg-max of the reference is restated order-free
//    (max count, then smallest origin of the last contributing instance in list order, then
//    earliest walk step — SURVEY.md Q7) so all (instance, step) pairs can vote concurrently;
//  * the dense DistanceKeeper int[2V] becomes a per-wave open-addressing vertex set in global
//    memory; distances are carried by the instances (an instance's back/front distance IS the
//    path distance of its end vertex), so no distance lookups remain;
//  * a push evaluates all occurrences of the pushed vertex lane-parallel against the pre-push
//    state and resolves the (rare) occurrences that fall into the same gap between two
//    instances with a closed-form prefix rule that reproduces the sequential semantics;
//  * Compatible's unbounded `used` walk (path.h:387-393) becomes a masked bitmap range test
//    evaluated after the distance tests (it is a pure function);
//  * a seed reads the `used` VIEW its record names (the live bitmap or a predicted state built by the engine,
//    engine.cpp) and reports the FOOTPRINT of the bits it read as 0, which is what makes speculation exact;
//  * the replay of the forward extension (blocksfinder.h:271-284) restarts at a checkpoint taken at a best point.
//
// All cross-lane operations (__ballot/__shfl/LCB_WAVE_SYNC) sit in wave-uniform control flow.
// Integer arithmetic only; no MFMA — the work is indexing, not contraction.
#ifndef LCB_KERNEL_H
#define LCB_KERNEL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define LCB_EMPTY_KEY INT32_MIN
// The flight-recorder sites (LCB_MARK) are compiled into every kernel variant and cost one predictable branch each
// when the recorder is off (measured: 0.3 %). An earlier build hung or faulted on gfx950 with the sites compiled out
// (no longer reproducible); the recorder is what would localise a recurrence, so it stays in.
#define LCB_FLIGHT_RECORDER 1

// Three kernel variants by where the per-path state lives. Seeds that overflow one are re-run by the host in the next:
//   mode 0 "small":  instances + vote table in 79 KB of LDS  -> 2 workgroups per CU
//   mode 1 "medium": 2x the capacities in 154 KB of LDS       -> 1 workgroup per CU
//   mode 2 "big":    instances + vote table in the global-memory workspace, capacities chosen by the host
#define LCB_IC_SMALL 512u    // instances
#define LCB_VC_SMALL 2048u   // vote-table slots (power of two)
#define LCB_IC_MEDIUM 1024u
#define LCB_VC_MEDIUM 4096u
#define LCB_BLOOM_WORDS 1024u  // LDS Bloom filter in front of the path vertex set (32768 bits, 2 hashes)

enum LcbStatus : uint32_t {
    LCB_ST_OK = 0,
    LCB_ST_INST_OVF = 1,   // instance pool full
    LCB_ST_VOTE_OVF = 2,   // vote table full
    LCB_ST_PATH_OVF = 3,   // path vertex set / body full
    LCB_ST_BEST_OVF = 4,   // result snapshot buffer full
    LCB_ST_ARENA_OVF = 5,  // batch result arena full
    LCB_ST_DIST_OVF = 6,   // path distance does not fit 32 bits (unsupported, > 2 Gbp paths)
};

struct LcbTables {
    const uint32_t* chrStart;   // [nChr+1]
    const int32_t* posId;       // [nPos]  Position::id
    const uint32_t* posPos;     // [nPos]  Position::pos
    const uint8_t* posCh;       // [nPos]  seq[pos + k]              (JunctionSequentialIterator::GetChar, + strand)
    const uint8_t* posRevCh;    // [nPos]  ReverseChar(seq[pos - 1]) or 'N' at pos 0   (- strand)
    const uint32_t* occStart;   // [nVertex+1] CSR over |vertex id|
    const uint4* occRec;        // [nPos]  per occurrence, ascending in g: {g, chr, Position::pos, Position::id} (one 16-B load)
    const uint32_t* used;       // bitmap over g: bit g = Position::used of (chr, idx); view v of it starts at used + v * usedStride
    uint32_t usedStride;        // words between consecutive `used` views (view 0 = the live state, views 1.. = predicted states)
    uint32_t nChr, nVertex, nPos;
};

struct LcbKParams { int32_t k, minBlock, maxBranch, maxFlank, depth; };
struct LcbKSeed { int32_t vid; int32_t ch; uint32_t view; uint32_t pad; };   // view: which `used` view this seed reads

struct LcbSeedOut {            // per-seed header written by the kernel
    uint32_t nInst;
    uint32_t status;
    int64_t bestScore;
    uint64_t arenaOff;
    uint64_t fpOff;            // first footprint interval of this seed in the footprint arena
    uint32_t nFp;              // number of footprint intervals (= instances ever created)
    uint32_t pad;
    uint64_t ctr[8];           // lcb_counters order in stats mode, a cheap profile otherwise
};

struct LcbWork {               // per-wave global-memory workspace slots
    uint8_t* base;
    uint64_t slotBytes;
    uint32_t pathCap;          // power of two
    uint32_t bodyCap;
    uint32_t bestCap;
    uint32_t instCap;          // big mode only
    uint32_t voteCap;          // big mode only, power of two
    uint32_t* cursor;          // work-queue head: monotone ticket counter, never reset ...
    uint32_t cursorBase;       // ... tickets of this launch are [cursorBase, cursorBase + nSeeds)
    unsigned long long* arenaCursor;   // monotone result-arena allocator
    unsigned long long arenaBase;
    unsigned long long* fpCursor;      // monotone footprint-arena allocator
    unsigned long long fpBase;
    uint32_t* dbg;             // optional flight recorder: 16 words per workgroup (host watchdog prints them), or null
};

// ---- workspace layout (shared by host and device) -------------------------------------------
struct LcbSlotLayout {
    uint64_t pKeys, pSlots, body, best, ck;                   // always (ck: forward-extension checkpoint, 6 words per instance)
    uint64_t inst, ordKey, ordIdx, good, vKey, vCount, vLast, vTouched, fp;   // big mode
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
    L.ck = o; o = lcb_align16(o + 6ull * 4 * bestCap);
    L.inst = o; o = lcb_align16(o + 10ull * 4 * instCap);
    L.ordKey = o; o = lcb_align16(o + 2ull * 4 * instCap);
    L.ordIdx = o; o = lcb_align16(o + 2ull * 4 * instCap);
    L.good = o; o = lcb_align16(o + 4ull * instCap);
    L.vKey = o; o = lcb_align16(o + 4ull * voteCap);
    L.vCount = o; o = lcb_align16(o + 4ull * voteCap);
    L.vLast = o; o = lcb_align16(o + 8ull * voteCap);
    L.vTouched = o; o = lcb_align16(o + 4ull * voteCap);
    L.fp = o; o = lcb_align16(o + 2ull * 4 * instCap);
    L.total = lcb_align16(o);
    return L;
}

// ---- wave primitives ---------------------------------------------------------------------------
// Orders LDS/global accesses between the lanes of the (single) wavefront of the workgroup.
#define LCB_WAVE_SYNC()                                           \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");    \
    } while (0)

// Flight recorder: lane 0 stores progress words the host watchdog can read while the kernel is running.
#define LCB_MARK(S, slot, value)                                                         \
    do {                                                                                 \
        if (LCB_FLIGHT_RECORDER && (S).dbg && (S).lane == 0) __hip_atomic_store(&(S).dbg[(slot)], (uint32_t)(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
    } while (0)

#define LCB_FLAG_POS 1u
#define LCB_FLAG_BACKFIN 2u
#define LCB_FLAG_FRONTFIN 4u

struct LcbState {
    LcbTables T;
    LcbKParams P;
    uint32_t lane;
    // instance pool (SoA) — pool index order == allInstance_ order (path.h:684)
    uint32_t *iFrontG, *iBackG, *iFrontPos, *iBackPos, *iChr, *iLo, *iHi, *iFlags;
    int32_t *iFrontDist, *iBackDist;
    uint32_t* ordKey;          // instance_ ordered sets flattened: keys (flat compare position) ... [2][instCap], half `cur` is live
    uint32_t* ordIdx;          // ... and pool indices, double buffered the same way
    uint32_t* good;            // goodInstance_ (path.h:685), pool indices in append order
    // Footprint: per instance ever created, the range of flat positions whose `used` bit was read as 0 and could
    // have changed the result (its span plus every look-ahead window walked from its ends). A result computed
    // against an older `used` snapshot is still exact iff no bit inside these ranges has been set since
    // (bits only go 0 -> 1 and the computation is deterministic). Survives the replay's Clear.
    uint32_t *fpLo, *fpHi;
    uint32_t nFp;
    uint32_t instCap;
    // vote table
    int32_t* vKey;
    uint32_t* vCount;
    unsigned long long* vLast; // (list ordinal << 32) | step of the last contribution
    uint32_t* vTouched;
    uint32_t* vNTouched;       // LDS counter
    uint32_t voteCap, voteShift;
    uint32_t* scr;             // LDS scratch, 4 * 64 words
    uint32_t* bloom;           // LDS Bloom filter over the path vertex set
    uint32_t* mail;            // LDS mailbox to the helper wavefronts (LCB_MAIL_*)
    unsigned long long* mailWalk;   // stats: walk steps counted by the helpers
    uint32_t nWaves;           // wavefronts in this workgroup (1 = no helpers)
    // path vertex set + bodies + result snapshot (global workspace)
    int32_t* pKeys;
    uint32_t* pSlots;
    uint32_t pathCap, pathShift;
    unsigned long long* body;  // right body: (strand << 32) | g of the iterator whose outgoing edge was pushed
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
    uint32_t pfPush, pfVote, pfMaxProbe, pfMaxInst;   // cheap always-on per-seed profile (wave-uniform)
    uint64_t pfTVote, pfTPush, pfTScore;              // 10 ns ticks spent in the vote / push / score+snapshot sections
    // per-lane event counters (stats mode)
    uint64_t cWalk, cOcc, cCompatCall, cCompatStep, cVote, cPush;
};

__device__ __forceinline__ uint32_t lcb_hash(int32_t vid, uint32_t shift)
{
    return ((uint32_t)vid * 2654435761u) >> shift;
}

__device__ __forceinline__ bool lcb_used_bit(const uint32_t* used, uint32_t g)
{
    return (used[g >> 5] >> (g & 31)) & 1u;
}

// JunctionSequentialIterator::IsUsed (junctionstorage.h:270-283): `used` marks the edge idx -> idx+1.
__device__ __forceinline__ bool lcb_it_used(const LcbTables& T, uint32_t g, bool positive, uint32_t lo)
{
    if (positive) return lcb_used_bit(T.used, g);
    return g > lo ? lcb_used_bit(T.used, g - 1) : false;
}

// JunctionSequentialIterator::GetChar (junctionstorage.h:234-243)
__device__ __forceinline__ uint8_t lcb_it_char(const LcbTables& T, uint32_t g, bool positive)
{
    return positive ? T.posCh[g] : T.posRevCh[g];
}

// Any used bit in [a, b)?  This is the `used` walk of Path::Compatible (path.h:387-393) for both strands:
// + strand visits bits a..b-1 going up, - strand visits bits b-1..a going down.
__device__ inline bool lcb_range_any_used(const uint32_t* used, uint32_t a, uint32_t b)
{
    if (a >= b) return false;
    const uint32_t wa = a >> 5, wb = (b - 1) >> 5;
    const uint32_t ma = 0xFFFFFFFFu << (a & 31), mb = 0xFFFFFFFFu >> (31 - ((b - 1) & 31));
    if (wa == wb) return (used[wa] & ma & mb) != 0;
    if (used[wa] & ma) return true;
    for (uint32_t w = wa + 1; w < wb; w++)
        if (used[w]) return true;
    return (used[wb] & mb) != 0;
}

// Number of iterations the reference's walk makes over [a, b) (stats mode only).
__device__ inline uint32_t lcb_range_walk_steps(const uint32_t* used, uint32_t a, uint32_t b, bool up)
{
    uint32_t steps = 0;
    if (up) { for (uint32_t g = a; g < b; g++) { steps++; if (lcb_used_bit(used, g)) break; } }
    else { for (uint32_t g = b; g > a; g--) { steps++; if (lcb_used_bit(used, g - 1)) break; } }
    return steps;
}

__device__ __forceinline__ uint32_t lcb_bcast(uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)src); }

__device__ __forceinline__ int64_t lcb_wave_sum(int64_t v)
{
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)v >> 32), o);
        v += (int64_t)(((uint64_t)hi << 32) | lo);
    }
    return v;
}

// ---- path vertex set (DistanceKeeper::IsSet / Set / Unset, distancekeeper.h:17-35) --------------
// Exact open-addressing set in the wave's global workspace, fronted by an LDS Bloom filter: the common
// answer during a look-ahead walk is "not in the path", which the filter gives from LDS without touching
// global memory; only filter hits probe the exact table.
__device__ __forceinline__ uint32_t lcb_bloom1(int32_t vid) { return ((uint32_t)vid * 2654435761u) >> 17; }
__device__ __forceinline__ uint32_t lcb_bloom2(int32_t vid) { return ((uint32_t)vid * 0xC2B2AE35u + 0x27D4EB2Fu) >> 17; }

__device__ __forceinline__ bool lcb_bloom_maybe(const LcbState& S, int32_t vid)
{
    const uint32_t a = lcb_bloom1(vid), b = lcb_bloom2(vid);
    return ((S.bloom[a >> 5] >> (a & 31)) & (S.bloom[b >> 5] >> (b & 31)) & 1u) != 0;
}

__device__ inline bool lcb_path_probe(const LcbState& S, int32_t vid, uint32_t& probes)
{
    uint32_t h = lcb_hash(vid, S.pathShift);
    const uint32_t mask = S.pathCap - 1;
    for (uint32_t probe = 0; probe < S.pathCap; probe++) {      // the set is at most half full; the bound only guards a corrupted table
        const int32_t k = S.pKeys[h];
        if (k == vid || k == LCB_EMPTY_KEY) { probes = probe; return k == vid; }
        h = (h + 1) & mask;
    }
    probes = S.pathCap;
    return false;
}

__device__ __forceinline__ bool lcb_path_contains(const LcbState& S, int32_t vid)
{
    if (!lcb_bloom_maybe(S, vid)) return false;
    uint32_t probes;
    return lcb_path_probe(S, vid, probes);
}

// same, also reporting the probe length (profiling); wave-uniform callers only
__device__ __forceinline__ bool lcb_path_contains_p(LcbState& S, int32_t vid)
{
    if (!lcb_bloom_maybe(S, vid)) return false;
    uint32_t probes = 0;
    const bool r = lcb_path_probe(S, vid, probes);
    if (probes > S.pfMaxProbe) S.pfMaxProbe = probes;
    return r;
}

// Wave-uniform: inserts vid (not present). Lane 0 writes.
__device__ inline void lcb_path_insert(LcbState& S, int32_t vid)
{
    if ((S.nPath + 1) * 2 > S.pathCap) { S.status = LCB_ST_PATH_OVF; return; }
    uint32_t h = lcb_hash(vid, S.pathShift);
    const uint32_t mask = S.pathCap - 1;
    uint32_t probe = 0;
    while (S.pKeys[h] != LCB_EMPTY_KEY && probe < S.pathCap) { h = (h + 1) & mask; probe++; }
    if (probe == S.pathCap) { S.status = LCB_ST_PATH_OVF; return; }
    LCB_WAVE_SYNC();               // every lane has finished probing before lane 0 publishes the key
    if (S.lane == 0) {
        S.pKeys[h] = vid; S.pSlots[S.nPath] = h;
        const uint32_t a = lcb_bloom1(vid), b = lcb_bloom2(vid);
        S.bloom[a >> 5] |= 1u << (a & 31);
        S.bloom[b >> 5] |= 1u << (b & 31);
    }
    S.nPath++;
    LCB_WAVE_SYNC();
}

// Path::Clear (path.h:650-677): wave-uniform. The right-body list is kept (the replay reads it).
__device__ inline void lcb_path_clear(LcbState& S)
{
    for (uint32_t i = S.lane; i < S.nPath; i += 64) S.pKeys[S.pSlots[i]] = LCB_EMPTY_KEY;
    if (S.nPath) for (uint32_t i = S.lane; i < LCB_BLOOM_WORDS; i += 64) S.bloom[i] = 0;
    S.nPath = 0; S.nRight = 0; S.nLeft = 0; S.nInst = 0; S.nGood = 0; S.cur = 0;
    S.rightFlank = 0; S.leftFlank = 0;
    LCB_WAVE_SYNC();
}

// ---- instance helpers ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lcb_absdiff(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

// Instance::RealLength (path.h:165-168): |front.GetPosition() - back.GetPosition()| (k cancels).
__device__ __forceinline__ int64_t lcb_real_length(const LcbState& S, uint32_t i)
{
    return (int64_t)lcb_absdiff(S.iFrontPos[i], S.iBackPos[i]);
}

// Rebuilds the ordered index after m inserts of one chunk. The insert list (position in the OLD order,
// key, pool index) is in scr[0..m), scr[64..64+m), scr[128..128+m), ascending by position then key.
__device__ inline void lcb_order_merge(LcbState& S, uint32_t m)
{
    const uint32_t n = S.nInst - m;     // old element count (nInst already includes the new ones)
    const uint32_t c = S.cur, d = c ^ 1;
    const uint32_t* ck = S.ordKey + c * S.instCap; const uint32_t* ci = S.ordIdx + c * S.instCap;
    uint32_t* dk = S.ordKey + d * S.instCap; uint32_t* di = S.ordIdx + d * S.instCap;
    for (uint32_t i = S.lane; i < n; i += 64) {
        uint32_t cnt = 0;
        for (uint32_t r = 0; r < m; r++) cnt += (S.scr[r] <= i) ? 1u : 0u;
        dk[i + cnt] = ck[i];
        di[i + cnt] = ci[i];
    }
    if (S.lane < m) {
        const uint32_t at = S.scr[S.lane] + S.lane;
        dk[at] = S.scr[64 + S.lane];
        di[at] = S.scr[128 + S.lane];
    }
    S.cur = d;
    LCB_WAVE_SYNC();
}

// Path::Init (path.h:33-46): one instance per unused occurrence of vid whose next character is ch.
template <bool STATS>
__device__ inline void lcb_path_init(LcbState& S, int32_t vid, int32_t ch)
{
    const LcbTables& T = S.T;
    lcb_path_insert(S, vid);          // distanceKeeper_.Set(vid, 0)
    if (S.status) return;
    const uint32_t av = (uint32_t)(vid < 0 ? -vid : vid);
    const uint32_t o0 = T.occStart[av], o1 = T.occStart[av + 1];
    for (uint32_t base = o0; base < o1; base += 64) {
        const uint32_t j = base + S.lane;
        bool ok = false;
        uint32_t g = 0, chr = 0, lo = 0, hi = 0, pos = 0;
        bool positive = false;
        if (j < o1) {
            if (STATS) S.cOcc++;
            const uint4 rec = T.occRec[j];
            g = rec.x; chr = rec.y; pos = rec.z;
            lo = T.chrStart[chr]; hi = T.chrStart[chr + 1];
            positive = ((int32_t)rec.w == vid);
            ok = !lcb_it_used(T, g, positive, lo) && (int32_t)lcb_it_char(T, g, positive) == ch;
        }
        const unsigned long long m = __ballot(ok);
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (S.nInst + cnt > S.instCap) { S.status = LCB_ST_INST_OVF; return; }
        if (ok) {
            const uint32_t i = S.nInst + (uint32_t)__popcll(m & ((1ull << S.lane) - 1));
            S.iFrontG[i] = g; S.iBackG[i] = g; S.iFrontPos[i] = pos; S.iBackPos[i] = pos;
            S.iFrontDist[i] = 0; S.iBackDist[i] = 0; S.iChr[i] = chr; S.iLo[i] = lo; S.iHi[i] = hi;
            S.iFlags[i] = positive ? LCB_FLAG_POS : 0u;
            if (i >= S.nFp) { S.fpLo[i] = g; S.fpHi[i] = g; }        // the replay re-creates instance i at the same occurrence
            S.ordKey[S.cur * S.instCap + i] = g;   // occurrences ascend in g, so pool order == key order here
            S.ordIdx[S.cur * S.instCap + i] = i;
        }
        S.nInst += cnt;
        if (S.nInst > S.nFp) S.nFp = S.nInst;
    }
    LCB_WAVE_SYNC();
}

// ---- the vote: MostPopularVertex (blocksfinder.h:708-768) --------------------------------------
// Returns the chosen vertex (0 = none) and the pool index of the origin instance.
// The voter walks of one vote. With helper wavefronts (nWaves > 1) wave w takes list entries e == w (mod nWaves); all
// waves accumulate into the shared LDS vote table with atomics, so the split needs no merging.
template <bool STATS>
__device__ inline bool lcb_vote_walk(LcbState& S, bool forward, bool tryUsed, bool useGood, uint32_t nList, int32_t flank,
                                     uint32_t waveId, uint32_t nWaves)
{
    const LcbTables& T = S.T;
    const uint32_t touchedCap = S.voteCap - (S.voteCap >> 2);
    const uint32_t vmask = S.voteCap - 1;
    bool ovf = false;
    // One voter = one instance whose end vertex is the path end (blocksfinder.h:716-717); its look-ahead window is
    // walked 64 steps per pass, lanes = steps. The three table reads of a pass (pos, id, used word) are independent and
    // issued together; the first pass of the NEXT voter is issued before the current one is consumed, so its latency
    // hides behind the LDS work of this one.
    struct Voter { uint32_t e, i, g0, pos0, lo, hi, weight; int32_t dir; bool positive; };
    struct Walk { uint32_t g, pos; int32_t id; bool valid, used; };
    auto nextVoter = [&](uint32_t e, Voter& v) -> bool {
        for (; e < nList; e += nWaves) {
            const uint32_t i = useGood ? S.good[e] : e;
            // inst->Back().GetVertexId() == path end vertex  <=>  equal path distances (strictly monotone)
            if ((forward ? S.iBackDist[i] : S.iFrontDist[i]) != flank) continue;
            v.e = e; v.i = i;
            v.positive = (S.iFlags[i] & LCB_FLAG_POS) != 0;
            v.g0 = forward ? S.iBackG[i] : S.iFrontG[i];
            v.pos0 = forward ? S.iBackPos[i] : S.iFrontPos[i];
            v.lo = S.iLo[i]; v.hi = S.iHi[i];
            v.weight = lcb_absdiff(S.iFrontPos[i], S.iBackPos[i]) + 1u;            // blocksfinder.h:719
            v.dir = (forward == v.positive) ? 1 : -1;
            return true;
        }
        return false;
    };
    auto issue = [&](const Voter& v, uint32_t c) -> Walk {
        Walk w;
        const uint32_t d = c * 64 + S.lane + 1;
        const int64_t gg = (int64_t)v.g0 + (int64_t)v.dir * (int64_t)d;
        w.valid = gg >= (int64_t)v.lo && gg < (int64_t)v.hi;                       // it.Valid()
        w.g = (uint32_t)gg; w.pos = 0; w.id = 0; w.used = false;
        if (w.valid) {
            w.pos = T.posPos[w.g];
            w.id = T.posId[w.g];
            w.used = lcb_it_used(T, w.g, v.positive, v.lo);
        }
        return w;
    };
    Voter cur, nxt;
    Walk wcur, wnxt;
    bool have = nextVoter(waveId, cur);
    if (have) wcur = issue(cur, 0);
    while (have) {
        const bool haveNext = nextVoter(cur.e + nWaves, nxt);
        if (haveNext) wnxt = issue(nxt, 0);
        for (uint32_t c = 0;; c++) {
            const Walk w = c == 0 ? wcur : issue(cur, c);
            const uint32_t d = c * 64 + S.lane + 1;
            const bool cond = w.valid && (d < (uint32_t)S.P.depth || lcb_absdiff(w.pos, cur.pos0) <= (uint32_t)S.P.maxBranch);
            const int32_t vid = cur.positive ? w.id : -w.id;
            const bool stop = cond && ((!tryUsed && w.used) || lcb_path_contains(S, vid));
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
                uint32_t probe = 0;
                for (; probe < S.voteCap; probe++) {
                    const int32_t old = atomicCAS(&S.vKey[h], LCB_EMPTY_KEY, vid);
                    if (old == LCB_EMPTY_KEY) {
                        const uint32_t t = atomicAdd(S.vNTouched, 1u);
                        if (t < touchedCap) S.vTouched[t] = h; else ovf = true;
                        break;
                    }
                    if (old == vid) break;
                    h = (h + 1) & vmask;
                }
                if (probe == S.voteCap) ovf = true;
                else {
                    atomicAdd(&S.vCount[h], cur.weight);
                    atomicMax(&S.vLast[h], ((unsigned long long)cur.e << 32) | d);
                }
            }
            if (first < 64) {
                if (!tryUsed && S.lane == 0) {
                    // steps 1 .. c*64+first-1 read used == 0 (one step of slack keeps the - strand's bit g-1 inside)
                    const int64_t ext = (int64_t)cur.g0 + (int64_t)cur.dir * (int64_t)(c * 64 + first);
                    const uint32_t ge = ext < (int64_t)cur.lo ? cur.lo : (ext >= (int64_t)cur.hi ? cur.hi - 1 : (uint32_t)ext);
                    if (ge < S.fpLo[cur.i]) S.fpLo[cur.i] = ge;
                    if (ge > S.fpHi[cur.i]) S.fpHi[cur.i] = ge;
                }
                break;
            }
        }
        have = haveNext; cur = nxt; wcur = wnxt;
    }
    return __ballot(ovf) != 0;
}

// Mailbox words through which wave 0 hands a vote to the helper wavefronts of its workgroup.
enum { LCB_MAIL_CMD = 0, LCB_MAIL_FLAGS, LCB_MAIL_NLIST, LCB_MAIL_FLANK, LCB_MAIL_OVF, LCB_MAIL_WORDS = 8 };
enum { LCB_CMD_VOTE = 1, LCB_CMD_EXIT = 2 };

template <bool STATS, bool PROF>
__device__ inline int32_t lcb_vote(LcbState& S, bool forward, bool tryUsed, uint32_t& originInst)
{
    const bool useGood = S.nGood >= 2;                             // blocksfinder.h:713
    const uint32_t nList = useGood ? S.nGood : S.nInst;
    const uint32_t touchedCap = S.voteCap - (S.voteCap >> 2);
    const int32_t flank = forward ? S.rightFlank : S.leftFlank;
    if (STATS && S.lane == 0) S.cVote++;
    if (PROF) S.pfVote++;
    bool ovfAny;
    if (S.nWaves > 1 && nList >= 4) {
        // wake the helper wavefronts: they walk their share of the voters while this wave walks its own
        if (S.lane == 0) {
            S.mail[LCB_MAIL_FLAGS] = (forward ? 1u : 0u) | (tryUsed ? 2u : 0u) | (useGood ? 4u : 0u);
            S.mail[LCB_MAIL_NLIST] = nList; S.mail[LCB_MAIL_FLANK] = (uint32_t)flank; S.mail[LCB_MAIL_OVF] = 0;
            S.mail[LCB_MAIL_CMD] = LCB_CMD_VOTE;
        }
        __syncthreads();
        const bool mine = lcb_vote_walk<STATS>(S, forward, tryUsed, useGood, nList, flank, 0, S.nWaves);
        __syncthreads();
        ovfAny = mine || S.mail[LCB_MAIL_OVF] != 0;
        if (STATS && S.lane == 0) { S.cWalk += *S.mailWalk; }
        LCB_WAVE_SYNC();
        if (STATS && S.lane == 0) *S.mailWalk = 0;
    } else {
        ovfAny = lcb_vote_walk<STATS>(S, forward, tryUsed, useGood, nList, flank, 0, 1);
    }
    LCB_WAVE_SYNC();
    uint32_t nTouched = *S.vNTouched;
    if (ovfAny || nTouched > touchedCap) { S.status = LCB_ST_VOTE_OVF; if (nTouched > touchedCap) nTouched = touchedCap; }
    // order-free arg-max: max count; ties -> smallest origin (strand, g) of the last contributing
    // instance; ties -> earliest step.  key = (positive << 63) | (g0 << 31 >> ...) packed below.
    uint32_t bestCount = 0, bestSlot = 0xFFFFFFFFu;
    unsigned long long bestKey = ~0ull;
    for (uint32_t t = S.lane; t < nTouched; t += 64) {
        const uint32_t h = S.vTouched[t];
        const uint32_t cnt = S.vCount[h];
        const unsigned long long last = S.vLast[h];
        const uint32_t e = (uint32_t)(last >> 32), d = (uint32_t)last;
        const uint32_t i = useGood ? S.good[e] : e;
        const uint32_t g0 = forward ? S.iBackG[i] : S.iFrontG[i];
        // JunctionSequentialIterator::operator< (junctionstorage.h:349-362): negative strand first, then chr, idx
        const unsigned long long key = ((unsigned long long)(S.iFlags[i] & LCB_FLAG_POS) << 62) |
                                       ((unsigned long long)g0 << 30) | (unsigned long long)(d & 0x3FFFFFFFu);
        if (cnt > bestCount || (cnt == bestCount && key < bestKey)) { bestCount = cnt; bestKey = key; bestSlot = h; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t oc = (uint32_t)__shfl_xor((int)bestCount, o);
        const uint32_t os = (uint32_t)__shfl_xor((int)bestSlot, o);
        const uint32_t klo = (uint32_t)__shfl_xor((int)(uint32_t)bestKey, o);
        const uint32_t khi = (uint32_t)__shfl_xor((int)(uint32_t)(bestKey >> 32), o);
        const unsigned long long ok = ((unsigned long long)khi << 32) | klo;
        if (oc > bestCount || (oc == bestCount && ok < bestKey)) { bestCount = oc; bestKey = ok; bestSlot = os; }
    }
    int32_t bestVid = 0;
    originInst = 0;
    if (bestSlot != 0xFFFFFFFFu && bestCount > 0) {
        bestVid = S.vKey[bestSlot];
        const uint32_t e = (uint32_t)(S.vLast[bestSlot] >> 32);
        originInst = useGood ? S.good[e] : e;
    }
    LCB_WAVE_SYNC();
    for (uint32_t t = S.lane; t < nTouched; t += 64) {              // blocksfinder.h:761-766
        const uint32_t h = S.vTouched[t];
        S.vKey[h] = LCB_EMPTY_KEY; S.vCount[h] = 0; S.vLast[h] = 0;
    }
    if (S.lane == 0) *S.vNTouched = 0;
    LCB_WAVE_SYNC();
    return bestVid;
}

// The table reads one push needs about the walked edge, loadable one step ahead of the push itself.
struct LcbStep { int32_t idIt, idN; uint32_t posIt, posN; int32_t ech; };

template <bool BACK>
__device__ __forceinline__ LcbStep lcb_load_step(const LcbTables& T, uint32_t gIt, bool itPositive)
{
    const uint32_t gN = BACK ? (itPositive ? gIt + 1 : gIt - 1) : (itPositive ? gIt - 1 : gIt + 1);
    LcbStep st;
    st.idIt = T.posId[gIt]; st.idN = T.posId[gN]; st.posIt = T.posPos[gIt]; st.posN = T.posPos[gN];
    // e.GetChar(): outgoing -> char at gIt, ingoing -> char at the previous position gN (junctionstorage.h:191-227)
    st.ech = (int32_t)lcb_it_char(T, BACK ? gIt : gN, itPositive);
    return st;
}

// One occurrence of the pushed vertex: its record plus the three `used` words around it, so that IsUsed and (almost
// always) the Compatible gap test need no further global loads.
struct LcbOcc { uint4 rec; uint32_t lo, wbase, uw0, uw1, uw2; };

__device__ __forceinline__ LcbOcc lcb_load_occ(const LcbTables& T, uint32_t j, bool active)
{
    LcbOcc o;
    o.rec = uint4{0u, 0u, 0u, 0u}; o.lo = 0; o.wbase = 0; o.uw0 = o.uw1 = o.uw2 = 0;
    if (active) {
        o.rec = T.occRec[j];
        o.lo = T.chrStart[o.rec.y];
        const uint32_t wi = o.rec.x >> 5;
        o.wbase = wi ? wi - 1 : 0;
        o.uw0 = T.used[o.wbase]; o.uw1 = T.used[o.wbase + 1]; o.uw2 = T.used[o.wbase + 2];
    }
    return o;
}

__device__ __forceinline__ bool lcb_occ_bit(const LcbOcc& o, uint32_t g)
{
    const uint32_t w = (g >> 5) - o.wbase;                // 0..2 for g and g-1
    const uint32_t word = w == 0 ? o.uw0 : (w == 1 ? o.uw1 : o.uw2);
    return (word >> (g & 31)) & 1u;
}

// lcb_range_any_used over [a, b), served from the cached words when the range lies inside them.
__device__ inline bool lcb_range_any_used_c(const LcbTables& T, const LcbOcc& o, uint32_t a, uint32_t b)
{
    if (a >= b) return false;
    const uint32_t wa = a >> 5, wb = (b - 1) >> 5;
    if (wa < o.wbase || wb > o.wbase + 2) return lcb_range_any_used(T.used, a, b);
    bool any = false;
    for (uint32_t w = wa; w <= wb; w++) {
        uint32_t word = (w - o.wbase) == 0 ? o.uw0 : ((w - o.wbase) == 1 ? o.uw1 : o.uw2);
        if (w == wa) word &= 0xFFFFFFFFu << (a & 31);
        if (w == wb) word &= 0xFFFFFFFFu >> (31 - ((b - 1) & 31));
        any = any || word != 0;
    }
    return any;
}

// ---- a push: PointPushBack / PointPushFront with their workers (path.h:430-602) ------------------
// Per-occurrence outcomes.
#define LCB_ACT_NONE 0u
#define LCB_ACT_SKIP 1u      // Within() the upper-bound instance -> `continue`
#define LCB_ACT_EXT_P 2u     // extend the predecessor instance
#define LCB_ACT_EXT_X 3u     // extend the upper-bound instance
#define LCB_ACT_INSERT 4u    // new single-point instance

// BACK=true:  PointPushBack(e), e = OutgoingEdge of iterator (gIt, itPositive): vertex = end vertex.
// BACK=false: PointPushFront(e), e = IngoingEdge of iterator (gIt, itPositive): vertex = start vertex.
// Returns false iff the vertex is already in the path (path.h:571-574,589-592).
template <bool BACK, bool STATS, bool PROF>
__device__ inline bool lcb_push(LcbState& S, uint32_t gIt, bool itPositive, bool record, const LcbStep& st)
{
    const LcbTables& T = S.T;
    const int32_t vertex = itPositive ? st.idN : -st.idN;            // pushed vertex
    const int32_t otherVertex = itPositive ? st.idIt : -st.idIt;     // e.GetEndVertex() for a front push
    // the CSR lookup is issued before the path-set probe so that the two global round trips overlap
    const uint32_t av = (uint32_t)(vertex < 0 ? -vertex : vertex);
    const uint32_t o0 = T.occStart[av], o1 = T.occStart[av + 1];
    if (PROF ? lcb_path_contains_p(S, vertex) : lcb_path_contains(S, vertex)) return false;
    const uint32_t length = lcb_absdiff(st.posN, st.posIt);
    const int32_t ech = st.ech;
    const int64_t dist64 = BACK ? (int64_t)S.rightFlank + length : (int64_t)S.leftFlank - (int64_t)length;
    if (dist64 > INT32_MAX || dist64 < -(int64_t)INT32_MAX) { S.status = LCB_ST_DIST_OVF; return false; }
    const int32_t distance = (int32_t)dist64;
    // first chunk of occurrences: in flight while the vertex is published in the path set
    LcbOcc occ = lcb_load_occ(T, o0 + S.lane, o0 + S.lane < o1);
    lcb_path_insert(S, vertex);
    if (S.status) return false;

    const int64_t B = S.P.maxBranch;
    for (uint32_t base = o0; base < o1; base += 64) {
        const uint32_t j = base + S.lane;
        const bool active = j < o1;
        const uint32_t n = S.nInst;
        const uint32_t* oKey = S.ordKey + S.cur * S.instCap;
        const uint32_t* oIdx = S.ordIdx + S.cur * S.instCap;
        uint32_t g = 0, chr = 0, lo = 0, pos = 0, u = 0, cand = 0, act = LCB_ACT_NONE;
        uint32_t stCall = 0, stStep = 0;                             // stats: Compatible calls / walk steps of this occurrence
        bool positive = false, usedS = false, usesP = false;
        if (base != o0) occ = lcb_load_occ(T, j, active);
        if (active) {
            if (STATS) S.cOcc++;
            g = occ.rec.x; chr = occ.rec.y; pos = occ.rec.z;
            lo = occ.lo;
            positive = ((int32_t)occ.rec.w == vertex);               // JunctionIterator::IsPositiveStrand
            usedS = positive ? lcb_occ_bit(occ, g) : (g > lo ? lcb_occ_bit(occ, g - 1) : false);   // JunctionSequentialIterator::IsUsed
            // instanceSet.upper_bound(Instance(seqIt, 0)): first key > g
            uint32_t a = 0, b = n;
            while (a < b) { const uint32_t mid = (a + b) >> 1; if (g < oKey[mid]) b = mid; else a = mid + 1; }
            u = a;
            const bool hasX = u < n && S.iChr[oIdx[u]] == chr;
            const bool hasP = u > 0 && S.iChr[oIdx[u - 1]] == chr;
            bool skip = false;
            if (hasX) {                                              // Instance::Within (path.h:170-175)
                const uint32_t x = oIdx[u];
                const uint32_t f = S.iFrontG[x], bk = S.iBackG[x];
                skip = g >= (f < bk ? f : bk) && g <= (f < bk ? bk : f);
            }
            if (skip) act = LCB_ACT_SKIP;
            else {
                usesP = BACK ? positive : !positive;
                const bool has = usesP ? hasP : hasX;
                bool compat = false;
                if (has) {
                    cand = usesP ? oIdx[u - 1] : oIdx[u];
                    if (STATS) stCall = 1;
                    const bool cpos = (S.iFlags[cand] & LCB_FLAG_POS) != 0;
                    if (cpos == positive) {                          // path.h:382-385
                        const uint32_t cg = BACK ? S.iBackG[cand] : S.iFrontG[cand];
                        const uint32_t cp = BACK ? S.iBackPos[cand] : S.iFrontPos[cand];
                        const int32_t cd = BACK ? S.iBackDist[cand] : S.iFrontDist[cand];
                        // Compatible(start, end, e): BACK: start = cand.Back(), end = seqIt; FRONT: start = seqIt, end = cand.Front()
                        const int64_t startPos = BACK ? cp : pos, endPos = BACK ? pos : cp;
                        const int64_t realDiff = positive ? endPos - startPos : startPos - endPos;
                        const int64_t ancestralDiff = BACK ? (int64_t)distance - cd : (int64_t)cd - distance;
                        const uint32_t ga = g < cg ? g : cg, gb = g < cg ? cg : g;
                        if (STATS) stStep = lcb_range_walk_steps(T.used, ga, gb, positive);
                        bool okDist = realDiff >= 0;
                        if (okDist && (realDiff > B || ancestralDiff > B)) {
                            // only an exact next-edge continuation is accepted (path.h:407-411,420-424)
                            const uint32_t gs = BACK ? cg : g, ge = BACK ? g : cg;      // start, end
                            const bool adjacent = positive ? (ge == gs + 1) : (gs == ge + 1);
                            okDist = adjacent && (int32_t)lcb_it_char(T, gs, positive) == ech;
                            if (okDist && !BACK) {
                                const int32_t idE = T.posId[ge];
                                okDist = (positive ? idE : -idE) == otherVertex;       // start1.GetVertexId() == e.GetEndVertex()
                            }
                        }
                        compat = okDist && !lcb_range_any_used_c(T, occ, ga, gb);
                        // `inst->Back().GetVertexId() != vertex` (path.h:541; :472 for the front): a candidate that already ends at
                        // the pushed vertex — it was inserted or extended by an occurrence of an EARLIER 64-lane chunk of this very
                        // push (within a chunk the prefix rule below does the same) — is not extended again: else branch.
                        // Equal path distance <=> same vertex (distances are strictly monotone along the path).
                        if (cd == distance) compat = false;
                    }
                }
                if (compat) {
                    const bool fin = (S.iFlags[cand] & (BACK ? LCB_FLAG_BACKFIN : LCB_FLAG_FRONTFIN)) != 0;
                    act = fin ? LCB_ACT_NONE : (usesP ? LCB_ACT_EXT_P : LCB_ACT_EXT_X);
                } else act = usedS ? LCB_ACT_NONE : LCB_ACT_INSERT;
            }
        }
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
                stStep = (posT != 0) == positive ? lcb_range_walk_steps(T.used, gT, g, positive) : 0u;
            }
            S.cCompatCall += stCall; S.cCompatStep += stStep;
        }
        // apply
        const bool ins = active && act == LCB_ACT_INSERT;
        const bool ext = active && (act == LCB_ACT_EXT_P || act == LCB_ACT_EXT_X);
        bool becameGood = false;
        if (ext) {
            const int64_t before = lcb_real_length(S, cand);
            if (BACK) {                                              // Instance::ChangeBack (path.h:124-133)
                S.iBackG[cand] = g; S.iBackPos[cand] = pos; S.iBackDist[cand] = distance;
                if (positive) S.ordKey[S.cur * S.instCap + u - 1] = g;   // compareIdx_ follows the + strand back
                if (usedS) S.iFlags[cand] |= LCB_FLAG_BACKFIN;
            } else {                                                 // Instance::ChangeFront (path.h:113-122)
                S.iFrontG[cand] = g; S.iFrontPos[cand] = pos; S.iFrontDist[cand] = distance;
                if (!positive) S.ordKey[S.cur * S.instCap + u - 1] = g;  // compareIdx_ follows the - strand front
                if (usedS) S.iFlags[cand] |= LCB_FLAG_FRONTFIN;
            }
            if (g < S.fpLo[cand]) S.fpLo[cand] = g;
            if (g > S.fpHi[cand]) S.fpHi[cand] = g;
            becameGood = before < (int64_t)S.P.minBlock && lcb_real_length(S, cand) >= (int64_t)S.P.minBlock;
        }
        const unsigned long long insM = __ballot(ins);
        const unsigned long long goodM = __ballot(becameGood);
        const uint32_t m = (uint32_t)__popcll(insM);
        if (S.nInst + m > S.instCap) { S.status = LCB_ST_INST_OVF; return true; }
        if (becameGood) S.good[S.nGood + (uint32_t)__popcll(goodM & ((1ull << S.lane) - 1))] = cand;
        S.nGood += (uint32_t)__popcll(goodM);
        if (ins) {
            const uint32_t r = (uint32_t)__popcll(insM & ((1ull << S.lane) - 1));
            const uint32_t i = S.nInst + r;
            S.iFrontG[i] = g; S.iBackG[i] = g; S.iFrontPos[i] = pos; S.iBackPos[i] = pos;
            S.iFrontDist[i] = distance; S.iBackDist[i] = distance; S.iChr[i] = chr; S.iLo[i] = lo;
            S.iHi[i] = T.chrStart[chr + 1];
            S.iFlags[i] = positive ? LCB_FLAG_POS : 0u;
            if (i >= S.nFp) { S.fpLo[i] = g; S.fpHi[i] = g; }
            S.scr[r] = u; S.scr[64 + r] = g; S.scr[128 + r] = i;
        }
        S.nInst += m;
        if (S.nInst > S.nFp) S.nFp = S.nInst;
        LCB_WAVE_SYNC();
        if (m) lcb_order_merge(S, m);
    }
    if (BACK) {
        if (record) {
            if (S.nRight >= S.bodyCap) { S.status = LCB_ST_PATH_OVF; return true; }
            if (S.lane == 0) S.body[S.nRight] = ((unsigned long long)(itPositive ? 1u : 0u) << 32) | gIt;
        }
        S.nRight++;
        S.rightFlank = distance;
    } else {
        S.nLeft++;
        S.leftFlank = distance;
    }
    if (STATS && S.lane == 0) S.cPush++;
    if (PROF) { S.pfPush++; if (S.nInst > S.pfMaxInst) S.pfMaxInst = S.nInst; }
    return true;
}

// Path::Score (path.h:604-628)
__device__ inline int64_t lcb_score(const LcbState& S)
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
    sum = lcb_wave_sum(sum);
    return anyBad ? -(int64_t)INT32_MAX : sum;
}

// bestInstance <- goodInstance_ (blocksfinder.h:820-824,883-887)
__device__ inline void lcb_snapshot(LcbState& S)
{
    if (S.nGood > S.bestCap) { S.status = LCB_ST_BEST_OVF; return; }
    for (uint32_t e = S.lane; e < S.nGood; e += 64) {
        const uint32_t i = S.good[e];
        uint4 r;
        r.x = S.iChr[i]; r.y = S.iFrontG[i] - S.iLo[i]; r.z = S.iBackG[i] - S.iLo[i]; r.w = (S.iFlags[i] & LCB_FLAG_POS);
        S.best[e] = r;
    }
    S.nBest = S.nGood;
}

// Forward pushes only append instances, good-list entries and path vertices and only change the Back fields, the flags and
// the ordered index of existing instances; the best-scoring point only moves forward. A checkpoint taken at a best point
// therefore stays a valid restart for the replay whatever follows.
#define LCB_CK_EVERY 32u
__device__ inline void lcb_checkpoint(LcbState& S)
{
    const uint32_t n = S.nInst, cap = S.bestCap;
    if (n > cap) return;
    uint32_t* c = S.ck;
    const uint32_t* oKey = S.ordKey + S.cur * S.instCap;
    const uint32_t* oIdx = S.ordIdx + S.cur * S.instCap;
    for (uint32_t i = S.lane; i < n; i += 64) {
        c[i] = S.iBackG[i]; c[cap + i] = S.iBackPos[i]; c[2 * cap + i] = (uint32_t)S.iBackDist[i]; c[3 * cap + i] = S.iFlags[i];
        c[4 * cap + i] = oKey[i]; c[5 * cap + i] = oIdx[i];
    }
    S.ckN = S.nRight; S.ckInst = n; S.ckGood = S.nGood; S.ckPath = S.nPath; S.ckFlank = S.rightFlank;
}

// Back to the checkpoint: the state after ckN forward pushes (Front fields, chromosome data and the right-body list are
// untouched by forward pushes; the Bloom filter and the footprints keep their supersets).
__device__ inline void lcb_restore_checkpoint(LcbState& S)
{
    LCB_WAVE_SYNC();
    for (uint32_t i = S.ckPath + S.lane; i < S.nPath; i += 64) S.pKeys[S.pSlots[i]] = LCB_EMPTY_KEY;
    const uint32_t n = S.ckInst, cap = S.bestCap;
    const uint32_t* c = S.ck;
    uint32_t* oKey = S.ordKey + S.cur * S.instCap;
    uint32_t* oIdx = S.ordIdx + S.cur * S.instCap;
    for (uint32_t i = S.lane; i < n; i += 64) {
        S.iBackG[i] = c[i]; S.iBackPos[i] = c[cap + i]; S.iBackDist[i] = (int32_t)c[2 * cap + i]; S.iFlags[i] = c[3 * cap + i];
        oKey[i] = c[4 * cap + i]; oIdx[i] = c[5 * cap + i];
    }
    S.nPath = S.ckPath; S.nInst = n; S.nGood = S.ckGood; S.nRight = S.ckN; S.nLeft = 0;
    S.rightFlank = S.ckFlank; S.leftFlank = 0;
    LCB_WAVE_SYNC();
}

// ExtendPathForward / ExtendPathBackward (blocksfinder.h:770-895)
template <bool FORWARD, bool STATS, bool PROF>
__device__ inline bool lcb_extend(LcbState& S, uint32_t& bestRightSize, int64_t& bestScore, int64_t& nowScore)
{
    const LcbTables& T = S.T;
    uint32_t oi = 0;
    LCB_MARK(S, 4, S.nRight); LCB_MARK(S, 5, S.nLeft); LCB_MARK(S, 6, 1);
    const uint64_t tv0 = PROF ? wall_clock64() : 0;
    int32_t next = lcb_vote<STATS, PROF>(S, FORWARD, false, oi);
    LCB_MARK(S, 6, 2); LCB_MARK(S, 7, (uint32_t)next);
    if (S.status) return false;
    if (FORWARD && next == 0) {                                      // blocksfinder.h:782-785 (forward only, Q2)
        next = lcb_vote<STATS, PROF>(S, true, true, oi);
        if (S.status) return false;
    }
    if (PROF) S.pfTVote += wall_clock64() - tv0;
    bool success = false;
    if (next != 0) {
        const bool positive = (S.iFlags[oi] & LCB_FLAG_POS) != 0;
        uint32_t g = FORWARD ? S.iBackG[oi] : S.iFrontG[oi];
        const int dir = (FORWARD == positive) ? 1 : -1;
        const uint32_t lo = S.iLo[oi], hi = S.iHi[oi];
        // the walk from the origin to the chosen vertex reads consecutive positions: keep the position after the next
        // one in flight, so a push never waits for its own edge data
        struct At { int32_t id; uint32_t pos; int32_t ch; };
        auto loadAt = [&](int64_t q) -> At {
            At a; a.id = 0; a.pos = 0; a.ch = 0;
            if (q >= (int64_t)lo && q < (int64_t)hi) { a.id = T.posId[q]; a.pos = T.posPos[q]; a.ch = (int32_t)lcb_it_char(T, (uint32_t)q, positive); }
            return a;
        };
        At cur = loadAt(g), nxt = loadAt((int64_t)g + dir);
        for (;;) {
            if ((positive ? cur.id : -cur.id) == next) break;
            const At ahead = loadAt((int64_t)g + 2 * dir);
            LcbStep st;
            st.idIt = cur.id; st.idN = nxt.id; st.posIt = cur.pos; st.posN = nxt.pos; st.ech = FORWARD ? cur.ch : nxt.ch;
            LCB_MARK(S, 6, 3); LCB_MARK(S, 8, g);
            const uint64_t tp0 = PROF ? wall_clock64() : 0;
            success = lcb_push<FORWARD, STATS, PROF>(S, g, positive, true, st);
            const uint64_t tp1 = PROF ? wall_clock64() : 0;
            if (PROF) S.pfTPush += tp1 - tp0;
            LCB_MARK(S, 6, 4);
            if (S.status) return false;
            if (success) {
                nowScore = lcb_score(S);
                if (nowScore > bestScore) {
                    bestScore = nowScore;
                    if (FORWARD) bestRightSize = S.nRight + 1;
                    if (nowScore > 0) { lcb_snapshot(S); if (S.status) return false; }
                    if (FORWARD && !STATS && S.nRight >= S.ckN + LCB_CK_EVERY) lcb_checkpoint(S);
                }
                if (PROF) S.pfTScore += wall_clock64() - tp1;
            }
            g = (uint32_t)((int64_t)g + dir);
            cur = nxt; nxt = ahead;
        }
    }
    return success;
}

// ProcessVertex::Process (blocksfinder.h:228-310)
template <bool STATS, bool PROF>
__device__ inline void lcb_process_seed(LcbState& S, int32_t vid, int32_t ch, int64_t& bestScoreOut)
{
    int64_t score = 0, bestScore = 0;
    S.nBest = 0; S.status = LCB_ST_OK; S.ckN = 0;
    LCB_MARK(S, 2, 1);
    lcb_path_init<STATS>(S, vid, ch);
    LCB_MARK(S, 2, 2); LCB_MARK(S, 3, S.nInst);
    uint32_t bestRightSize = 1;
    const int64_t minRun = 2 * (int64_t)S.P.maxBranch;
    if (!S.status) {
        for (;;) {                                                   // blocksfinder.h:255-269
            bool ret = true, positive = false;
            const int64_t prevLength = (int64_t)S.rightFlank - S.leftFlank;
            while ((ret = lcb_extend<true, STATS, PROF>(S, bestRightSize, bestScore, score)) &&
                   ((int64_t)S.rightFlank - S.leftFlank) - prevLength <= minRun)
                positive = positive || (score > 0);
            if (!ret || !positive || S.status) break;
        }
    }
    LCB_MARK(S, 2, 3);
    if (!S.status) {                                                 // replay, blocksfinder.h:271-284
        const uint32_t nEdge = bestRightSize - 1;
        uint32_t from = 0;
        if (!STATS && S.ckN && S.ckN <= nEdge) {
            lcb_restore_checkpoint(S);    // the state after ckN pushes; stats mode replays from Init like the reference (event counts)
            from = S.ckN;
        } else {
            lcb_path_clear(S);            // keeps the body list, resets everything else
            lcb_path_init<STATS>(S, vid, ch);
        }
        for (uint32_t i = from; i < nEdge && !S.status; i++) {
            const unsigned long long b = S.body[i];
            const LcbStep st = lcb_load_step<true>(S.T, (uint32_t)b, (b >> 32) != 0);
            lcb_push<true, STATS, PROF>(S, (uint32_t)b, (b >> 32) != 0, false, st);
        }
    }
    LCB_MARK(S, 2, 4);
    if (!S.status) {
        for (;;) {                                                   // blocksfinder.h:292-306 (stray ';' at :297, Q1)
            bool ret = true;
            const int64_t prevLength = (int64_t)S.rightFlank - S.leftFlank;
            while ((ret = lcb_extend<false, STATS, PROF>(S, bestRightSize, bestScore, score)) &&
                   ((int64_t)S.rightFlank - S.leftFlank) - prevLength <= minRun)
                ;
            const bool positive = score > 0;
            if (!ret || !positive || S.status) break;
        }
    }
    LCB_MARK(S, 2, 5);
    lcb_path_clear(S);                // Path::Clear (blocksfinder.h:308)
    if (S.status == LCB_ST_VOTE_OVF) {
        // the vote table may hold stale keys after an overflow: wipe it
        for (uint32_t h = S.lane; h < S.voteCap; h += 64) { S.vKey[h] = LCB_EMPTY_KEY; S.vCount[h] = 0; S.vLast[h] = 0; }
        if (S.lane == 0) *S.vNTouched = 0;
    }
    LCB_WAVE_SYNC();
    bestScoreOut = bestScore;
}

// ---- the kernel --------------------------------------------------------------------------------
// Per-launch arguments that are only touched between seeds (work queue, result arenas). They are parked in LDS so that
// they do not occupy scalar registers for the whole kernel: the per-seed code already needs the 102-SGPR budget.
struct LcbLaunchArgs {
    const LcbKSeed* seeds;
    LcbSeedOut* out;
    uint4* arena;
    uint2* fpArena;
    unsigned long long arenaCap, fpCap, arenaBase, fpBase;
    unsigned long long* arenaCursor;
    unsigned long long* fpCursor;
    uint32_t* cursor;
    uint32_t cursorBase, nSeeds;
    const uint32_t* usedView;      // `used` view of the seed wave 0 is working on (read by the helpers at each vote)
};

// NW = wavefronts per workgroup: wave 0 runs the per-seed algorithm, waves 1..NW-1 are vote helpers.
template <int MODE, bool STATS, int NW, bool PROF>
__device__ inline void lcb_process_body(const LcbTables& T, const LcbKParams& P, const LcbKSeed* seeds, uint32_t nSeeds,
                                        const LcbWork& W, LcbSeedOut* out, uint4* arena, unsigned long long arenaCap,
                                        uint2* fpArena, unsigned long long fpCap)
{
    constexpr bool BIG = MODE == 2;
    constexpr uint32_t IC = BIG ? 1u : (MODE == 1 ? LCB_IC_MEDIUM : LCB_IC_SMALL);
    constexpr uint32_t VC = BIG ? 1u : (MODE == 1 ? LCB_VC_MEDIUM : LCB_VC_SMALL);
    __shared__ uint32_t sInst[10 * IC];
    __shared__ uint32_t sOrdKey[2 * IC];
    __shared__ uint32_t sOrdIdx[2 * IC];
    __shared__ uint32_t sGood[IC];
    __shared__ uint32_t sFp[2 * IC];
    __shared__ int32_t sVKey[VC];
    __shared__ uint32_t sVCount[VC];
    __shared__ unsigned long long sVLast[VC];
    __shared__ uint32_t sVTouched[VC];
    __shared__ uint32_t sBloom[LCB_BLOOM_WORDS];
    __shared__ uint32_t sScr[4 * 64];
    __shared__ uint32_t sMisc[4];
    __shared__ uint32_t sMail[LCB_MAIL_WORDS];
    __shared__ unsigned long long sMailWalk[1];
    __shared__ LcbLaunchArgs sArgs;
    if (threadIdx.x == 0) {
        sArgs.seeds = seeds; sArgs.out = out; sArgs.arena = arena; sArgs.fpArena = fpArena; sArgs.arenaCap = arenaCap; sArgs.fpCap = fpCap;
        sArgs.arenaBase = W.arenaBase; sArgs.fpBase = W.fpBase; sArgs.arenaCursor = W.arenaCursor; sArgs.fpCursor = W.fpCursor;
        sArgs.cursor = W.cursor; sArgs.cursorBase = W.cursorBase; sArgs.nSeeds = nSeeds;
    }

    LcbState S;
    S.T = T; S.P = P;
    S.lane = threadIdx.x & 63u;
    uint8_t* slot = W.base + (uint64_t)blockIdx.x * W.slotBytes;
    const LcbSlotLayout L = lcb_slot_layout(W.pathCap, W.bodyCap, W.bestCap, BIG ? W.instCap : 0, BIG ? W.voteCap : 0);
    S.pKeys = (int32_t*)(slot + L.pKeys);
    S.pSlots = (uint32_t*)(slot + L.pSlots);
    S.pathCap = W.pathCap; S.pathShift = 32u - (uint32_t)__ffs((int)W.pathCap) + 1u;
    S.body = (unsigned long long*)(slot + L.body); S.bodyCap = W.bodyCap;
    S.best = (uint4*)(slot + L.best); S.bestCap = W.bestCap;
    S.ck = (uint32_t*)(slot + L.ck); S.ckN = S.ckInst = S.ckGood = S.ckPath = 0; S.ckFlank = 0;
    uint32_t* instBase;
    if (BIG) {
        instBase = (uint32_t*)(slot + L.inst); S.instCap = W.instCap;
        S.ordKey = (uint32_t*)(slot + L.ordKey);
        S.ordIdx = (uint32_t*)(slot + L.ordIdx);
        S.good = (uint32_t*)(slot + L.good);
        S.fpLo = (uint32_t*)(slot + L.fp); S.fpHi = S.fpLo + W.instCap;
        S.vKey = (int32_t*)(slot + L.vKey); S.vCount = (uint32_t*)(slot + L.vCount);
        S.vLast = (unsigned long long*)(slot + L.vLast); S.vTouched = (uint32_t*)(slot + L.vTouched);
        S.voteCap = W.voteCap;
    } else {
        instBase = sInst; S.instCap = IC;
        S.ordKey = sOrdKey;
        S.ordIdx = sOrdIdx;
        S.good = sGood;
        S.fpLo = sFp; S.fpHi = sFp + IC;
        S.vKey = sVKey; S.vCount = sVCount; S.vLast = sVLast; S.vTouched = sVTouched;
        S.voteCap = VC;
        for (uint32_t h = S.lane; h < VC; h += 64) { sVKey[h] = LCB_EMPTY_KEY; sVCount[h] = 0; sVLast[h] = 0; }
    }
    S.bloom = sBloom;
    for (uint32_t h = S.lane; h < LCB_BLOOM_WORDS; h += 64) sBloom[h] = 0;
    S.voteShift = 32u - (uint32_t)__ffs((int)S.voteCap) + 1u;
    S.iFrontG = instBase; S.iBackG = instBase + S.instCap; S.iFrontPos = instBase + 2 * S.instCap;
    S.iBackPos = instBase + 3 * S.instCap; S.iChr = instBase + 4 * S.instCap; S.iLo = instBase + 5 * S.instCap;
    S.iHi = instBase + 6 * S.instCap; S.iFlags = instBase + 7 * S.instCap;
    S.iFrontDist = (int32_t*)(instBase + 8 * S.instCap); S.iBackDist = (int32_t*)(instBase + 9 * S.instCap);
    S.scr = sScr; S.vNTouched = &sMisc[0];
    S.mail = sMail; S.mailWalk = sMailWalk; S.nWaves = NW;
    const uint32_t waveId = threadIdx.x >> 6;
    S.dbg = (LCB_FLIGHT_RECORDER && W.dbg && waveId == 0) ? W.dbg + 16u * blockIdx.x : nullptr;
    LCB_MARK(S, 0, 1);
    if (S.lane == 0) { sMisc[0] = 0; sMail[LCB_MAIL_CMD] = 0; sMail[LCB_MAIL_OVF] = 0; sMailWalk[0] = 0; }
    S.nInst = S.nGood = S.cur = S.nPath = S.nRight = S.nLeft = S.nBest = 0; S.status = 0;
    S.rightFlank = S.leftFlank = 0;
    S.cWalk = S.cOcc = S.cCompatCall = S.cCompatStep = S.cVote = S.cPush = 0;
    S.pfPush = S.pfVote = S.pfMaxProbe = S.pfMaxInst = 0; S.pfTVote = S.pfTPush = S.pfTScore = 0; S.nFp = 0;
    LCB_WAVE_SYNC();
    if (NW > 1) {
        __syncthreads();           // the LDS tables above were initialised (redundantly) by every wave
        if (waveId != 0) {
            // helper wavefront: sleeps at the barrier until wave 0 posts a vote, walks its share of the voters
            for (;;) {
                __syncthreads();
                if (S.mail[LCB_MAIL_CMD] == LCB_CMD_EXIT) return;
                S.T.used = sArgs.usedView;
                const uint32_t flags = S.mail[LCB_MAIL_FLAGS];
                S.cWalk = 0;
                const bool ovf = lcb_vote_walk<STATS>(S, (flags & 1u) != 0, (flags & 2u) != 0, (flags & 4u) != 0, S.mail[LCB_MAIL_NLIST],
                                                      (int32_t)S.mail[LCB_MAIL_FLANK], waveId, NW);
                if (ovf && S.lane == 0) atomicOr(&S.mail[LCB_MAIL_OVF], 1u);
                if (STATS) {
                    const unsigned long long w = (unsigned long long)lcb_wave_sum((int64_t)S.cWalk);
                    if (S.lane == 0 && w) atomicAdd(S.mailWalk, w);
                }
                __syncthreads();
            }
        }
    }

    for (;;) {
        uint32_t s = 0;
        if (S.lane == 0) s = atomicAdd(sArgs.cursor, 1u) - sArgs.cursorBase;   // every workgroup overshoots by exactly one ticket
        s = lcb_bcast(s, 0);
        if (s >= sArgs.nSeeds) break;
        LCB_MARK(S, 1, s + 1);
        S.cWalk = S.cOcc = S.cCompatCall = S.cCompatStep = S.cVote = S.cPush = 0;
        S.nInst = S.nGood = S.cur = S.nRight = S.nLeft = 0; S.rightFlank = S.leftFlank = 0;
        S.nFp = 0;
        int64_t bestScore = 0;
        S.pfPush = S.pfVote = S.pfMaxProbe = S.pfMaxInst = 0; S.pfTVote = S.pfTPush = S.pfTScore = 0;
        const uint64_t tick0 = PROF ? wall_clock64() : 0;
        const LcbKSeed sd = sArgs.seeds[s];
        S.T.used = T.used + (size_t)sd.view * T.usedStride;
        if (NW > 1 && S.lane == 0) sArgs.usedView = S.T.used;     // published to the helpers by the vote's first barrier
        lcb_process_seed<STATS, PROF>(S, sd.vid, sd.ch, bestScore);
        const uint64_t ticks = PROF ? wall_clock64() - tick0 : 0;
        const uint32_t n = S.status ? 0u : S.nBest;
        unsigned long long off = 0;
        if (n) {
            uint32_t olo = 0, ohi = 0;
            if (S.lane == 0) { off = atomicAdd(sArgs.arenaCursor, (unsigned long long)n) - sArgs.arenaBase; olo = (uint32_t)off; ohi = (uint32_t)(off >> 32); }
            olo = lcb_bcast(olo, 0); ohi = lcb_bcast(ohi, 0);
            off = ((unsigned long long)ohi << 32) | olo;
            if (off + n > sArgs.arenaCap) S.status = LCB_ST_ARENA_OVF;
            else { uint4* ar = sArgs.arena; for (uint32_t e = S.lane; e < n; e += 64) ar[off + e] = S.best[e]; }
        }
        // footprint intervals (one per instance ever created), widened by one position on the low side
        // because the - strand reads bit g-1
        uint2* fpa = sArgs.fpArena;
        const uint32_t nfp = (S.status == LCB_ST_OK && fpa) ? S.nFp : 0u;
        unsigned long long fpo = 0;
        if (nfp) {
            uint32_t olo = 0, ohi = 0;
            if (S.lane == 0) { fpo = atomicAdd(sArgs.fpCursor, (unsigned long long)nfp) - sArgs.fpBase; olo = (uint32_t)fpo; ohi = (uint32_t)(fpo >> 32); }
            olo = lcb_bcast(olo, 0); ohi = lcb_bcast(ohi, 0);
            fpo = ((unsigned long long)ohi << 32) | olo;
            if (fpo + nfp > sArgs.fpCap) S.status = LCB_ST_ARENA_OVF;
            else for (uint32_t e = S.lane; e < nfp; e += 64) { uint2 r; r.x = S.fpLo[e] ? S.fpLo[e] - 1 : 0u; r.y = S.fpHi[e]; fpa[fpo + e] = r; }
        }
        uint64_t c[6];
        if (STATS) {
            c[0] = (uint64_t)lcb_wave_sum((int64_t)S.cWalk); c[1] = (uint64_t)lcb_wave_sum((int64_t)S.cOcc);
            c[2] = (uint64_t)lcb_wave_sum((int64_t)S.cCompatCall); c[3] = (uint64_t)lcb_wave_sum((int64_t)S.cCompatStep);
            c[4] = (uint64_t)lcb_wave_sum((int64_t)S.cVote); c[5] = (uint64_t)lcb_wave_sum((int64_t)S.cPush);
        }
        if (S.lane == 0) {
            LcbSeedOut o;
            o.nInst = n;   // kept on ARENA_OVF so the host can track the allocator
            o.status = S.status; o.bestScore = bestScore; o.arenaOff = off;
            o.fpOff = fpo; o.nFp = nfp; o.pad = 0;
            for (int q = 0; q < 8; q++) o.ctr[q] = 0;
            if (!STATS && PROF) { o.ctr[0] = ticks; o.ctr[1] = S.pfPush; o.ctr[2] = S.pfVote; o.ctr[3] = S.pfMaxProbe; o.ctr[4] = S.pfMaxInst; o.ctr[5] = S.pfTVote; o.ctr[6] = S.pfTPush; o.ctr[7] = S.pfTScore; }
            if (STATS) { o.ctr[0] = c[0]; o.ctr[1] = c[1]; o.ctr[2] = c[2]; o.ctr[3] = c[3]; o.ctr[4] = o.nInst; o.ctr[5] = c[4]; o.ctr[6] = c[5]; o.ctr[7] = 1; }
            sArgs.out[s] = o;
        }
        LCB_MARK(S, 2, 6);
    }
    if (NW > 1) {                  // release the helper wavefronts
        if (S.lane == 0) S.mail[LCB_MAIL_CMD] = LCB_CMD_EXIT;
        __syncthreads();
    }
    LCB_MARK(S, 0, 2);
}

#endif
