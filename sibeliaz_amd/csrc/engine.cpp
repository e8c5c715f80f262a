// engine.cpp — the phase loop of BlocksFinder::FindBlocks (blocksfinder.h:334-433,453-530) as a speculative,
// exactly-validated round engine with predicted `used` views.
//
// Reference semantics that must hold bit for bit: (i) every seed of a phase of 256 is processed against the `used`
// bits as of the START of its phase — call that result E; (ii) results are committed strictly in seed order; (iii) an E
// that fails the weak conflict check is re-processed against the live state — call that result F — and F is committed
// without a second check.
//
// One phase per launch starves a GPU, and worse: almost all of the time goes into the LONGEST seed of each dependent
// launch (a block of a few kbp is ~1000 sequential extension steps), so what matters is the number of dependent
// launches. A seed's computation is a deterministic function of the `used` bits it read, and the kernel reports a
// FOOTPRINT that covers every position whose bit it read as 0 (lcb_kernel.h). Hence a result computed against ANY bitmap
// W equals the one the reference computes against the true state T iff
//     (1) W has no bit that T lacks (reads that returned 1 are not in the footprint, so W must not over-predict), and
//     (2) no bit of T \ W lies inside the footprint.
// The engine exploits this twice. A ROUND of many phases is launched against the state at the round start (W = an old
// T: (1) holds because bits only go 0 -> 1). Then, whenever the ordered commit needs a result it does not have — an E
// whose footprint was hit by an earlier commit, or the F of a conflicting seed — it does not launch that one seed: it
// dry-runs the rest of the round with the results it has as PREDICTIONS of what will be committed, and launches every
// seed that will need a new E or F at once, each against a predicted view W = live state + the blocks predicted to be
// committed before its turn (the processor materialises these nested views; lcb_kernel.h reads the view a seed names).
// Neighbouring blocks, which invalidate each other one after the other, are thereby recomputed in ONE launch instead of
// a chain of launches; a result whose prediction did not come true fails (1) or (2) and is simply computed again at the
// next stop, where the first missing result always runs against the live state itself, so the loop makes progress.
// A stop needs ONE result to go on - the first job of its plan, which runs against the live state; everything else the plan
// launches is speculation. With a processor that has side lanes (LcbProcessor::sideBegin, the product's device) the speculation
// runs in the background and the stop waits for its one result only: a launch is as long as its longest seed, and the longest
// job of a plan is usually not the one the commit is waiting for. Background results are taken when the commit reaches their seed,
// if every predicted mark of their view has come true by then; they are validated like any other result.
// With world > 1 every launch — a round's speculative launch and every job launch — is dealt round-robin to the ranks and
// the per-seed results + footprints are all-gathered; every rank then runs the identical dry runs and commit, so all
// ranks hold the same `used` state and block list without further traffic.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "lcb_host.h"

namespace {

// Disjoint, sorted position ranges [lo, hi), behind a coarse bitmap: bit q of `page` is set iff a range touches the positions
// [q << shift, (q + 1) << shift). Almost every question the engine asks ("does a mark of this epoch lie inside this footprint
// interval?") is answered "no" by two or three words of that bitmap without a search in the ranges - the validation loops and the
// dry runs ask it hundreds of millions of times per pass.
thread_local int g_pageShift = 12;          // set per engine run from the number of positions (at most 2^16 pages)
struct RangeSet {
    std::vector<std::pair<uint64_t, uint64_t>> r;
    std::vector<uint64_t> page;
    int shift = g_pageShift;
    // For the mark set of a predicted view: the ranges not yet known to have come true (allPending). A mark that is true stays true,
    // so the list only shrinks - the validity questions about a view are asked again at every stop of the round.
    mutable std::vector<uint32_t> pend;
    mutable bool pendInit = false;
    void clear() { r.clear(); page.clear(); pend.clear(); pendInit = false; }
    // cond(q) for every range q that is not known to be true yet (isTrue(q): it has come true - it is never asked about again); stops at the first false
    template <class T, class F>
    bool allPending(T isTrue, F cond) const
    {
        if (!pendInit) { pend.resize(r.size()); for (size_t i = 0; i < r.size(); i++) pend[i] = (uint32_t)i; pendInit = true; }
        for (size_t k = 0; k < pend.size();) {
            const auto& q = r[pend[k]];
            if (isTrue(q)) { pend[k] = pend.back(); pend.pop_back(); continue; }
            if (!cond(q)) return false;
            k++;
        }
        return true;
    }
    bool empty() const { return r.empty(); }
    bool pageAny(uint64_t lo, uint64_t hi) const      // any range that touches a page of [lo, hi] (inclusive)?
    {
        if (r.empty()) return false;
        const uint64_t p0 = lo >> shift, p1 = hi >> shift, nBits = (uint64_t)page.size() << 6;
        if (p0 >= nBits) return false;
        const uint64_t q1 = p1 < nBits ? p1 : nBits - 1, w0 = p0 >> 6, w1 = q1 >> 6;
        for (uint64_t w = w0; w <= w1; w++) {
            uint64_t m = ~0ull;
            if (w == w0) m &= ~0ull << (p0 & 63);
            if (w == w1) m &= ~0ull >> (63 - (q1 & 63));
            if (page[(size_t)w] & m) return true;
        }
        return false;
    }
    void add(uint64_t lo, uint64_t hi)
    {
        if (hi <= lo) return;
        const uint64_t p1 = (hi - 1) >> shift;
        if ((p1 >> 6) >= page.size()) page.resize((size_t)(p1 >> 6) + 1, 0ull);
        for (uint64_t q = lo >> shift; q <= p1; q++) page[(size_t)(q >> 6)] |= 1ull << (q & 63);
        auto it = std::lower_bound(r.begin(), r.end(), std::make_pair(lo, (uint64_t)0));
        if (it != r.begin() && (it - 1)->second >= lo) --it;
        auto first = it;
        while (it != r.end() && it->first <= hi) { lo = std::min(lo, it->first); hi = std::max(hi, it->second); ++it; }
        it = r.erase(first, it);
        r.insert(it, std::make_pair(lo, hi));
    }
    // any position p of the set with lo <= p <= hi ?
    bool hits(uint64_t lo, uint64_t hi) const
    {
        if (!pageAny(lo, hi)) return false;
        auto it = std::upper_bound(r.begin(), r.end(), std::make_pair(hi, UINT64_MAX));
        if (it == r.begin()) return false;
        --it;
        return it->second > lo;
    }
    // is [lo, hi) entirely inside the set ?
    bool covers(uint64_t lo, uint64_t hi) const
    {
        if (hi <= lo) return true;
        if (!pageAny(lo, lo)) return false;
        auto it = std::upper_bound(r.begin(), r.end(), std::make_pair(lo, UINT64_MAX));
        if (it == r.begin()) return false;
        --it;
        return it->first <= lo && it->second >= hi;
    }
    // any position p of the set with lo <= p <= hi that is NOT in `except` (null = empty) ?
    bool hitsOutside(uint64_t lo, uint64_t hi, const RangeSet* except) const
    {
        if (!pageAny(lo, hi)) return false;
        if (!except || except->empty()) return hits(lo, hi);
        auto it = std::upper_bound(r.begin(), r.end(), std::make_pair(lo, UINT64_MAX));
        if (it != r.begin()) --it;
        for (; it != r.end() && it->first <= hi; ++it) {
            const uint64_t a = std::max(it->first, lo), b = std::min(it->second, hi + 1);
            if (a < b && !except->covers(a, b)) return true;
        }
        return false;
    }
};

struct Results {                       // per-seed results of one process() call, in call order
    std::vector<uint64_t> off, fpOff;
    std::vector<lcb_instance> inst;
    std::vector<lcb_fp> fp;
    std::vector<lcb_counters> ctr;     // countEvents only
};

inline void addCounters(lcb_counters& a, const lcb_counters& b)
{
    a.n_walk += b.n_walk; a.n_occ += b.n_occ; a.n_compat_call += b.n_compat_call; a.n_compat_step += b.n_compat_step;
    a.n_inst_out += b.n_inst_out; a.n_vote += b.n_vote; a.n_push += b.n_push; a.n_process += b.n_process;
}

// A computed result together with the state it was computed against: the live state at launch `epoch` plus the
// predicted marks of `view` (-1 = none).
struct Cand {
    int32_t epoch = 0, view = -1;
    uint32_t checkedTo = 0;            // marks of the epochs before this one are known not to touch the footprint
    bool viewOk = false;               // every predicted mark of the view is known to have come true
    std::vector<lcb_instance> inst;
    std::vector<lcb_fp> fp;
    lcb_counters ctr{};
};

void pack(const Results& r, int64_t n, std::vector<unsigned char>& buf)
{
    const uint64_t nInst = r.off[(size_t)n], nFp = r.fpOff[(size_t)n];
    buf.resize((size_t)n * 8 + nInst * sizeof(lcb_instance) + nFp * sizeof(lcb_fp));
    uint32_t* head = (uint32_t*)buf.data();
    for (int64_t i = 0; i < n; i++) { head[2 * i] = (uint32_t)(r.off[i + 1] - r.off[i]); head[2 * i + 1] = (uint32_t)(r.fpOff[i + 1] - r.fpOff[i]); }
    unsigned char* q = buf.data() + (size_t)n * 8;
    if (nInst) memcpy(q, r.inst.data(), nInst * sizeof(lcb_instance));
    if (nFp) memcpy(q + nInst * sizeof(lcb_instance), r.fp.data(), nFp * sizeof(lcb_fp));
}

inline void instRange(const lcb_graph* g, const lcb_instance& in, uint64_t& lo, uint64_t& hi)
{
    const uint64_t base = g->chrStart[in.chr];
    lo = base + (in.front_idx < in.back_idx ? in.front_idx : in.back_idx);
    hi = base + (in.front_idx < in.back_idx ? in.back_idx : in.front_idx);
}

}  // namespace

void lcb_engine_run(const lcb_graph* g, const lcb_params* p, const lcb_seed* seeds, int64_t nSeeds, LcbProcessor& proc,
                    const LcbEngineConfig& cfg, std::vector<lcb_block>& blocks, LcbEngineStats* stats)
{
    const auto t0 = std::chrono::steady_clock::now();
    auto msSince = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    const int64_t phase = p->phase_size > 0 ? p->phase_size : 256;
    // Round size adapts between 1 and maxRound phases: the speculative launch pays where few results are invalidated
    // (sparse stretches: many seeds that walk a little and commit nothing) and only adds redundant work where every seed
    // of a region is invalidated by the region's first commit, so the size doubles after a round with few recomputed
    // seeds and halves after one with many.
    int maxRound = cfg.roundPhases > 0 ? cfg.roundPhases : 256;
    const bool fixedRound = cfg.roundFixed;
    int roundPhases = fixedRound ? maxRound : 1;
    const int eagerPhases = cfg.eagerPhases < 0 ? 0 : (cfg.eagerPhases ? cfg.eagerPhases : 256);   // how far ahead a dry run plans
    const int maxViews = cfg.maxViews < 0 ? 0 : (cfg.maxViews ? std::min(cfg.maxViews, proc.maxViews()) : proc.maxViews());
    const bool debug = getenv("LCB_ENGINE_DEBUG") != nullptr;                                      // diagnostics only
    // A dry run stops planning ahead beyond this many jobs: a launch is as long as its longest seed as long as every job
    // has a workgroup of its own; jobs beyond the processor's seeds in flight only queue up behind speculation that may
    // never be used.
    const size_t maxJobs = (size_t)std::max(1, cfg.maxJobs > 0 ? cfg.maxJobs : proc.concurrency());
    // (Measured and removed, profiles/r05/ab_fourth.txt: a cap that adapts to the share of void background jobs of the last round. Config 3
    // - heavy, mutually invalidating seeds, 37 % void - gains 5 % with a fixed cap of 512 and nothing with the adaptive one; the light
    // long paths of the k = 25 shapes lose 5 % at 512 and 3 % with the adaptive one.)
    const int predictF = cfg.predictF > 0 ? cfg.predictF : 3;       // how the dry run predicts the F of a conflicting seed:
                                                                    // 1 nothing, 2 the still-free instances of E, 3 a stale F if there is one, else as 2
    const int world = cfg.world > 0 ? cfg.world : 1, rank = cfg.rank;
    if (world > 1 && !cfg.allgather) throw LcbError("world > 1 needs an all-gather callback");
    if (rank < 0 || rank >= world) throw LcbError("bad rank");
    if (cfg.countEvents && world > 1) throw LcbError("event counting runs on one rank");

    g_pageShift = 12;
    while ((g->nPos() >> g_pageShift) > 65536) g_pageShift++;       // the coarse bitmap of a RangeSet: at most 8 KB
    lcb_committer com(g, *p);
    proc.reset();
    LcbEngineStats st;
    st.seeds = nSeeds;
    std::vector<uint64_t> pending;                 // marks not yet applied to the processor's `used` state
    bool inPlan = false;
    auto flush = [&]() {
        if (pending.empty()) return;
        const auto tf = std::chrono::steady_clock::now();
        proc.mark(pending.data(), (int64_t)(pending.size() / 2));
        pending.clear();
        if (!inPlan) st.sectionMs[LCB_SEC_FLUSH] += msSince(tf);       // (inside a dry run it is part of the dry run's time)
    };
    std::vector<RangeSet> epochMarks;              // epochMarks[e]: ranges marked after launch e of this round and before launch e+1
    double lastInvalid = 1.0;
    auto takeMarks = [&]() {
        for (size_t i = 0; i + 1 < com.marks.size(); i += 2) {
            epochMarks.back().add(com.marks[i], com.marks[i + 1]);
            pending.push_back(com.marks[i]); pending.push_back(com.marks[i + 1]);
        }
        com.marks.clear();
    };

    int64_t portion = nSeeds / 50;                                                       // progressPortion_, blocksfinder.h:509-513
    if (portion == 0) portion = 1;
    if (cfg.progress) std::cout << '[' << std::flush;

    Results mine, round, tmp, tmp2;
    std::vector<lcb_seed> sub;
    std::vector<uint32_t> subView;
    std::vector<unsigned char> sendBuf, recvBuf;
    std::vector<Cand> cands;                        // job results of this round
    std::vector<int32_t> eIdx, fIdx;                // per seed of the round: its newest E / F job result in cands, or -1
    std::vector<uint32_t> e0Checked;                // per seed: epochs already checked for its round-launch result
    std::vector<RangeSet> viewSets;                 // predicted mark sets of this round's views
    // Seeds of the round whose round-launch result has a footprint or instances, ascending. The others - after the first rounds
    // almost all: their Path::Init found every occurrence used - read no bit as 0 and commit nothing: no commit can void them and
    // no loop below has to look at them.
    std::vector<int32_t> liveIdx;
    auto liveFrom = [&](int64_t i) { return (size_t)(std::lower_bound(liveIdx.begin(), liveIdx.end(), (int32_t)i) - liveIdx.begin()); };

    // ---- asynchronous job batches (side lanes of the processor). With several ranks (`multi`: every launch is dealt to the ranks) a
    // batch is dealt like a launch - job k of it runs on a lane of rank k % world - and results travel through exchangeSide below.
    const bool multi = world > 1 || (cfg.exchangeAlways && cfg.allgather);
    const bool useSide = !cfg.syncJobs && !cfg.countEvents && proc.sideLanes() > 0;
    // ... and the results a stop cannot go on without are computed while the host still plans the rest of the stop's jobs. With several
    // ranks (round 6) EVERY rank computes them itself instead of being dealt a share: they run against the live state, which is the same on
    // all ranks, and the kernels are deterministic - so every rank holds the same results without a collective, and a stop's own F (one
    // seed: nothing to divide) no longer costs two all-gathers.
    const bool useEarly = useSide;
    std::vector<lcb_seed> earlySeeds;
    struct SideJob { int64_t seed; bool isF; int32_t set, epoch; int lane; int64_t k; uint8_t state; };   // lane: index into `batches`; k: job of the batch; state: 0 in flight, 1 taken, 2 dropped
    struct SideBatch { int lane; int pending; };    // the processor's lane on THIS rank (-1: this rank has no job of the batch), jobs in flight (on all ranks)
    std::vector<SideBatch> batches;                 // of this round
    std::vector<uint8_t> sideSent;                  // multi: per job of sideJobs, 1 = this rank has published its result (exchangeSide)
    std::vector<SideJob> sideJobs;                  // of this round
    size_t sideScan = 0;                            // jobs before this index are no longer in flight
    int64_t frozenTo = 0;                           // seeds of the round before this index have their final phase-start result
    std::vector<int32_t> sideE, sideF;              // per seed of the round: its job in flight for E / F (index into sideJobs), or -1
    std::vector<int> laneOrder;                     // batches with jobs in flight, oldest first
    std::vector<lcb_instance> sInst;
    std::vector<lcb_fp> sFp;
    auto laneDone = [&](int b) {                    // one job of the batch fewer in flight; the last one frees its lane (on every rank)
        if (--batches[(size_t)b].pending == 0) {
            if (batches[(size_t)b].lane >= 0) proc.sideRelease(batches[(size_t)b].lane);
            auto it = std::find(laneOrder.begin(), laneOrder.end(), b);
            if (it != laneOrder.end()) laneOrder.erase(it);
        }
    };
    auto dropSide = [&](int32_t idx) {
        SideJob& sj = sideJobs[(size_t)idx];
        int32_t& ref = (sj.isF ? sideF : sideE)[(size_t)sj.seed];
        if (ref == idx) ref = -1;
        if (sj.state == 0) { sj.state = 2; st.sideVoid++; laneDone(sj.lane); }
    };

    // Runs ProcessVertex::Process for n seeds (seed i against view[i]) on ALL ranks: seed i goes to rank i % world, the
    // per-seed results and footprints are all-gathered (sizes first, then the padded payload), so every rank ends with the
    // same `out`. The size exchange also carries an error flag and a tag of the call (launch ordinal, n, caller's tag): a rank
    // that failed, or ranks that disagree about what they are computing (different LCB knobs, diverged state), stop ALL ranks
    // with an error instead of leaving the others blocked in the next collective.
    uint64_t launchOrdinal = 0;
    std::vector<lcb_seed> shareSeeds;
    std::vector<uint32_t> shareView;
    // All-gather of one variable-size buffer per rank: recvBuf holds world strides of `stride` bytes afterwards. ONE collective when every
    // rank's buffer is small (round 6): each rank sends a fixed-size record - a 32-byte head (size, error flag, ordinal and tag of the call)
    // followed by the first LCB_GATHER_INLINE bytes of its buffer - and only if some rank's buffer is larger does a second collective carry the
    // buffers padded to the largest (until round 5: always two, sizes first). The results of a stop's jobs, the published results of background
    // jobs and the one-byte agreements are almost always small; a round's speculative launch is not. The head carries an error flag and a tag
    // of the call, so that a rank that failed - or ranks that disagree about the collective they are in - stop ALL ranks with an error instead
    // of leaving the others blocked.
    constexpr uint64_t LCB_GATHER_INLINE = 8192 - 32;
    std::vector<unsigned char> gatherSend, gatherRecv;
    auto gatherV = [&](uint64_t failed, const std::string& what, uint64_t tag, uint64_t& stride, std::vector<uint64_t>* sizes) {
        const uint64_t head[4] = {sendBuf.size(), failed, launchOrdinal, tag};
        const uint64_t rec = 32 + LCB_GATHER_INLINE;
        gatherSend.assign((size_t)rec, 0);
        memcpy(gatherSend.data(), head, 32);
        if (!sendBuf.empty()) memcpy(gatherSend.data() + 32, sendBuf.data(), (size_t)std::min<uint64_t>(sendBuf.size(), LCB_GATHER_INLINE));
        gatherRecv.resize((size_t)(rec * world));
        if (cfg.allgather(cfg.allgatherUser, gatherSend.data(), rec, gatherRecv.data())) throw LcbError("all-gather failed");
        st.collectives++;
        uint64_t maxBytes = 0;
        if (sizes) sizes->assign((size_t)world, 0);
        for (int r = 0; r < world; r++) {
            uint64_t h[4];
            memcpy(h, gatherRecv.data() + (size_t)(rec * r), 32);
            if (h[1]) throw LcbError(r == rank ? "rank " + std::to_string(r) + " failed: " + what : "rank " + std::to_string(r) + " failed; stopping all ranks");
            if (h[2] != head[2] || h[3] != head[3])
                throw LcbError("ranks disagree about the launch they are in (are the engine and device options identical on every rank?)");
            maxBytes = std::max(maxBytes, h[0]);
            if (sizes) (*sizes)[(size_t)r] = h[0];
        }
        if (maxBytes <= LCB_GATHER_INLINE) {
            recvBuf.resize((size_t)maxBytes * world);
            for (int r = 0; r < world && maxBytes; r++) memcpy(recvBuf.data() + (size_t)(maxBytes * r), gatherRecv.data() + (size_t)(rec * r) + 32, (size_t)maxBytes);
        } else {
            sendBuf.resize((size_t)maxBytes);
            recvBuf.resize((size_t)maxBytes * world);
            if (cfg.allgather(cfg.allgatherUser, sendBuf.data(), maxBytes, recvBuf.data())) throw LcbError("all-gather failed");
            st.collectives++;
        }
        st.exchanges++;
        stride = maxBytes;
    };
    auto processSharded = [&](const lcb_seed* sd, const uint32_t* view, int64_t n, Results& out, uint64_t tag) {
        launchOrdinal++;
        if (!multi) {
            const auto tp = std::chrono::steady_clock::now();
            proc.ctrSink = cfg.countEvents ? &out.ctr : nullptr;
            proc.process(sd, view, n, out.off, out.inst, out.fpOff, out.fp);
            proc.ctrSink = nullptr;
            if (cfg.countEvents && (int64_t)out.ctr.size() != n) throw LcbError("the processor does not count events (stats mode off?)");
            st.processMs += msSince(tp);
            return;
        }
        shareSeeds.clear(); shareView.clear();
        for (int64_t i = rank; i < n; i += world) { shareSeeds.push_back(sd[i]); if (view) shareView.push_back(view[i]); }
        uint64_t failed = 0;
        std::string what;
        try {
            const auto tp = std::chrono::steady_clock::now();
            proc.process(shareSeeds.data(), view ? shareView.data() : nullptr, (int64_t)shareSeeds.size(), mine.off, mine.inst, mine.fpOff, mine.fp);
            st.processMs += msSince(tp);
            pack(mine, (int64_t)shareSeeds.size(), sendBuf);
        } catch (std::exception& e) { failed = 1; what = e.what(); sendBuf.clear(); }
        uint64_t maxBytes = 0;
        gatherV(failed, what, ((uint64_t)n << 20) ^ tag, maxBytes, nullptr);
        // seed i came from rank i % world, local index i / world
        out.off.assign((size_t)n + 1, 0); out.fpOff.assign((size_t)n + 1, 0);
        std::vector<const unsigned char*> base((size_t)world);
        std::vector<uint64_t> instAt((size_t)world, 0), fpAt((size_t)world, 0), nLocal((size_t)world), instTot((size_t)world, 0);
        for (int r = 0; r < world; r++) {
            base[r] = recvBuf.data() + (size_t)r * maxBytes;
            nLocal[r] = n > r ? (uint64_t)((n - r + world - 1) / world) : 0;
            const uint32_t* h = (const uint32_t*)base[r];
            for (uint64_t j = 0; j < nLocal[r]; j++) instTot[r] += h[2 * j];
        }
        uint64_t ti = 0, tf = 0;
        for (int64_t i = 0; i < n; i++) {
            const uint32_t* h = (const uint32_t*)base[i % world];
            out.off[i] = ti; out.fpOff[i] = tf;
            ti += h[2 * (i / world)]; tf += h[2 * (i / world) + 1];
        }
        out.off[n] = ti; out.fpOff[n] = tf;
        out.inst.resize(ti); out.fp.resize(tf);
        for (int64_t i = 0; i < n; i++) {
            const int r = (int)(i % world);
            const uint32_t* h = (const uint32_t*)base[r];
            const uint32_t ci = h[2 * (i / world)], cf = h[2 * (i / world) + 1];
            const unsigned char* instBase = base[r] + nLocal[r] * 8;
            const unsigned char* fpBase = instBase + instTot[r] * sizeof(lcb_instance);
            if (ci) memcpy(&out.inst[out.off[i]], instBase + instAt[r] * sizeof(lcb_instance), ci * sizeof(lcb_instance));
            if (cf) memcpy(&out.fp[out.fpOff[i]], fpBase + fpAt[r] * sizeof(lcb_fp), cf * sizeof(lcb_fp));
            instAt[r] += ci; fpAt[r] += cf;
        }
    };

    // Lazy tail of a round. Where many results are void (the dense stretches: the head of the seed order) the adaptive size shrinks
    // towards one phase per round - and with it the horizon of the dry runs, which plan the E's of LATER phases of the round against
    // predicted views while the commit is still busy with the F chain of the current phase. A round therefore spans at least
    // `lazySpan` phases; its speculative launch covers the first roundPhases of them (what the adaptation decided is worth computing
    // against the round-start state), the others have no phase-start result until a dry run asks for one as a job - in the
    // background, against the state predicted for their turn.
    // (Only with background batches and predicted views: without side lanes every stop would compute the lazy phases synchronously, without
    // views against the live state - results that the next commit voids. The randomized emulator campaign found that configuration a
    // hundred times slower than the round launch it replaces.)
    const int lazySpan = (!useSide || maxViews <= 0 || cfg.lazySpan < 0) ? 0 : (cfg.lazySpan ? cfg.lazySpan : 8);

    // ---- results the host settles itself. Path::Init (path.h:33-46) creates one instance per UNUSED occurrence of the seed's vertex that carries
    // the seed's character; with none, Process() finds no vertex to go to in either direction and returns nothing (blocksfinder.h:781-786) - and
    // it has read no bit as 0, so that result holds against every later state too (bits only go 0 -> 1). The device's screening kernel decides the
    // same thing for the seeds of a launch; here the host decides it for a seed at the moment its result is needed - against the committer's own
    // bitmap, which IS the live state - so that no launch, no job and no stop is spent on it. P: predicted marks on top of the live state (dry runs).
    const bool hostScreen = cfg.sparseRounds >= 0 && !cfg.countEvents;           // (a counting pass walks through the motions on the device)
    auto usedAt = [&](uint64_t q, const RangeSet* P) -> bool { return ((com.used[(size_t)(q >> 5)] >> (q & 31)) & 1u) != 0 || (P && P->hits(q, q)); };
    auto hostDead = [&](const lcb_seed& sd, const RangeSet* P) -> bool {
        const uint32_t av = (uint32_t)(sd.vid < 0 ? -sd.vid : sd.vid);
        for (uint64_t j = g->occStart[av]; j < g->occStart[av + 1]; j++) {
            const uint64_t q = g->occG[j];
            const bool positive = g->posId[q] == sd.vid;                         // JunctionIterator::IsPositiveStrand
            if ((int32_t)(signed char)(positive ? g->posCh[q] : g->posRevCh[q]) != sd.ch) continue;
            // JunctionSequentialIterator::IsUsed (junctionstorage.h:270-283): + strand bit idx, - strand bit idx-1 (none at the chromosome start)
            const bool isUsed = positive ? usedAt(q, P) : (q > g->chrStart[g->occChr[j]] ? usedAt(q - 1, P) : false);
            if (!isUsed) return false;
        }
        return true;
    };
    // Sparse speculative launches. The seeds are sorted by (count, chromosomes of the occurrences, resolve position) - Bundle::operator<,
    // blocksfinder.h:195-208; the resolve position is the smallest position of a + strand occurrence, on chromosome resolve_chr -, so the seeds
    // of one collinear stretch follow each other a few bp apart ON THEIR RESOLVE CHROMOSOME (stretches that resolve to different chromosomes
    // interleave in the list): a CLUSTER is a run of seeds of one resolve chromosome whose positions ascend in steps of at most clusterGap. The
    // first live seed of a cluster finds the block; every other seed of the cluster lies on that block's path. Those that share its phase see the phase-start state
    // like it and compute the block again (the reference does the same, blocksfinder.h:345-370); those of LATER phases are dead when their
    // phase starts. A launch of a whole round computes the block once per seed of the cluster all the same - thousands of times where the
    // blocks are long (k = 25 shapes: 12 phases per block) -, the adaptive round size answers by shrinking to one phase, and the blocks are
    // found one after the other, one block-sized seed per launch. A sparse round launches the seeds of the FIRST phase of every cluster only
    // and spans as many phases as it takes to collect roundPhases x 256 of them (at most sparseSpan phases): the first phases of many blocks
    // are in flight at once. The other seeds are lazy (no result): dead ones are settled by the host when their phase starts, a live one
    // (the guess was wrong: its cluster holds more than one block) is computed then, like the seeds of a lazy tail. Exactness is untouched.
    const bool sparse = hostScreen && lazySpan > 0 && cfg.sparseRounds > 0;    // (measured on the MI355X, profiles/r06: off by default)
    const int sparseSpan = 1024;
    // (the guess costs time, never exactness: a gap too small launches the seeds of a block's later phases for nothing, one too large leaves the first
    // phase of the next block to a stop of the commit - what a lazy tail does)
    const uint64_t clusterGap = getenv("LCB_CLUSTER_GAP") ? (uint64_t)atoll(getenv("LCB_CLUSTER_GAP")) : 16 * (uint64_t)std::max<int64_t>(p->max_branch, 1);   // (environment: experiments)
    struct ClusterEnd { uint64_t round = 0, pos = 0; int64_t phase = 0; };   // per resolve chromosome: the newest cluster of this round (its last position, the phase it began in)
    std::vector<ClusterEnd> clusterOf(std::max<size_t>(1, (size_t)g->nChr()));
    uint64_t roundStamp = 0;
    const int screenThreads = std::max(1, std::min(omp_get_max_threads(), 32));
    std::vector<uint8_t> isEager, deadE;           // per seed of a sparse round: part of its launch / its phase-start result is the empty one (settled by the host)
    std::vector<lcb_seed> eagerSeeds;
    Results packed;
    for (int64_t pos = 0; pos < nSeeds;) {
        int64_t nRound = std::min<int64_t>(nSeeds - pos, (int64_t)std::max(roundPhases, std::min(lazySpan, maxRound)) * phase);
        int64_t nEager = std::min<int64_t>(nRound, (int64_t)roundPhases * phase);     // seeds [nEager, nRound) are lazy (a sparse round: nEager = number of launched seeds, isEager says which)
        if (sparse) {
            // the launched seeds: phase by phase, the seeds whose cluster began in that phase (the first cluster of the round begins with it: what
            // the round before has left of it is dead by now, or alive and wanted)
            const int64_t budget = (int64_t)roundPhases * phase, spanMax = std::min<int64_t>(nSeeds - pos, (int64_t)sparseSpan * phase);
            isEager.assign((size_t)spanMax, 0);
            int64_t nSel = 0, end = 0;
            roundStamp++;
            for (int64_t ph = 0; ph < spanMax; ph += phase) {
                const int64_t n = std::min<int64_t>(phase, spanMax - ph);
                int64_t sel = 0;
                for (int64_t i = ph; i < ph + n; i++) {
                    const lcb_seed& a = seeds[pos + i];
                    ClusterEnd& c = clusterOf[(size_t)a.resolve_chr < clusterOf.size() ? (size_t)a.resolve_chr : 0];
                    if (c.round != roundStamp || a.resolve_pos < c.pos || a.resolve_pos - c.pos > clusterGap) { c.round = roundStamp; c.phase = ph; }   // a new cluster
                    c.pos = a.resolve_pos;
                    if (c.phase == ph) { isEager[(size_t)i] = 1; sel++; }
                }
                if (ph > 0 && nSel + sel > budget && ph >= (int64_t)lazySpan * phase) break;  // the budget is spent: the round ends in front of this phase
                nSel += sel; end = ph + n;
            }
            nRound = end; nEager = nSel;
            isEager.resize((size_t)nRound);
            eagerSeeds.clear();
            for (int64_t i = 0; i < nRound; i++) if (isEager[(size_t)i]) eagerSeeds.push_back(seeds[pos + i]);
        }
        auto eager = [&](int64_t i) -> bool { return sparse ? isEager[(size_t)i] != 0 : i < nEager; };
        int64_t eagerRecomputed = 0;
        st.rounds++;
        flush();                                    // processor state == live state at the start of phase `pos`
        epochMarks.assign(1, RangeSet());
        // ---- speculative launch of the whole round (dealt to the ranks; on one rank with the ordered commit of the round's clean
        // prefix chained behind its kernels where the processor can do that)
        if (!sparse) processSharded(seeds + pos, nullptr, nEager, round, (uint64_t)pos);
        else {
            processSharded(eagerSeeds.data(), nullptr, nEager, packed, (uint64_t)pos);
            // the launched seeds ascend, so their instances and footprints are in round order already: only the offsets spread out
            round.inst.swap(packed.inst); round.fp.swap(packed.fp);
            round.off.assign((size_t)nRound + 1, 0); round.fpOff.assign((size_t)nRound + 1, 0);
            int64_t k = 0;
            for (int64_t i = 0; i < nRound; i++) {
                round.off[(size_t)i] = packed.off[(size_t)k]; round.fpOff[(size_t)i] = packed.fpOff[(size_t)k];
                if (isEager[(size_t)i]) k++;
            }
            round.off[(size_t)nRound] = packed.off[(size_t)k]; round.fpOff[(size_t)nRound] = packed.fpOff[(size_t)k];
            st.lazySeeds += nRound - nEager;
        }
        const auto tSetup = std::chrono::steady_clock::now();
        if (!sparse && nEager < nRound) {           // the lazy seeds: no result, nothing known
            round.off.resize((size_t)nRound + 1, round.off[(size_t)nEager]); round.fpOff.resize((size_t)nRound + 1, round.fpOff[(size_t)nEager]);
            if (cfg.countEvents) round.ctr.resize((size_t)nRound, lcb_counters{});
            st.lazySeeds += nRound - nEager;
        }
        cands.clear(); viewSets.clear();
        eIdx.assign((size_t)nRound, -1); fIdx.assign((size_t)nRound, -1);
        e0Checked.assign((size_t)nRound, 0);
        liveIdx.clear();
        if (!sparse) {
            for (int64_t i = 0; i < nEager; i++) if (round.off[i + 1] != round.off[i] || round.fpOff[i + 1] != round.fpOff[i]) liveIdx.push_back((int32_t)i);
            for (int64_t i = nEager; i < nRound; i++) liveIdx.push_back((int32_t)i);
        } else {
            // the lazy seeds that are dead NOW are dead for good: what the screening kernel does for the seeds of a launch, on all host threads
            deadE.assign((size_t)nRound, 0);
#pragma omp parallel for num_threads(screenThreads) schedule(dynamic, 1024)
            for (int64_t i = 0; i < nRound; i++) if (!isEager[(size_t)i] && hostDead(seeds[pos + i], nullptr)) deadE[(size_t)i] = 1;
            for (int64_t i = 0; i < nRound; i++) {
                if (isEager[(size_t)i]) { if (round.off[i + 1] != round.off[i] || round.fpOff[i + 1] != round.fpOff[i]) liveIdx.push_back((int32_t)i); }
                else if (!deadE[(size_t)i]) liveIdx.push_back((int32_t)i);
                else st.hostDead++;
            }
        }
        if (hostScreen && !sparse) deadE.assign((size_t)nRound, 0);
        if (useSide) { sideJobs.clear(); sideSent.clear(); batches.clear(); laneOrder.clear(); sideScan = 0; sideE.assign((size_t)nRound, -1); sideF.assign((size_t)nRound, -1); }
        st.sectionMs[LCB_SEC_SETUP] += msSince(tSetup);

        // the newest E result of seed i: instances / footprint / provenance
        auto eInst = [&](int64_t i, const lcb_instance*& r, uint64_t& cnt) {
            if (hostScreen && deadE[(size_t)i]) { r = nullptr; cnt = 0; }
            else if (eIdx[(size_t)i] >= 0) { const Cand& c = cands[(size_t)eIdx[(size_t)i]]; r = c.inst.data(); cnt = c.inst.size(); }
            else { r = round.inst.data() + round.off[i]; cnt = round.off[i + 1] - round.off[i]; }
        };
        // Conditions (1) and (2) of the header for a result computed at launch `epoch` against view `view`, judged against
        // the live state now. `checkedTo` caches the closed epochs already examined; `viewOk` caches (1), which is monotone.
        auto validNow = [&](int32_t epoch, int32_t view, uint32_t& checkedTo, bool* viewOk, const lcb_fp* f, size_t nf) -> bool {
            const RangeSet* P = view >= 0 ? &viewSets[(size_t)view] : nullptr;
            if (P && !(viewOk && *viewOk)) {
                if (!P->allPending([&](const std::pair<uint64_t, uint64_t>& q) { return com.allUsed(q.first, q.second); },
                                   [&](const std::pair<uint64_t, uint64_t>& q) { if (debug) std::cerr << "   (1) over-predicted [" << q.first << "," << q.second << ")\n"; return false; })) { st.overPredicted++; return false; }
                if (viewOk) *viewOk = true;
            }
            const uint32_t last = (uint32_t)epochMarks.size() - 1;
            for (uint32_t e = std::max((uint32_t)epoch, checkedTo); e <= last; e++) {
                if (epochMarks[e].empty()) continue;
                for (size_t k = 0; k < nf; k++) if (epochMarks[e].hitsOutside(f[k].lo, f[k].hi, P)) { if (debug) std::cerr << "   (2) footprint [" << f[k].lo << "," << f[k].hi << "] hit in epoch " << e << "\n"; return false; }
            }
            checkedTo = last;                       // closed epochs need no second look; the open one is re-checked
            return true;
        };
        auto eValidNow = [&](int64_t i) -> bool {
            if (hostScreen && deadE[(size_t)i]) return true;
            bool ok = false;                        // (a lazy seed without a job result: nothing has been computed for it yet)
            if (eIdx[(size_t)i] >= 0) { Cand& c = cands[(size_t)eIdx[(size_t)i]]; ok = validNow(c.epoch, c.view, c.checkedTo, &c.viewOk, c.fp.data(), c.fp.size()); }
            else if (eager(i)) ok = validNow(0, -1, e0Checked[(size_t)i], nullptr, round.fp.data() + round.fpOff[i], (size_t)(round.fpOff[i + 1] - round.fpOff[i]));
            // no exact result at hand: if no unused occurrence is left the seed needs none (called at the phase start: the state it is judged by)
            if (!ok && hostScreen && hostDead(seeds[pos + i], nullptr)) { deadE[(size_t)i] = 1; st.hostDead++; return true; }
            return ok;
        };

        // Every predicted mark of the job's view is true by now - or, inside a dry run, still predicted (simP).
        auto sidePlausible = [&](const SideJob& sj) -> bool {
            if (sj.set < 0) return true;
            return viewSets[(size_t)sj.set].allPending([&](const std::pair<uint64_t, uint64_t>& q) { return com.allUsed(q.first, q.second); },
                                                       [&](const std::pair<uint64_t, uint64_t>&) { return false; });
        };
        // The commit needs the E / F of seed i now and a job for it is in flight: its result is taken (waiting for that one job)
        // if its view came true - at this point every commit the view predicted has either happened or will never happen -,
        // otherwise the job is dropped. true: a new candidate result is in place (validate it like any other).
        // the result (sInst, sFp; r = sidePoll's verdict) of a background job becomes the candidate result of its seed
        auto storeSide = [&](int32_t ref, int r, bool viewChecked) -> bool {
            const SideJob sj = sideJobs[(size_t)ref];
            sideJobs[(size_t)ref].state = 1;
            int32_t& cur = (sj.isF ? sideF : sideE)[(size_t)sj.seed];
            if (cur == ref) cur = -1;
            laneDone(sj.lane);
            if (r != 1) { st.sideFailed++; return false; }
            // The phase-start results of a phase are final once the phase has been validated (the commit below uses them without
            // another look): a background E that arrives after that is of no use.
            if (!sj.isF && sj.seed < frozenTo) { st.sideVoid++; return false; }
            st.sideTaken++;
            int32_t& slot = sj.isF ? fIdx[(size_t)sj.seed] : eIdx[(size_t)sj.seed];
            if (slot < 0) { slot = (int32_t)cands.size(); cands.emplace_back(); }
            Cand& c = cands[(size_t)slot];
            c.epoch = sj.epoch; c.view = sj.set; c.checkedTo = (uint32_t)sj.epoch; c.viewOk = viewChecked || sj.set < 0;
            c.inst = sInst; c.fp = sFp; c.ctr = lcb_counters{};
            return true;
        };
        // Several ranks: the result of a background job lives on the rank that ran it. exchangeSide is a collective every rank enters at
        // the same point of the (identical) commit: each rank publishes the results of its own jobs that have finished since its last
        // exchange - the owner of job `need` (-1: none) waits for that one first - and all ranks store all published results, in the
        // same order. Decisions rest on published results only, so the ranks stay in step whatever the timing of their lanes.
        bool needStored = false;
        auto exchangeSide = [&](int32_t need) {
            const auto tp = std::chrono::steady_clock::now();
            launchOrdinal++;
            uint64_t failed = 0;
            std::string what;
            sendBuf.clear();
            try {
                for (size_t q = sideScan; q < sideJobs.size(); q++) {
                    const SideJob& sj = sideJobs[q];
                    if (sj.state != 0 || (int)(sj.k % world) != rank || sideSent[q]) continue;
                    sInst.clear(); sFp.clear();
                    const int r = proc.sidePoll(batches[(size_t)sj.lane].lane, sj.k / world, (int32_t)q == need, sInst, sFp);
                    if (r == 0) continue;
                    sideSent[q] = 1;
                    const uint32_t h[4] = {(uint32_t)q, (uint32_t)r, r == 1 ? (uint32_t)sInst.size() : 0u, r == 1 ? (uint32_t)sFp.size() : 0u};
                    const size_t at = sendBuf.size();
                    sendBuf.resize(at + 16 + (size_t)h[2] * sizeof(lcb_instance) + (size_t)h[3] * sizeof(lcb_fp));
                    memcpy(sendBuf.data() + at, h, 16);
                    if (h[2]) memcpy(sendBuf.data() + at + 16, sInst.data(), (size_t)h[2] * sizeof(lcb_instance));
                    if (h[3]) memcpy(sendBuf.data() + at + 16 + (size_t)h[2] * sizeof(lcb_instance), sFp.data(), (size_t)h[3] * sizeof(lcb_fp));
                }
            } catch (std::exception& e) { failed = 1; what = e.what(); sendBuf.clear(); }
            uint64_t stride = 0;
            std::vector<uint64_t> sizes;
            gatherV(failed, what, 0x51DEull ^ ((uint64_t)(need + 1) << 20), stride, &sizes);
            needStored = false;
            for (int r = 0; r < world; r++) {
                const unsigned char* b = recvBuf.data() + (size_t)r * stride;
                for (uint64_t at = 0; at + 16 <= sizes[(size_t)r];) {
                    uint32_t h[4];
                    memcpy(h, b + at, 16);
                    const size_t bytes = 16 + (size_t)h[2] * sizeof(lcb_instance) + (size_t)h[3] * sizeof(lcb_fp);
                    if (h[0] >= sideJobs.size() || at + bytes > sizes[(size_t)r]) throw LcbError("engine: a published background result does not fit the batch it names");
                    sInst.resize(h[2]); sFp.resize(h[3]);
                    if (h[2]) memcpy(sInst.data(), b + at + 16, (size_t)h[2] * sizeof(lcb_instance));
                    if (h[3]) memcpy(sFp.data(), b + at + 16 + (size_t)h[2] * sizeof(lcb_instance), (size_t)h[3] * sizeof(lcb_fp));
                    at += bytes;
                    if (sideJobs[h[0]].state != 0) continue;           // (dropped since its owner saw it finish: identical on every rank)
                    const bool ok = storeSide((int32_t)h[0], (int)h[1], (int32_t)h[0] == need);
                    if ((int32_t)h[0] == need) needStored = ok;
                }
            }
            while (sideScan < sideJobs.size() && sideJobs[sideScan].state != 0) sideScan++;
            st.processMs += msSince(tp);
        };
        auto takeSide = [&](int64_t i, bool isF) -> bool {
            const int32_t ref = (isF ? sideF : sideE)[(size_t)i];
            if (ref < 0) return false;
            if (sideJobs[(size_t)ref].state != 0 || !sidePlausible(sideJobs[(size_t)ref])) { dropSide(ref); return false; }
            if (multi) {
                exchangeSide(ref);
                if (sideJobs[(size_t)ref].state == 0) throw LcbError("engine: the owner of a background job did not publish its result");
                return needStored;
            }
            sInst.clear(); sFp.clear();
            const auto tp = std::chrono::steady_clock::now();
            const int r = proc.sidePoll(batches[(size_t)sideJobs[(size_t)ref].lane].lane, sideJobs[(size_t)ref].k, true, sInst, sFp);
            st.processMs += msSince(tp);
            return storeSide(ref, r, true);
        };
        // Background jobs that have finished become candidate results like the jobs of a synchronous launch (a dry run then judges
        // them by their footprints; their lane is free once all its jobs are in). Called at every stop.
        auto harvestSide = [&]() {
            if (multi) { if (sideScan < sideJobs.size()) exchangeSide(-1); return; }
            const auto tp = std::chrono::steady_clock::now();
            for (size_t q = sideScan; q < sideJobs.size(); q++) {
                if (sideJobs[q].state != 0) { if (q == sideScan) sideScan++; continue; }
                sInst.clear(); sFp.clear();
                const int r = proc.sidePoll(batches[(size_t)sideJobs[q].lane].lane, sideJobs[q].k, false, sInst, sFp);
                if (r != 0) storeSide((int32_t)q, r, false);
            }
            st.processMs += msSince(tp);
        };

        // ---- dry run + job launch ---------------------------------------------------------------------------------------
        // Called when the commit below stops at seed `stopAt` of the phase starting at `ph0` (midPhase: it needs F of that
        // seed; otherwise it is at the phase start and needs E of some seeds). Simulates the rest of the round with the
        // results at hand as predictions, collects every seed that will need a new E or F, and launches them all, each
        // against the predicted state at its turn.
        auto planAndLaunch = [&](int64_t ph0, int64_t stopAt, bool midPhase) {
            if (useSide) harvestSide();
            const auto tPlan = std::chrono::steady_clock::now();
            inPlan = true;
            flush();                                // processor state == live state
            inPlan = false;
            st.sectionMs[LCB_SEC_PLAN_FLUSH] += msSince(tPlan);
            const auto tSim = std::chrono::steady_clock::now();
            RangeSet simP;                          // predicted marks on top of the live state
            std::vector<uint8_t> simChr(com.invalidChr.begin(), com.invalidChr.end());
            std::vector<uint32_t> simChrList(com.invalidList.begin(), com.invalidList.end());
            std::vector<LcbViewMark> vmarks;        // marks of the views of this launch
            std::vector<std::pair<uint64_t, uint64_t>> sinceView;   // predicted marks not yet part of a view
            int nViews = 0;                         // views of this launch so far (device ids 1..nViews)
            int32_t curSet = -1;                    // viewSets index of the newest view (-1: the live state)
            struct Job { int64_t seed; bool isF; uint32_t devView; int32_t set; };
            std::vector<Job> jobs;
            auto currentView = [&]() {
                if (!sinceView.empty() && nViews < maxViews) {
                    nViews++;
                    for (auto& q : sinceView) vmarks.push_back(LcbViewMark{(uint32_t)nViews, q.first, q.second});
                    sinceView.clear();
                    viewSets.push_back(simP);
                    curSet = (int32_t)viewSets.size() - 1;
                }
                // with no view left the newest one is used: it under-predicts, which is legal (condition (2) decides)
            };
            auto simAdd = [&](const lcb_instance* r, uint64_t cnt) {
                for (uint64_t k = 0; k < cnt; k++) {
                    if (!simChr[r[k].chr]) { simChr[r[k].chr] = 1; simChrList.push_back(r[k].chr); }
                    uint64_t lo, hi; instRange(g, r[k], lo, hi);
                    if (hi > lo && !(com.allUsed(lo, hi))) { simP.add(lo, hi); sinceView.emplace_back(lo, hi); }
                }
            };
            auto simConflicts = [&](const lcb_instance* r, uint64_t cnt) -> bool {
                for (uint64_t k = 0; k < cnt; k++) {
                    if (!simChr[r[k].chr]) continue;
                    uint64_t lo, hi; instRange(g, r[k], lo, hi);
                    if (hi > lo && (com.anyUsed(lo, hi) || simP.hits(lo, hi - 1))) return true;
                }
                return false;
            };
            // would this result still be exact if the predicted marks came true? (conditions (1) and (2) against live + simP)
            // "Is every predicted mark of view set v true by now or predicted by this dry run?" is asked once per view set and plan, not once
            // per result or job that read the view (hundreds share one): the answer at the first question stands for the rest of the plan
            // (a "yes" stays true - simP only grows; a "no" that would turn into a "yes" a few seeds later only costs a recomputation).
            std::vector<int8_t> setFits(viewSets.size(), (int8_t)-1);
            auto viewFits = [&](int32_t v) -> bool {
                if (v < 0) return true;
                if ((size_t)v >= setFits.size()) setFits.resize(viewSets.size(), (int8_t)-1);
                int8_t& m = setFits[(size_t)v];
                if (m < 0) m = viewSets[(size_t)v].allPending([&](const std::pair<uint64_t, uint64_t>& q) { return com.allUsed(q.first, q.second); },
                                                              [&](const std::pair<uint64_t, uint64_t>& q) { return simP.covers(q.first, q.second); }) ? 1 : 0;
                return m != 0;
            };
            auto simValid = [&](int32_t epoch, int32_t view, uint32_t& checkedTo, const lcb_fp* f, size_t nf) -> bool {
                const RangeSet* P = view >= 0 ? &viewSets[(size_t)view] : nullptr;
                // a predicted mark that is neither true yet nor predicted now voids the result (partly true + partly predicted is rare: treated as void)
                if (P && !viewFits(view)) return false;
                const uint32_t last = (uint32_t)epochMarks.size() - 1;
                for (uint32_t e = std::max((uint32_t)epoch, checkedTo); e <= last; e++) {
                    if (epochMarks[e].empty()) continue;
                    for (size_t k = 0; k < nf; k++) if (epochMarks[e].hitsOutside(f[k].lo, f[k].hi, P)) return false;
                }
                checkedTo = last;                   // closed epochs are final (the open one is looked at again)
                if (!simP.empty()) for (size_t k = 0; k < nf; k++) if (simP.hitsOutside(f[k].lo, f[k].hi, P)) return false;
                return true;
            };
            // a job of an earlier plan is in flight for this result and its view still fits what is true or predicted now: no new job
            auto onItsWay = [&](int64_t j, bool isF) -> bool {
                if (!useSide) return false;
                const int32_t ref = (isF ? sideF : sideE)[(size_t)j];
                if (ref < 0) return false;
                if (sideJobs[(size_t)ref].state == 0 && viewFits(sideJobs[(size_t)ref].set)) return true;
                dropSide(ref);
                return false;
            };
            // Early critical launch: the stop's own jobs - the F of the stopping seed, or the missing E of the phase that is about to
            // start - run against the live state and are known before anything is simulated; their computation begins here and the
            // dry run below runs in its shadow.
            size_t nEarly = 0;                       // the first nEarly jobs of the plan were begun ahead of it
            double earlyMs = 0;                      // (time inside processBegin: processor time, not planning time)
            auto beginEarly = [&]() {
                const auto tp = std::chrono::steady_clock::now();
                if (proc.processBegin(earlySeeds.data(), (int64_t)earlySeeds.size())) nEarly = earlySeeds.size();
                earlyMs = msSince(tp);
                st.processMs += earlyMs;
            };
            if (useEarly && midPhase) { earlySeeds.assign(1, seeds[pos + stopAt]); beginEarly(); }
            const int64_t lim = std::min<int64_t>(nRound, ph0 + (int64_t)(eagerPhases + 1) * phase);
            int64_t screenLeft = 16384;              // seeds this dry run may screen on the host (its time is the stop's time)
            std::vector<lcb_instance> guess;
            size_t lv = liveFrom((midPhase ? stopAt : ph0));          // cursor into liveIdx (seeds that are not in it need nothing, commit nothing)
            for (int64_t ph = ph0; ph < lim; ph += phase) {
                if (jobs.size() >= maxJobs) break;      // enough speculation for one launch (the first job is always there)
                if (hostScreen && screenLeft <= 0 && !jobs.empty()) break;
                const int64_t n = std::min<int64_t>(phase, nRound - ph);
                const bool first = ph == ph0;
                while (lv < liveIdx.size() && liveIdx[lv] < (first && midPhase ? stopAt : ph)) lv++;
                size_t lvEnd = lv;
                while (lvEnd < liveIdx.size() && liveIdx[lvEnd] < ph + n) lvEnd++;
                if (!(first && midPhase)) {
                    // phase start: which seeds will lack an exact E?
                    bool any = false;
                    for (size_t q = lv; q < lvEnd; q++) {
                        const int64_t j = liveIdx[q];
                        bool ok;
                        if (hostScreen && deadE[(size_t)j]) ok = true;
                        else if (eIdx[(size_t)j] >= 0) { Cand& c = cands[(size_t)eIdx[(size_t)j]]; ok = simValid(c.epoch, c.view, c.checkedTo, c.fp.data(), c.fp.size()); }
                        else if (!eager(j)) ok = false;
                        else ok = simValid(0, -1, e0Checked[(size_t)j], round.fp.data() + round.fpOff[j], (size_t)(round.fpOff[j + 1] - round.fpOff[j]));
                        // dead against the live state plus the commits this dry run predicts before its phase: it will need no job (the host
                        // settles it when its phase starts, if the predictions come true; a dead seed of the stop's own phase is settled already)
                        if (!ok && hostScreen && screenLeft > 0 && !(first && !midPhase)) { screenLeft--; if (hostDead(seeds[pos + j], simP.empty() ? nullptr : &simP)) ok = true; }
                        if (!ok && !onItsWay(j, false)) {
                            if (!any) { currentView(); any = true; }
                            jobs.push_back(Job{j, false, (uint32_t)nViews, curSet});
                        }
                    }
                    if (!first) { for (uint32_t c : simChrList) simChr[c] = 0; simChrList.clear(); }
                    else if (useEarly && !jobs.empty() && nViews == 0) {
                        earlySeeds.clear();
                        for (auto& jb : jobs) earlySeeds.push_back(seeds[pos + jb.seed]);
                        beginEarly();
                    }
                }
                // ordered commit, simulated with the newest E of each seed as the prediction of its E
                for (size_t q = lv; q < lvEnd; q++) {
                    const int64_t j = liveIdx[q];
                    const lcb_instance* r; uint64_t cnt;
                    eInst(j, r, cnt);
                    if (cnt <= 1) continue;
                    if (!simConflicts(r, cnt)) { simAdd(r, cnt); continue; }
                    // it will need F
                    bool have = false;
                    if (fIdx[(size_t)j] >= 0) {
                        Cand& c = cands[(size_t)fIdx[(size_t)j]];
                        have = simValid(c.epoch, c.view, c.checkedTo, c.fp.data(), c.fp.size());
                        if (have && c.inst.size() > 1) simAdd(c.inst.data(), c.inst.size());
                    }
                    if (!have) {
                        // (a seed that is dead once the commits predicted before it have happened re-processes to nothing: the host settles that too)
                        if (hostScreen && screenLeft > 0 && !(first && midPhase && j == stopAt)) { screenLeft--; if (hostDead(seeds[pos + j], simP.empty() ? nullptr : &simP)) continue; }
                        if (!onItsWay(j, true)) {
                            currentView();
                            jobs.push_back(Job{j, true, (uint32_t)nViews, curSet});
                        }
                        if (predictF >= 3 && fIdx[(size_t)j] >= 0) {
                            // a stale F (computed against a view that did not come true) is the best guess at hand
                            const Cand& c = cands[(size_t)fIdx[(size_t)j]];
                            if (c.inst.size() > 1) simAdd(c.inst.data(), c.inst.size());
                        } else if (predictF >= 2) {
                            // prediction of F: the parts of E that are free, if the seed's own stretch survives; erring on
                            // the small side is harmless (under-prediction), erring on the large side voids later views
                            guess.clear();
                            for (uint64_t k = 0; k < cnt; k++) {
                                uint64_t lo, hi; instRange(g, r[k], lo, hi);
                                if (hi > lo && !com.anyUsed(lo, hi) && !simP.hits(lo, hi - 1)) guess.push_back(r[k]);
                            }
                            if (guess.size() > 1) simAdd(guess.data(), guess.size());
                        }
                    }
                }
                for (uint32_t c : simChrList) simChr[c] = 0;
                simChrList.clear();
                lv = lvEnd;
            }
            st.sectionMs[LCB_SEC_PLAN_SIM] += msSince(tSim) - earlyMs;
            // ---- launch
            if (jobs.empty()) throw LcbError("engine: commit stopped but the dry run found nothing to compute");
            sub.clear(); subView.clear();
            for (auto& jb : jobs) { sub.push_back(seeds[pos + jb.seed]); subView.push_back(jb.devView); }
            if (debug && getenv("LCB_ENGINE_DEBUG_JOBS"))
                for (size_t k = 0; k < jobs.size(); k++) {
                    const lcb_instance* r; uint64_t cnt; eInst(jobs[k].seed, r, cnt);
                    uint64_t longest = 0, fl = 0;
                    for (uint64_t q = 0; q < cnt; q++) { uint64_t lo, hi; instRange(g, r[q], lo, hi); longest = std::max(longest, hi - lo); }
                    if (fIdx[(size_t)jobs[k].seed] >= 0) for (auto& in : cands[(size_t)fIdx[(size_t)jobs[k].seed]].inst) { uint64_t lo, hi; instRange(g, in, lo, hi); fl = std::max(fl, hi - lo); }
                    std::cerr << "   job " << k << " seed " << (pos + jobs[k].seed) << (jobs[k].isF ? " F" : " E") << " eLongest " << longest << " eCnt " << cnt << " staleF " << (fIdx[(size_t)jobs[k].seed] >= 0 ? (int64_t)fl : -1) << " view " << jobs[k].devView << "\n";
                }
            st.planMs += msSince(tPlan) - earlyMs;
            // The results the commit cannot go on without: the F of the stopping seed, or every missing E of the phase that is about to
            // start. They are the first jobs of the plan and run against the live state.
            size_t nCrit = 1;
            if (!midPhase) while (nCrit < jobs.size() && !jobs[nCrit].isF && jobs[nCrit].seed < ph0 + phase && jobs[nCrit].devView == 0) nCrit++;
            if ((midPhase && jobs[0].seed != stopAt) || jobs[0].devView != 0) throw LcbError("engine: the first job of a plan is not the stop's own");
            // Side lanes: everything else runs in the background (the oldest batch gives way if no lane is free).
            int lane = -1;                            // the batch (index into `batches`) that took the rest of the plan, or < 0
            if (useSide && jobs.size() > nCrit) {
                const auto tp = std::chrono::steady_clock::now();
                // One rank: the processor's lane, or -1 (no lane is free) / -2 (the batch fits no lane). Several ranks: this rank's share
                // of the batch (jobs k with k % world == rank) on a lane of its own processor; the ranks agree on the worst answer, so
                // that all of them either have the batch or fall back together.
                auto beginBatch = [&]() -> int {
                    const lcb_seed* sd = sub.data() + nCrit; const uint32_t* vw = subView.data() + nCrit;
                    const int64_t n = (int64_t)(sub.size() - nCrit);
                    int mine;
                    if (!multi) mine = proc.sideBegin(sd, vw, n, nViews, vmarks.data(), (int64_t)vmarks.size());
                    else {
                        shareSeeds.clear(); shareView.clear();
                        for (int64_t k = rank; k < n; k += world) { shareSeeds.push_back(sd[k]); shareView.push_back(vw[k]); }
                        mine = shareSeeds.empty() ? -3 : proc.sideBegin(shareSeeds.data(), shareView.data(), (int64_t)shareSeeds.size(), nViews, vmarks.data(), (int64_t)vmarks.size());
                        launchOrdinal++;
                        sendBuf.assign(1, (unsigned char)(mine == -1 ? 1 : (mine == -2 ? 2 : 0)));
                        uint64_t stride = 0;
                        gatherV(0, std::string(), 0xBA7C4ull, stride, nullptr);
                        unsigned char worst = 0;
                        for (int r = 0; r < world; r++) worst = std::max(worst, recvBuf[(size_t)r * stride]);
                        if (worst) { if (mine >= 0) proc.sideRelease(mine); return worst == 1 ? -1 : -2; }
                    }
                    if (mine < 0 && mine != -3) return mine;
                    batches.push_back(SideBatch{mine >= 0 ? mine : -1, (int)n});
                    return (int)batches.size() - 1;
                };
                lane = beginBatch();
                if (lane == -1 && !laneOrder.empty()) {      // (-2: the batch fits no lane - nothing would be gained by stopping another one)
                    // every lane holds a batch with jobs still running: the oldest one gives way (what it has finished is kept)
                    const int old = laneOrder.front();
                    for (size_t q = sideScan; q < sideJobs.size(); q++) if (sideJobs[q].lane == old && sideJobs[q].state == 0) dropSide((int32_t)q);
                    lane = beginBatch();
                }
                st.processMs += msSince(tp);
            }
            const size_t nSync = lane >= 0 ? nCrit : jobs.size();
            if (nEarly) {
                // the stop's own jobs are on their way since before the dry run (nothing has touched the processor's state since)
                if (nEarly != nCrit) throw LcbError("engine: the jobs begun ahead of the plan are not the stop's own");
                const auto tp = std::chrono::steady_clock::now();
                proc.processEnd(tmp.off, tmp.inst, tmp.fpOff, tmp.fp);
                st.processMs += msSince(tp);
                st.earlyCritical++;
            }
            if (lane < 0 && nViews > 0) proc.buildViews(nViews, vmarks.data(), (int64_t)vmarks.size());
            if (nViews > 0) st.viewsBuilt += nViews;
            // every rank plans the same jobs (same state, same results); the jobs of one launch are independent given their
            // views, so they are dealt to the ranks like a round's seeds and gathered the same way
            if (!nEarly) processSharded(sub.data(), lane >= 0 ? nullptr : subView.data(), (int64_t)nSync, tmp, (uint64_t)(pos + stopAt));
            else if (nSync > nCrit) {
                // no lane took the rest of the plan: it runs now, and its results follow the early ones
                processSharded(sub.data() + nCrit, subView.data() + nCrit, (int64_t)(nSync - nCrit), tmp2, (uint64_t)(pos + stopAt));
                const uint64_t i0 = tmp.inst.size(), f0 = tmp.fp.size();
                tmp.inst.insert(tmp.inst.end(), tmp2.inst.begin(), tmp2.inst.end());
                tmp.fp.insert(tmp.fp.end(), tmp2.fp.begin(), tmp2.fp.end());
                for (size_t k = 1; k <= nSync - nCrit; k++) { tmp.off.push_back(i0 + tmp2.off[k]); tmp.fpOff.push_back(f0 + tmp2.fpOff[k]); }
            }
            st.recomputeLaunches++; st.recomputedSeeds += (int64_t)jobs.size();
            for (auto& jb : jobs) if (eager(jb.seed)) eagerRecomputed++;
            if (midPhase) st.conflictLaunches++;
            if (debug) std::cerr << "engine: stop at seed " << (pos + stopAt) << (midPhase ? " (F)" : " (E)") << " -> " << jobs.size() << " jobs, " << nViews << " views" << (lane >= 0 ? ", all but the first on side lane " + std::to_string(lane) : std::string()) << "\n";
            epochMarks.emplace_back();              // marks from here on belong to the new epoch
            const int32_t epoch = (int32_t)epochMarks.size() - 1;
            if (lane >= 0) {
                st.sideBatches++; st.sideJobs += (int64_t)(jobs.size() - nCrit);
                laneOrder.push_back(lane);
                for (size_t k = nCrit; k < jobs.size(); k++) {
                    const Job& jb = jobs[k];
                    if (jb.isF) st.conflictSeeds++;
                    int32_t& ref = (jb.isF ? sideF : sideE)[(size_t)jb.seed];
                    if (ref >= 0) dropSide(ref);     // (a job the dry run found unfit was dropped there already)
                    ref = (int32_t)sideJobs.size();
                    sideJobs.push_back(SideJob{jb.seed, jb.isF, jb.set, epoch, lane, (int64_t)(k - nCrit), 0});
                    sideSent.push_back(0);
                }
            }
            for (size_t k = 0; k < nSync; k++) {
                const Job& jb = jobs[k];
                if (jb.isF) st.conflictSeeds++;
                int32_t& slot = jb.isF ? fIdx[(size_t)jb.seed] : eIdx[(size_t)jb.seed];
                if (slot < 0) { slot = (int32_t)cands.size(); cands.emplace_back(); }
                Cand& c = cands[(size_t)slot];
                c.epoch = epoch; c.view = jb.set; c.checkedTo = (uint32_t)epoch; c.viewOk = jb.set < 0;
                c.inst.assign(tmp.inst.begin() + tmp.off[k], tmp.inst.begin() + tmp.off[k + 1]);
                c.fp.assign(tmp.fp.begin() + tmp.fpOff[k], tmp.fp.begin() + tmp.fpOff[k + 1]);
                if (cfg.countEvents) c.ctr = tmp.ctr[k];
            }
        };

        frozenTo = 0;

        // ---- walk the round's phases in order -----------------------------------------------------------------------
        for (int64_t ph = 0; ph < nRound; ph += phase) {
            const int64_t n = std::min<int64_t>(phase, nRound - ph);
            // (a) exact phase-start results for every seed of the phase
            const size_t lv0 = liveFrom(ph), lv1 = liveFrom(ph + n);   // the seeds of the phase that read or commit anything
            double inStop = st.processMs + st.planMs;       // (what the stops inside a section cost is accounted there)
            auto tSec = std::chrono::steady_clock::now();
            for (;;) {
                bool all = true;
                for (size_t q = lv0; q < lv1 && all; q++) {
                    const int64_t i = liveIdx[q];
                    bool ok = eValidNow(i);
                    if (!ok && useSide && takeSide(i, false)) ok = eValidNow(i);
                    if (debug && getenv("LCB_ENGINE_DEBUG_SEED") && pos + i == atoll(getenv("LCB_ENGINE_DEBUG_SEED"))) {
                        std::cerr << "   [seed " << (pos + i) << "] phase-start verdict " << ok << ", E from " << (eIdx[(size_t)i] >= 0 ? "a job" : "the round launch") << ", e0Checked " << e0Checked[(size_t)i] << ", footprint:";
                        const lcb_fp* f = eIdx[(size_t)i] >= 0 ? cands[(size_t)eIdx[(size_t)i]].fp.data() : round.fp.data() + round.fpOff[i];
                        const size_t nf = eIdx[(size_t)i] >= 0 ? cands[(size_t)eIdx[(size_t)i]].fp.size() : (size_t)(round.fpOff[i + 1] - round.fpOff[i]);
                        for (size_t k = 0; k < nf; k++) std::cerr << " [" << f[k].lo << "," << f[k].hi << "]";
                        std::cerr << "\n   epochs:";
                        for (size_t e = 0; e < epochMarks.size(); e++) { std::cerr << " {" << e << ":"; for (auto& q : epochMarks[e].r) std::cerr << " [" << q.first << "," << q.second << ")"; std::cerr << "}"; }
                        std::cerr << "\n";
                    }
                    if (!ok) all = false;
                }
                if (all) break;
                planAndLaunch(ph, ph, false);
            }
            frozenTo = ph + n;
            st.sectionMs[LCB_SEC_VALIDATE] += msSince(tSec) - (st.processMs + st.planMs - inStop);
            inStop = st.processMs + st.planMs; tSec = std::chrono::steady_clock::now();
            // (b) ordered commit (blocksfinder.h:372-414)
            if (cfg.progress) for (int64_t i = ((pos + ph + portion - 1) / portion) * portion; i < pos + ph + n; i += portion) std::cout << '.' << std::flush;
            // the phase-start result of every seed is exact now: its events are the ones the reference's Process() call has
            if (cfg.countEvents) for (int64_t i = ph; i < ph + n; i++) addCounters(st.events, eIdx[(size_t)i] >= 0 ? cands[(size_t)eIdx[(size_t)i]].ctr : round.ctr[(size_t)i]);
            for (size_t q = lv0; q < lv1; q++) {
                const int64_t i = liveIdx[q];
                const lcb_instance* r; uint64_t cnt;
                eInst(i, r, cnt);
                if (cnt <= 1) continue;                                                  // blocksfinder.h:375
                if (!com.conflicts(r, cnt)) {
                    if (debug) {
                        std::cerr << "   commit E of seed " << (pos + i) << " -> block " << (com.blocksFound + 1) << " (" << cnt << " inst)";
                        if (eIdx[(size_t)i] >= 0) { const Cand& c = cands[(size_t)eIdx[(size_t)i]]; std::cerr << " from a job: epoch " << c.epoch << " view set " << c.view; }
                        else std::cerr << " from the round launch";
                        std::cerr << ", open epoch " << (epochMarks.size() - 1) << "\n";
                    }
                    com.finalize(r, cnt); takeMarks(); if (eIdx[(size_t)i] >= 0) st.jobsUsed++; continue;
                }
                st.failures++;                                                           // blocksfinder.h:406
                // re-processed against the live state (blocksfinder.h:407): nothing, if no unused occurrence is left - the usual fate of the seeds
                // that share a phase with the first seed of their block
                if (hostScreen && hostDead(seeds[pos + i], nullptr)) { st.hostDead++; continue; }
                for (;;) {
                    if (fIdx[(size_t)i] >= 0) {
                        Cand& c = cands[(size_t)fIdx[(size_t)i]];
                        if (validNow(c.epoch, c.view, c.checkedTo, &c.viewOk, c.fp.data(), c.fp.size())) break;
                    }
                    if (useSide && takeSide(i, true)) continue;      // a background result for it has arrived: validate that
                    planAndLaunch(ph, i, true);
                }
                const Cand& c = cands[(size_t)fIdx[(size_t)i]];
                st.jobsUsed++;
                if (cfg.countEvents) addCounters(st.events, c.ctr);                      // the re-processing (blocksfinder.h:407) is a second Process() call
                if (debug) {
                    uint64_t eb = 0, fb = 0;
                    for (uint64_t k = 0; k < cnt; k++) { uint64_t lo, hi; instRange(g, r[k], lo, hi); eb += hi - lo; }
                    for (auto& in : c.inst) { uint64_t lo, hi; instRange(g, in, lo, hi); fb += hi - lo; }
                    std::cerr << "   F of seed " << (pos + i) << ": E " << cnt << " inst / " << eb << " pos -> F " << c.inst.size() << " inst / " << fb << " pos -> block "
                              << (com.blocksFound + 1) << " (job of epoch " << c.epoch << ", view set " << c.view << ", open epoch " << (epochMarks.size() - 1) << ")\n";
                }
                if (c.inst.size() > 1) { com.finalize(c.inst.data(), c.inst.size()); takeMarks(); }   // blocksfinder.h:408-411
            }
            com.endPhase();
            st.sectionMs[LCB_SEC_COMMIT] += msSince(tSec) - (st.processMs + st.planMs - inStop);
        }
        if (useSide) for (size_t q = 0; q < sideJobs.size(); q++) if (sideJobs[q].state == 0) dropSide((int32_t)q);   // speculation beyond the round is void
        pos += nRound;
        lastInvalid = (double)eagerRecomputed / (double)nEager;       // (what the adaptation judges: the speculative launch)
        if (!fixedRound) {
            if (lastInvalid > 0.25) roundPhases = std::max(1, roundPhases / 2);
            else if (lastInvalid < 0.05) roundPhases = std::min(maxRound, roundPhases * 2);
        }
    }
    flush();
    if (cfg.progress) std::cout << ']' << std::endl;
    blocks = com.blocks;
    st.blocksFound = com.blocksFound;
    st.wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
}
