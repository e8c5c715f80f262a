// lcb_host.h — host-side internals behind the C ABI in include/lcb.h.
#ifndef LCB_HOST_H
#define LCB_HOST_H

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "lcb.h"

// Structure-of-arrays replacement of Sibelia::JunctionStorage (junctionstorage.h:116-698).
// Every junction occurrence kept by the abundance filter has a FLAT index g = chrStart[chr] + idx (64-bit: the reference bounds a
// chromosome by 2^32, junctionstorage.h:120-151, not the input).
#define LCB_SEG_POSITIONS ((1ull << 32) - (1ull << 20))    // most positions of one device segment (lcb_segments.h), hence of one chromosome
struct lcb_graph {
    int k = 0;
    std::vector<uint64_t> chrStart;        // [C+1]
    std::vector<int32_t> posId;            // [P]  Position::id      (junctionstorage.h:142)
    std::vector<uint32_t> posPos;          // [P]  Position::pos     (junctionstorage.h:143)
    std::vector<uint8_t> posCh;            // [P]  Vertex::ch    = seq[pos+k]                     (junctionstorage.h:641)
    std::vector<uint8_t> posRevCh;         // [P]  Vertex::revCh = ReverseChar(seq[pos-1]) or 'N' (junctionstorage.h:642)
    uint32_t nVertex = 0;                  // vertex_.size() = max|id| + 1
    std::vector<uint64_t> occStart;        // [V+1] CSR over |id|, replaces vertex_ (junctionstorage.h:695)
    std::vector<uint64_t> occG;            // [P]  flat position of each occurrence, sorted by (chr, idx)
    std::vector<uint32_t> occChr;          // [P]
    std::vector<std::string> chrName;      // sequenceDescription_
    std::vector<std::string> seq;          // sequence_ (host only: chars above + block sequences for output)
    uint64_t nPos() const { return posId.size(); }
    uint32_t nChr() const { return (uint32_t)chrName.size(); }
};

struct LcbError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void lcb_set_error(const std::string& msg);

// graph.cpp
lcb_graph* lcb_graph_load_impl(const char* junctionFile, const std::vector<std::string>& fasta, int k, int abundance, int threads);
// bundles.cpp
void lcb_enumerate_seeds_impl(const lcb_graph& g, int threads, std::vector<lcb_seed>& out);
// output.cpp
void lcb_generate_output_impl(const lcb_graph& g, int64_t minBlock, const lcb_block* blocks, int64_t nBlocks, int64_t blocksFound,
                              const std::string& outDir, bool genSeq, int64_t chunks, int64_t* nTrimmed, double* coverage);

// commit.cpp — ordered commit (thread-0 section of ProcessVertex::operator(), blocksfinder.h:372-427)
struct lcb_committer {
    const lcb_graph* g;
    lcb_params p;
    std::vector<uint32_t> used;            // bitmap over g (Position::used, junctionstorage.h:144)
    std::vector<lcb_block> blocks;         // blocksInstance_ (blocksfinder.h:921)
    std::vector<uint64_t> marks;           // (lo, hi) pairs not yet taken by the device
    std::vector<uint8_t> invalidChr;       // invalidChr_ (blocksfinder.h:923)
    std::vector<uint32_t> invalidList;
    int64_t blocksFound = 0, failures = 0;
    lcb_committer(const lcb_graph* graph, const lcb_params& prm);
    bool anyUsed(uint64_t lo, uint64_t hi) const;
    bool allUsed(uint64_t lo, uint64_t hi) const;
    bool conflicts(const lcb_instance* inst, uint64_t n) const;   // the weak check of blocksfinder.h:377-398
    void finalize(const lcb_instance* inst, uint64_t n);
    void endPhase();                                              // invalidChr_.clear(), blocksfinder.h:416
    void commitPhase(const lcb_seed* seeds, int64_t n, const uint64_t* offsets, const lcb_instance* inst,
                     lcb_reprocess_fn fn, void* user);
};

// A footprint interval: flat positions [lo, hi] whose `used` bit a per-seed computation read as 0 (lcb_kernel.h).
struct lcb_fp { uint64_t lo, hi; };

// engine.cpp — the per-rank engine behind the phase loop: something that runs ProcessVertex::Process for a batch of
// seeds against ITS current `used` state. The product's implementation is the HIP device (device.hip); tests plug in
// a callback stand-in. Footprints are optional: an implementation that cannot produce them returns one interval
// [0, UINT32_MAX] per seed, which makes every speculative result conservatively invalid after any commit.
// One range of flat positions that a predicted `used` view has set in addition to the live state: it is set in the
// views firstView, firstView + 1, ... of the launch (views are nested prefixes of the predicted commit sequence).
struct LcbViewMark { uint32_t firstView; uint64_t lo, hi; };

struct LcbProcessor {
    virtual ~LcbProcessor() {}
    // view[i] (null = all 0) selects the `used` state seed i is processed against: 0 = the live state (every mark() so far),
    // v >= 1 = view v of the last buildViews() call.
    virtual void process(const lcb_seed* seeds, const uint32_t* view, int64_t n, std::vector<uint64_t>& off, std::vector<lcb_instance>& inst,
                         std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) = 0;
    virtual void mark(const uint64_t* ranges, int64_t n) = 0;
    virtual void reset() = 0;
    // when set, process() also stores the reference-semantics event counters of every seed here (a processor that can
    // count them: the device in stats mode; others leave it empty)
    std::vector<lcb_counters>* ctrSink = nullptr;
    // predicted views: maxViews() == 0 means the processor has none (everything runs against the live state)
    // Optional: start process(seeds, view 0) now, against the state of this moment (marks applied later must not reach it),
    // and collect it with processEnd — the engine plans the rest of a stop's jobs while the results the stop cannot go on without
    // are computed. false = not supported / not applicable.
    virtual bool processBegin(const lcb_seed* seeds, int64_t n) { (void)seeds; (void)n; return false; }
    virtual void processEnd(std::vector<uint64_t>& off, std::vector<lcb_instance>& inst, std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp)
    {
        (void)off; (void)inst; (void)fpOff; (void)fp; throw LcbError("processEnd without processBegin");
    }
    virtual int maxViews() const { return 0; }
    // seeds the processor works on at the same time: a dry run plans about this many jobs per launch (more only queue up)
    virtual int concurrency() const { return 16384; }
    virtual void buildViews(int nViews, const LcbViewMark* marks, int64_t nMarks) { (void)nViews; (void)marks; (void)nMarks; }

    // ---- asynchronous job batches ("side lanes"), optional. A stop of the ordered commit needs ONE result to go on (the first
    // job of its plan, computed against the live state); the other jobs of the plan are speculation. A processor with side lanes
    // runs that speculation in the background - sideBegin returns at once - while the engine computes the one result it waits for
    // synchronously and goes on committing; a background result is taken (sidePoll) when the commit reaches its seed.
    // The jobs read the live state while it is being marked: that is exact under the engine's validity rule, because every mark
    // applied after sideBegin lies in an epoch the rule checks the job's footprint against (a bit read as 1 because of such a mark
    // is in the true state at the job's turn; a bit read as 0 that has been marked since is inside the footprint).
    virtual int sideLanes() const { return 0; }          // batches that can be in flight at once (0: no side lanes)
    // Starts seed i against view[i] (0 = live state, v >= 1 = view v of the nViews predicted views given by marks; the views are
    // private to the batch). Returns the lane (>= 0), -1: no lane is free, or -2: the batch fits no lane (too many jobs, too many private pages).
    virtual int sideBegin(const lcb_seed* seeds, const uint32_t* view, int64_t n, int nViews, const LcbViewMark* marks, int64_t nMarks)
    {
        (void)seeds; (void)view; (void)n; (void)nViews; (void)marks; (void)nMarks; return -1;
    }
    // Job k of the batch on `lane`: 0 = still running (only if !wait), 1 = done: instances and footprint appended to inst / fp,
    // 2 = no result (the batch was told to stop, or the job needs a kernel variant the lane does not run).
    virtual int sidePoll(int lane, int64_t k, bool wait, std::vector<lcb_instance>& inst, std::vector<lcb_fp>& fp)
    {
        (void)lane; (void)k; (void)wait; (void)inst; (void)fp; return 2;
    }
    // Nobody will ask for the batch's results any more: its jobs stop at their next step, the lane is free once they have.
    virtual void sideRelease(int lane) { (void)lane; }

};

// Side lanes for processors that have none of their own (the test stand-ins: callback, wavefront emulator, oracle model): a
// batch is computed on the spot with the processor's own buildViews + process and kept until it is released. `delay` > 0 makes a
// job invisible to the first `delay` non-waiting polls, so that the engine's "not ready yet" paths run too.
struct LcbEagerSideLanes {
    struct Lane {
        bool busy = false, computed = false;
        std::vector<lcb_seed> seeds; std::vector<uint32_t> view; std::vector<LcbViewMark> marks; int nViews = 0;
        std::vector<uint64_t> off, fpOff; std::vector<lcb_instance> inst; std::vector<lcb_fp> fp; std::vector<int> polls;
    };
    std::vector<Lane> lanes;
    int delay = 0;
    // late: a batch is computed when its first result is asked for, against the live state of THAT moment (plus its predicted
    // marks) - the other extreme of what a background batch that reads the live state while it is being marked can see
    bool late = false;
    int64_t cap = 0;          // > 0: batches of more jobs are refused (the device refuses batches beyond its lanes' capacity)
    explicit LcbEagerSideLanes(int n = 0, int d = 0, bool l = false) : lanes((size_t)n), delay(d), late(l) {}
    void compute(LcbProcessor& p, Lane& L)
    {
        if (L.nViews > 0) p.buildViews(L.nViews, L.marks.data(), (int64_t)L.marks.size());
        p.process(L.seeds.data(), L.view.data(), (int64_t)L.seeds.size(), L.off, L.inst, L.fpOff, L.fp);
        L.computed = true;
    }
    int begin(LcbProcessor& p, const lcb_seed* seeds, const uint32_t* view, int64_t n, int nViews, const LcbViewMark* marks, int64_t nMarks)
    {
        if (cap > 0 && n > cap) return -2;
        for (size_t l = 0; l < lanes.size(); l++) {
            if (lanes[l].busy) continue;
            Lane& L = lanes[l];
            L.seeds.assign(seeds, seeds + n); L.view.assign(view, view + n); L.marks.assign(marks, marks + nMarks); L.nViews = nViews;
            L.busy = true; L.computed = false; L.polls.assign((size_t)n, 0);
            if (!late) compute(p, L);
            return (int)l;
        }
        return -1;
    }
    int poll(LcbProcessor& p, int lane, int64_t k, bool wait, std::vector<lcb_instance>& inst, std::vector<lcb_fp>& fp)
    {
        Lane& L = lanes[(size_t)lane];
        if (!L.busy) return 2;
        if (!wait && L.polls[(size_t)k]++ < delay) return 0;
        if (!L.computed) compute(p, L);
        inst.insert(inst.end(), L.inst.begin() + L.off[(size_t)k], L.inst.begin() + L.off[(size_t)k + 1]);
        fp.insert(fp.end(), L.fp.begin() + L.fpOff[(size_t)k], L.fp.begin() + L.fpOff[(size_t)k + 1]);
        return 1;
    }
    void release(int lane) { lanes[(size_t)lane].busy = false; }
};

// All-gather of a fixed-size buffer across ranks: recv holds world * bytes. Returns 0 on success.
typedef int (*lcb_allgather_fn)(void* user, const void* send, uint64_t bytes, void* recv);

struct LcbEngineConfig {
    int rank = 0, world = 1;
    lcb_allgather_fn allgather = nullptr;
    void* allgatherUser = nullptr;
    int roundPhases = 0;      // phases launched speculatively per round, upper bound of the adaptive size (0 = 256)
    bool progress = false;
    bool roundFixed = false;  // every round has roundPhases phases
    int eagerPhases = 0;      // phases a dry run plans ahead (0 = 256, -1 = none)
    int maxViews = 0;         // predicted views per job launch (0 = all the processor has, -1 = none)
    int maxJobs = 0;          // a dry run stops planning beyond this many jobs (0 = the processor's concurrency)
    int predictF = 0;         // 0 = default (3); 1 nothing, 2 free instances of E, 3 stale F else as 2
    bool countEvents = false; // sum the event counters of exactly the results the reference computes (stats-mode processor, one rank)
    bool exchangeAlways = false;   // world == 1 still goes through pack / all-gather / unpack (tests of the exchange path)
    int lazySpan = 0;         // a round spans at least this many phases: those beyond the adaptive launch size get their phase-start results as jobs (0 = 8, -1 = off)
    bool syncJobs = false;    // never use the processor's side lanes: every job of a stop's plan runs in one synchronous launch (the round-2 engine)
    int sparseRounds = 0;     // lcb_hooks.sparse_rounds: 0 = host-settled dead seeds (default), 1 = also sparse speculative launches, -1 = neither
};

enum { LCB_SEC_SETUP = 0,      // per round: bookkeeping after the round launch (lists of the seeds that read or commit anything)
       LCB_SEC_VALIDATE,       // phase-start validation of footprints against the marks since a result's launch
       LCB_SEC_COMMIT,         // weak conflict check, Finalize, marks into the epochs
       LCB_SEC_FLUSH,          // marks handed to the processor (LcbProcessor::mark)
       LCB_SEC_PLAN_FLUSH,     // inside the dry runs (planMs): marks handed to the processor before a plan ...
       LCB_SEC_PLAN_SIM };     // ... and the simulation of the rest of the round

struct LcbEngineStats {
    int64_t seeds = 0, blocksFound = 0, failures = 0, rounds = 0, recomputeLaunches = 0, recomputedSeeds = 0, conflictLaunches = 0,
            conflictSeeds = 0, exchanges = 0,
            collectives = 0;      // all-gathers issued for them (one per exchange whose buffers are all small, else two)
    // predictive job launches: recomputeLaunches/recomputedSeeds count them and their jobs; of those jobs,
    int64_t jobsUsed = 0;         // ... results that were committed from (exactly validated)
    int64_t viewsBuilt = 0;       // predicted `used` views materialised
    int64_t overPredicted = 0;    // job results dropped because their view held a mark that did not come true
    int64_t earlyCritical = 0;    // stops whose own jobs ran while the rest was planned
    int64_t lazySeeds = 0;        // seeds of the lazy tails of the rounds (no speculative launch: their phase-start results are jobs)
    int64_t hostDead = 0;         // results the host settled without the device: no unused occurrence of the seed's vertex with its character (empty Path::Init)
    int64_t sideBatches = 0, sideJobs = 0;      // asynchronous job batches and their jobs (recomputeLaunches / recomputedSeeds count them too)
    int64_t sideTaken = 0;        // ... results taken when the commit reached their seed (view came true)
    int64_t sideVoid = 0;         // ... jobs dropped because a mark of their view did not come true, or superseded by a newer plan
    int64_t sideFailed = 0;       // ... jobs that ended without a result (stopped, or needed another kernel variant)
    double wallMs = 0;
    double processMs = 0, planMs = 0;   // wall time inside the processor (launches + result gathering) / inside the dry runs
    double sectionMs[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // LCB_SEC_*: where the rest of the host's time goes (diagnostics: LCB_VERBOSE)
    lcb_counters events{};              // countEvents: totals over the phase-start result of every seed + the re-processed result of every conflict
};

void lcb_engine_run(const lcb_graph* g, const lcb_params* p, const lcb_seed* seeds, int64_t nSeeds, LcbProcessor& proc,
                    const LcbEngineConfig& cfg, std::vector<lcb_block>& blocks, LcbEngineStats* stats);

#endif
