// TEST-ONLY: the product's round engine (csrc/engine.cpp, unmodified) driven by the CPU oracle instead of the device, at sizes the
// wavefront emulator cannot reach (hundreds of thousands of seeds). Every process() call of the engine is a "launch"; the
// oracle's per-seed event counters (pushes) and pool sizes give a cost model of that launch, so that engine policies (job cap,
// F prediction, round sizing) can be compared on realistic inputs in a container without a GPU. The blocks are checked against
// the oracle's own FindBlocks: the engine with oracle footprints must reproduce them exactly.
//
//   engine_model <graph> <fasta> <k> <b> <m> <a> [seed limit]
//   environment: MODEL_THREADS (default 8), MODEL_LOG=1 (one line per launch), LCB_MAX_JOBS, LCB_PREDICT_F, LCB_EAGER_PHASES,
//                MODEL_ROUNDS (upper bound of phases per round, default 256), MODEL_CONCURRENCY (default 1280)
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unordered_map>
#include <omp.h>
#include "lcb.h"
#include "lcb_host.h"
extern "C" {
#include "../../oracle/lcb_oracle.h"
size_t orc_used_stride(void);
}

extern "C" const char* lcb_last_error(void) { return ""; }     // (capi.cpp is not linked)

namespace {

int envInt(const char* n, int d) { const char* e = getenv(n); return e && *e ? atoi(e) : d; }

struct Launch { int64_t n = 0, live = 0, maxPush = 0, sumPush = 0, nBig = 0, maxPushBig = 0, nWideOvf = 0, maxPool = 0; bool jobs = false; };

struct OracleProc : LcbProcessor {
    const lcb_graph* g = nullptr;
    orc_params op;
    int threads = 8;
    std::vector<orc_graph*> og;            // one copy of the oracle's state per thread (each job sees its own `used` view)
    std::vector<orc_worker*> ow;
    std::vector<LcbViewMark> vmarks;
    int nViews = 0, views = 256, conc = 1280;
    std::vector<Launch> launches;
    int stride = 1;
    // the last result of every (vertex, character): how many recomputations reproduce it, and how long they were
    struct LastRec { std::vector<lcb_instance> first; int64_t second = 0; std::vector<lcb_fp> fp; size_t markLen = 0; };
    std::unordered_map<uint64_t, LastRec> last;
    // Pricing of resumable seeds (round 6): the marks committed since a seed's previous computation that lie inside its previous footprint are what voided
    // that result; the new computation is identical to the old one up to its first read of one of those positions (the oracle counts the pushes made until
    // then: orc_watch_*), so a checkpoint taken before that read would be a valid restart. resume*: sums over all recomputations.
    std::vector<std::pair<uint64_t, uint64_t>> markLog;          // every range ever marked, in order
    int64_t resumeN = 0, resumeNoMarks = 0, resumeWhole = 0, resumePushes = 0, resumePrefix = 0, resumePushesHit = 0;
    int64_t criticalResume = 0;                                  // sum over the synchronous launches of their longest seed with the resumable prefix taken off
    const bool resumeClock = getenv("MODEL_RESUME") != nullptr;  // the virtual clock runs with resumption (durations without the resumable prefix)
    const int64_t ckEvery = 32;                                  // LCB_CK_EVERY: a checkpoint every 32 pushes
    int64_t recomputed = 0, identical = 0, identicalPushes = 0, recomputedPushes = 0, launchesLongestIdentical = 0;
    int64_t criticalNew = 0;       // critical path if no launch had to wait for a seed that merely reproduces its previous result
    // Virtual clock in pushes (a synchronous launch lasts as long as its longest seed; enough workgroups for every seed of a launch) and
    // side lanes on it (MODEL_SIDE_LANES=n): a background batch starts at the clock of its sideBegin, job k is done `pushes of k` later;
    // waiting for a job moves the clock to its finish. The clock at the end is the critical path of the asynchronous engine.
    int64_t now = 0, sideWaited = 0, sidePushes = 0;
    bool inSide = false;
    std::vector<int64_t> lastPushes;
    struct Lane { bool busy = false; int64_t start = 0; std::vector<int64_t> dur; std::vector<uint64_t> off, fpOff; std::vector<lcb_instance> inst; std::vector<lcb_fp> fp; };
    std::vector<Lane> lanes;
    int sideLanes() const override { return (int)lanes.size(); }
    int sideBegin(const lcb_seed* seeds, const uint32_t* view, int64_t n, int nv, const LcbViewMark* marks, int64_t nMarks) override
    {
        for (size_t l = 0; l < lanes.size(); l++) {
            if (lanes[l].busy) continue;
            Lane& L = lanes[l];
            if (nv > 0) buildViews(nv, marks, nMarks);
            inSide = true;
            process(seeds, view, n, L.off, L.inst, L.fpOff, L.fp);
            inSide = false;
            L.dur = lastPushes; L.start = now; L.busy = true;
            return (int)l;
        }
        return -1;
    }
    int sidePoll(int lane, int64_t k, bool wait, std::vector<lcb_instance>& inst, std::vector<lcb_fp>& fp) override
    {
        Lane& L = lanes[(size_t)lane];
        if (!L.busy) return 2;
        const int64_t ready = L.start + L.dur[(size_t)k];
        if (now < ready) { if (!wait) return 0; sideWaited += ready - now; now = ready; }
        inst.insert(inst.end(), L.inst.begin() + L.off[(size_t)k], L.inst.begin() + L.off[(size_t)k + 1]);
        fp.insert(fp.end(), L.fp.begin() + L.fpOff[(size_t)k], L.fp.begin() + L.fpOff[(size_t)k + 1]);
        return 1;
    }
    void sideRelease(int lane) override { lanes[(size_t)lane].busy = false; }
    // the engine's early critical launch (with side lanes; MODEL_NO_EARLY=1 refuses it): computed at the begin (nothing changes the state before the end), handed out at the end
    bool begun = false;
    int64_t bReady = 0;
    std::vector<uint64_t> bOff, bFpOff; std::vector<lcb_instance> bInst; std::vector<lcb_fp> bFp;
    bool processBegin(const lcb_seed* seeds, int64_t n) override
    {
        if (begun || getenv("MODEL_NO_EARLY")) return false;
        const int64_t t0 = now;
        process(seeds, nullptr, n, bOff, bInst, bFpOff, bFp);
        bReady = now; now = t0;                 // on the virtual clock the launch runs beside whatever the engine does until the end
        begun = true;
        return true;
    }
    void processEnd(std::vector<uint64_t>& off, std::vector<lcb_instance>& inst, std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) override
    {
        if (!begun) throw LcbError("processEnd without processBegin");
        begun = false;
        if (now < bReady) now = bReady;
        off = bOff; inst = bInst; fpOff = bFpOff; fp = bFp;
    }

    void setRange(orc_graph* o, uint64_t lo, uint64_t hi, std::vector<uint8_t*>* undo)
    {
        // flat positions [lo, hi) -> (chromosome, index)
        const std::vector<uint64_t>& cs = g->chrStart;
        size_t c = (size_t)(std::upper_bound(cs.begin(), cs.end(), lo) - cs.begin()) - 1;
        for (uint64_t q = lo; q < hi; q++) {
            while (q >= cs[c + 1]) c++;
            uint8_t* u = orc_chr_used(o, (int64_t)c) + (q - cs[c]) * (uint64_t)stride;
            if (!*u) { *u = 1; if (undo) undo->push_back(u); }
        }
    }

    void process(const lcb_seed* seeds, const uint32_t* view, int64_t n, std::vector<uint64_t>& off, std::vector<lcb_instance>& inst,
                 std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) override
    {
        std::vector<std::vector<lcb_instance>> ri((size_t)n);
        std::vector<std::vector<lcb_fp>> rf((size_t)n);
        std::vector<int64_t> pushes((size_t)n, 0), pool((size_t)n, 0);
        std::vector<int64_t> own((size_t)n, 0), firstHit((size_t)n, -2);      // pushes without the replay's; pushes before the first read of a voiding mark (-1: none read, -2: no mark inside the old footprint)
        const int64_t chunk = 16;
        const int64_t nChunks = (n + chunk - 1) / chunk;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (int64_t cI = 0; cI < nChunks; cI++) {
            const int t = omp_get_thread_num();
            orc_graph* o = og[(size_t)t];
            std::vector<uint8_t*> undo;
            uint32_t applied = 0;           // marks with firstView <= applied are set
            std::vector<orc_inst> buf(1 << 16);
            std::vector<orc_fp> fbuf(1 << 16);
            for (int64_t i = cI * chunk; i < std::min(n, (cI + 1) * chunk); i++) {
                const uint32_t v = view ? view[i] : 0u;
                if (v < applied) { for (uint8_t* u : undo) *u = 0; undo.clear(); applied = 0; }
                if (v > applied) {
                    for (auto& mk : vmarks) if (mk.firstView > applied && mk.firstView <= v) setRange(o, mk.lo, mk.hi, &undo);
                    applied = v;
                }
                // the marks that voided the seed's previous result: committed since (markLog) or predicted by this job's view, inside the previous footprint
                std::vector<uint8_t*> tagged;
                bool prev = false;
                {
                    const uint64_t key = ((uint64_t)(uint32_t)seeds[i].vid << 8) | (uint8_t)seeds[i].ch;
                    auto it = last.find(key);
                    if (it != last.end() && !it->second.fp.empty()) {
                        prev = true;
                        std::vector<std::pair<uint64_t, uint64_t>> iv;                  // the old footprint as disjoint ascending intervals [lo, hi]
                        for (const lcb_fp& f : it->second.fp) iv.emplace_back(f.lo, f.hi);
                        std::sort(iv.begin(), iv.end());
                        size_t w = 0;
                        for (size_t r = 1; r < iv.size(); r++) { if (iv[r].first <= iv[w].second + 1) iv[w].second = std::max(iv[w].second, iv[r].second); else iv[++w] = iv[r]; }
                        iv.resize(w + 1);
                        auto tag = [&](uint64_t mlo, uint64_t mhi) {                    // [mlo, mhi)
                            if (mlo >= mhi) return;
                            size_t a = (size_t)(std::upper_bound(iv.begin(), iv.end(), std::make_pair(mlo, (uint64_t)~0ull)) - iv.begin());
                            if (a > 0 && iv[a - 1].second >= mlo) a--;
                            const std::vector<uint64_t>& cs = g->chrStart;
                            for (; a < iv.size() && iv[a].first < mhi; a++) {
                                const uint64_t lo = std::max(mlo, iv[a].first), hi = std::min(mhi, iv[a].second + 1);
                                if (lo >= hi) continue;
                                size_t cI2 = (size_t)(std::upper_bound(cs.begin(), cs.end(), lo) - cs.begin()) - 1;
                                for (uint64_t q = lo; q < hi; q++) {
                                    while (q >= cs[cI2 + 1]) cI2++;
                                    uint8_t* u = orc_chr_used(o, (int64_t)cI2) + (q - cs[cI2]) * (uint64_t)stride;
                                    if (*u == 1) { *u = 3; tagged.push_back(u); }
                                }
                            }
                        };
                        for (size_t r = it->second.markLen; r < markLog.size(); r++) tag(markLog[r].first, markLog[r].second);
                        for (auto& mk : vmarks) if (mk.firstView <= v && mk.firstView > 0) tag(mk.lo, mk.hi);
                    }
                }
                orc_watch_begin();
                orc_counters c; memset(&c, 0, sizeof(c));
                int64_t score = 0, nfp = 0;
                const int64_t k = orc_worker_process(ow[(size_t)t], seeds[i].vid, seeds[i].ch, buf.data(), (int64_t)buf.size(), &score, &c, fbuf.data(), (int64_t)fbuf.size(), &nfp);
                {
                    int64_t ownPushes = 0;
                    const int64_t fh = orc_watch_end(&ownPushes);
                    own[(size_t)i] = ownPushes;
                    firstHit[(size_t)i] = !prev ? -3 : (tagged.empty() ? -2 : fh);     // -3: no previous result
                    for (uint8_t* u : tagged) *u = 1;
                }
                if (k > (int64_t)buf.size() || nfp > (int64_t)fbuf.size()) { fprintf(stderr, "model: result too large\n"); exit(2); }
                ri[(size_t)i].resize((size_t)k);
                for (int64_t e = 0; e < k; e++) ri[(size_t)i][(size_t)e] = lcb_instance{buf[e].chr, buf[e].front_idx, buf[e].back_idx, buf[e].positive ? 1u : 0u};
                rf[(size_t)i].resize((size_t)nfp);
                for (int64_t e = 0; e < nfp; e++) {
                    const uint32_t base = fbuf[e].chr < 0 ? 0u : (uint32_t)g->chrStart[(size_t)fbuf[e].chr];
                    const uint32_t lo = base + (uint32_t)fbuf[e].lo, hi = base + (uint32_t)fbuf[e].hi;
                    rf[(size_t)i][(size_t)e] = lcb_fp{lo ? lo - 1 : 0u, hi};      // the - strand reads bit g-1
                }
                pushes[(size_t)i] = (int64_t)c.n_push; pool[(size_t)i] = nfp;
            }
            for (uint8_t* u : undo) *u = 0;
        }
        off.assign((size_t)n + 1, 0); fpOff.assign((size_t)n + 1, 0);
        inst.clear(); fp.clear();
        Launch L; L.n = n; L.jobs = view != nullptr;
        for (int64_t i = 0; i < n; i++) {
            off[(size_t)i] = inst.size(); fpOff[(size_t)i] = fp.size();
            inst.insert(inst.end(), ri[(size_t)i].begin(), ri[(size_t)i].end());
            fp.insert(fp.end(), rf[(size_t)i].begin(), rf[(size_t)i].end());
            const int64_t pu = pushes[(size_t)i], pl = pool[(size_t)i];
            if (pl) L.live++;
            L.sumPush += pu; L.maxPush = std::max(L.maxPush, pu); L.maxPool = std::max(L.maxPool, pl);
            if (pl > 1024) { L.nBig++; L.maxPushBig = std::max(L.maxPushBig, pu); }
            else if (pl > 256) L.nWideOvf++;
        }
        off[(size_t)n] = inst.size(); fpOff[(size_t)n] = fp.size();
        // the resumable prefix of every recomputation (in pushes of the model's clock, which include the replay's)
        std::vector<int64_t> eff(pushes);
        for (int64_t i = 0; i < n; i++) {
            const int64_t fh = firstHit[(size_t)i];
            if (fh == -3) continue;
            resumeN++; resumePushes += own[(size_t)i];
            if (fh == -2) { resumeNoMarks++; continue; }            // voided by something else than a mark in its footprint (a prediction that did not come true)
            resumePushesHit += own[(size_t)i];
            int64_t prefix = fh < 0 ? own[(size_t)i] : (fh / ckEvery) * ckEvery;
            if (fh < 0) resumeWhole++;
            resumePrefix += prefix;
            if (own[(size_t)i] > 0) eff[(size_t)i] = pushes[(size_t)i] - prefix * pushes[(size_t)i] / own[(size_t)i];
        }
        {
            int64_t longest = -1, longestNew = 0; bool longestIdentical = false;
            for (int64_t i = 0; i < n; i++) {
                const uint64_t key = ((uint64_t)(uint32_t)seeds[i].vid << 8) | (uint8_t)seeds[i].ch;
                auto it = last.find(key);
                bool same = false;
                if (it != last.end()) {
                    recomputed++; recomputedPushes += pushes[(size_t)i];
                    same = it->second.first.size() == ri[(size_t)i].size() && (ri[(size_t)i].empty() || !memcmp(it->second.first.data(), ri[(size_t)i].data(), ri[(size_t)i].size() * sizeof(lcb_instance)));
                    if (same) { identical++; identicalPushes += pushes[(size_t)i]; }
                }
                if (pushes[(size_t)i] > longest) { longest = pushes[(size_t)i]; longestIdentical = same; }
                if (!same && pushes[(size_t)i] > longestNew) longestNew = pushes[(size_t)i];
                LastRec& rec = last[key];
                rec.first = ri[(size_t)i]; rec.second = pushes[(size_t)i]; rec.fp = rf[(size_t)i]; rec.markLen = markLog.size();
            }
            if (longestIdentical) launchesLongestIdentical++;
            criticalNew += longestNew;
        }
        lastPushes = resumeClock ? eff : pushes;
        if (inSide) { sidePushes += L.sumPush; return; }       // a background batch: not a launch the commit waits for
        { int64_t mx = 0; for (int64_t i = 0; i < n; i++) mx = std::max(mx, eff[(size_t)i]); criticalResume += mx; if (resumeClock) L.maxPush = mx; }
        now += L.maxPush;
        launches.push_back(L);
        if (getenv("LCB_ENGINE_DEBUG_JOBS")) for (int64_t i = 0; i < n; i++) fprintf(stderr, "   done %lld pushes %lld inst %zu pool %lld\n", (long long)i, (long long)pushes[(size_t)i], ri[(size_t)i].size(), (long long)pool[(size_t)i]);
        if (getenv("MODEL_LOG"))
            fprintf(stderr, "launch %zu %s n=%lld first=%lld live=%lld maxPush=%lld sumPush=%lld pool>1024: %lld (maxPush %lld) pool>256: %lld maxPool=%lld\n", launches.size(), L.jobs ? "jobs " : "round",
                    (long long)L.n, (long long)(n ? pushes[0] : 0), (long long)L.live, (long long)L.maxPush, (long long)L.sumPush, (long long)L.nBig, (long long)L.maxPushBig, (long long)L.nWideOvf, (long long)L.maxPool);
    }
    void mark(const uint64_t* r, int64_t n) override
    {
        for (int64_t i = 0; i < n; i++) markLog.emplace_back(r[2 * i], r[2 * i + 1]);
        for (orc_graph* o : og) for (int64_t i = 0; i < n; i++) setRange(o, r[2 * i], r[2 * i + 1], nullptr);
    }
    void reset() override { for (orc_graph* o : og) orc_reset_used(o); }
    int maxViews() const override { return views; }
    int concurrency() const override { return conc; }
    void buildViews(int nv, const LcbViewMark* marks, int64_t nMarks) override { nViews = nv; vmarks.assign(marks, marks + nMarks); }
};

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: engine_model <graph> <fasta> <k> <b> <m> <a> [seed limit]\n"); return 2; }
    const std::string graph = argv[1], fasta = argv[2];
    const int k = atoi(argv[3]), b = atoi(argv[4]), m = atoi(argv[5]), a = atoi(argv[6]);
    const int64_t limit = argc > 7 ? atoll(argv[7]) : 0;
    try {
        const int threads = envInt("MODEL_THREADS", 8);
        lcb_graph* g = lcb_graph_load_impl(graph.c_str(), {fasta}, k, a, threads);
        lcb_params p; memset(&p, 0, sizeof(p));
        p.k = k; p.min_block = m; p.max_branch = b; p.max_flank = b; p.looking_depth = 8; p.phase_size = 256;
        std::vector<lcb_seed> seeds;
        lcb_enumerate_seeds_impl(*g, threads, seeds);
        if (limit && (int64_t)seeds.size() > limit) seeds.resize((size_t)limit);
        OracleProc proc; proc.g = g; proc.threads = threads;
        proc.op.k = k; proc.op.min_block = m; proc.op.max_branch = b; proc.op.max_flank = b; proc.op.looking_depth = 8;
        proc.stride = (int)orc_used_stride();
        proc.conc = envInt("MODEL_CONCURRENCY", 1280);
        proc.lanes.resize((size_t)envInt("MODEL_SIDE_LANES", 0));
        char err[512];
        const char* fa[1] = {fasta.c_str()};
        for (int t = 0; t < threads; t++) {
            orc_graph* o = orc_load(graph.c_str(), fa, 1, k, a, err, sizeof(err));
            if (!o) { fprintf(stderr, "model: %s\n", err); return 1; }
            proc.og.push_back(o); proc.ow.push_back(orc_worker_new(o, &proc.op));
        }
        LcbEngineConfig cfg;
        cfg.roundPhases = envInt("MODEL_ROUNDS", 0);
        cfg.maxJobs = envInt("LCB_MAX_JOBS", 0);
        if (getenv("LCB_PREDICT_F")) cfg.predictF = std::max(1, envInt("LCB_PREDICT_F", 0));
        if (getenv("LCB_EAGER_PHASES")) cfg.eagerPhases = envInt("LCB_EAGER_PHASES", 0) ? envInt("LCB_EAGER_PHASES", 0) : -1;
        cfg.roundFixed = envInt("LCB_ROUND_FIXED", 0) != 0;
        if (getenv("LCB_LAZY_SPAN")) cfg.lazySpan = envInt("LCB_LAZY_SPAN", 0) ? envInt("LCB_LAZY_SPAN", 0) : -1;
        if (getenv("LCB_SPARSE_ROUNDS")) cfg.sparseRounds = atoi(getenv("LCB_SPARSE_ROUNDS"));   // -1 / 0 / 1 (lcb_hooks.sparse_rounds)
        std::vector<lcb_block> blocks;
        LcbEngineStats es;
        const auto t0 = std::chrono::steady_clock::now();
        lcb_engine_run(g, &p, seeds.data(), (int64_t)seeds.size(), proc, cfg, blocks, &es);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // cost model: variants run one after the other; a launch is as long as its longest seed or its total work over the slots.
        // us per push(+vote): compact 10, wide 8, big 20 (launch trace of config 3, profiles/r02); slots 1280 / 256 / 256
        double tRound = 0, tJobs = 0, tBig = 0;
        int64_t nRound = 0, nJobs = 0, jobSeeds = 0, critical = 0, total = 0, bigLaunches = 0;
        for (const Launch& L : proc.launches) {
            const bool wide = L.n <= 512;
            const double c = wide ? 8.0 : 10.0;
            const double slots = wide ? 256.0 : 1280.0;
            double t = std::max((double)L.maxPush * c, (double)L.sumPush * c / slots) + 50.0;
            double tb = 0;
            if (L.nBig) { tb = (double)L.maxPushBig * 20.0 + 50.0; bigLaunches++; }
            (L.jobs ? tJobs : tRound) += t; tBig += tb;
            if (L.jobs) { nJobs++; jobSeeds += L.n; } else nRound++;
            critical += L.maxPush; total += L.sumPush;
        }
        fprintf(stderr, "model: %zu seeds, %zu blocks, failures %lld, rounds %lld, job launches %lld (%lld jobs, %lld used), conflict launches %lld, over-predicted %lld, lazy seeds %lld, settled by the host %lld, %.1f s\n",
                seeds.size(), blocks.size(), (long long)es.failures, (long long)es.rounds, (long long)es.recomputeLaunches, (long long)es.recomputedSeeds, (long long)es.jobsUsed,
                (long long)es.conflictLaunches, (long long)es.overPredicted, (long long)es.lazySeeds, (long long)es.hostDead, sec);
        if (es.earlyCritical) fprintf(stderr, "model: early critical launches %lld of %lld stops\n", (long long)es.earlyCritical, (long long)es.recomputeLaunches);
        fprintf(stderr, "model: host ms: engine %.0f = processor %.0f + dry runs %.0f + commit / validation / other %.0f\n", es.wallMs, es.processMs, es.planMs, es.wallMs - es.processMs - es.planMs);
        fprintf(stderr, "model:   of the rest: round setup %.0f, validation %.0f, commit %.0f, marks to the processor %.0f | of the dry runs: marks to the processor %.0f, simulation %.0f; views built %lld\n", es.sectionMs[LCB_SEC_SETUP],
                es.sectionMs[LCB_SEC_VALIDATE], es.sectionMs[LCB_SEC_COMMIT], es.sectionMs[LCB_SEC_FLUSH], es.sectionMs[LCB_SEC_PLAN_FLUSH], es.sectionMs[LCB_SEC_PLAN_SIM], (long long)es.viewsBuilt);
        fprintf(stderr, "model: launches %zu (round %lld, job %lld), critical path %lld pushes, total %lld pushes | model ms: rounds %.0f + jobs %.0f + big %.0f (%lld launches) = %.0f\n",
                proc.launches.size(), (long long)nRound, (long long)nJobs, (long long)critical, (long long)total, tRound / 1000, tJobs / 1000, tBig / 1000, (long long)bigLaunches,
                (tRound + tJobs + tBig) / 1000);
        // N ranks: every launch is dealt to the ranks, so the throughput-bound part of a launch divides by N and its longest seed
        // does not; every launch ends with two all-gathers (headers, then instances + footprints) priced at 2 x 60 us for N > 1 (8 ranks on
        // xGMI: latency-bound messages). Background batches are not priced: they fill otherwise idle slots on every rank.
        for (int N : {1, 2, 4, 8}) {
            double t = 0;
            for (const Launch& L : proc.launches) {
                const bool wide = L.n <= 512;
                const double c = wide ? 8.0 : 10.0, slots = (wide ? 256.0 : 1280.0) * N;
                t += std::max((double)L.maxPush * c, (double)L.sumPush * c / slots) + 50.0 + (N > 1 ? 120.0 : 0.0);
                if (L.nBig) t += (double)L.maxPushBig * 20.0 + 50.0 + (N > 1 ? 120.0 : 0.0);
            }
            fprintf(stderr, "model: %d rank%s: %.0f ms of launches (x %.2f)\n", N, N > 1 ? "s" : "", t / 1000, (tRound + tJobs + tBig) / t);
        }
        fprintf(stderr, "model: virtual clock (pushes; a synchronous launch = its longest seed, a background job is ready its own pushes after its batch began): %lld | side lanes %zu: "
                        "%lld batches, %lld jobs (%lld taken, %lld dropped), %lld pushes of background work, the commit waited %lld pushes for background jobs\n",
                (long long)proc.now, proc.lanes.size(), (long long)es.sideBatches, (long long)es.sideJobs, (long long)es.sideTaken, (long long)es.sideVoid, (long long)proc.sidePushes, (long long)proc.sideWaited);
        fprintf(stderr, "model: recomputations %lld (%lld pushes), of which reproduced the previous result of the seed: %lld (%lld pushes); launches whose longest seed was such a reproduction: %lld; critical path without the reproductions: %lld pushes\n",
                (long long)proc.recomputed, (long long)proc.recomputedPushes, (long long)proc.identical, (long long)proc.identicalPushes, (long long)proc.launchesLongestIdentical, (long long)proc.criticalNew);
        fprintf(stderr, "model: resumable seeds (checkpoint every %lld pushes, pushes without the replay's): %lld recomputations of seeds with a previous footprint, %lld pushes; %lld of them (%lld pushes) had a mark inside "
                "their previous footprint - %lld of those never read one (the whole result stands), identical prefix %lld pushes = %.1f %% of their pushes, %.1f %% of all recomputation pushes; %lld had none (a prediction that did not come true)\n",
                (long long)proc.ckEvery, (long long)proc.resumeN, (long long)proc.resumePushes, (long long)(proc.resumeN - proc.resumeNoMarks), (long long)proc.resumePushesHit, (long long)proc.resumeWhole,
                (long long)proc.resumePrefix, 100.0 * proc.resumePrefix / std::max<int64_t>(1, proc.resumePushesHit), 100.0 * proc.resumePrefix / std::max<int64_t>(1, proc.resumePushes), (long long)proc.resumeNoMarks);
        fprintf(stderr, "model: critical path of the synchronous launches %lld pushes; with every recomputation resumed at its last clean checkpoint %lld pushes%s\n", (long long)critical, (long long)proc.criticalResume,
                proc.resumeClock ? " (MODEL_RESUME: the virtual clock above runs with resumption)" : "");
        if (getenv("MODEL_DUMP")) {
            FILE* f = fopen(getenv("MODEL_DUMP"), "w");
            for (auto& bl : blocks) fprintf(f, "%d\t%llu\t%llu\t%llu\n", bl.id, (unsigned long long)bl.chr, (unsigned long long)bl.start, (unsigned long long)bl.end);
            fclose(f);
        }
        // parity of the model itself: the oracle's own FindBlocks on a fresh state
        if (!limit && !getenv("MODEL_NOCHECK")) {
            orc_graph* o = proc.og[0];
            orc_reset_used(o);
            orc_block* ob = nullptr; orc_stats st; orc_counters fc; memset(&fc, 0, sizeof(fc));
            const int64_t nb = orc_find_blocks(o, &proc.op, &ob, &st, &fc);
            int diffs = nb != (int64_t)blocks.size();
            for (int64_t i = 0; i < nb && i < (int64_t)blocks.size(); i++)
                if (ob[i].id != blocks[i].id || ob[i].chr != blocks[i].chr || ob[i].start != blocks[i].start || ob[i].end != blocks[i].end) diffs++;
            for (int64_t i = 0; i < nb && i < (int64_t)blocks.size(); i++)
                if (ob[i].id != blocks[i].id || ob[i].chr != blocks[i].chr || ob[i].start != blocks[i].start || ob[i].end != blocks[i].end) {
                    fprintf(stderr, "model: first difference at block row %lld: oracle id %d chr %llu [%llu,%llu) vs engine id %d chr %llu [%llu,%llu)\n", (long long)i, ob[i].id,
                            (unsigned long long)ob[i].chr, (unsigned long long)ob[i].start, (unsigned long long)ob[i].end, blocks[i].id, (unsigned long long)blocks[i].chr,
                            (unsigned long long)blocks[i].start, (unsigned long long)blocks[i].end);
                    break;
                }
            fprintf(stderr, "model: blocks vs the oracle's FindBlocks: %s (%lld vs %zu)\n", diffs ? "DIFFERENT" : "equal", (long long)nb, blocks.size());
            return diffs ? 1 : 0;
        }
    } catch (std::exception& e) { fprintf(stderr, "model: error: %s\n", e.what()); return 1; }
    return 0;
}
