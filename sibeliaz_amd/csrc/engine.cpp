// engine.cpp — the phase loop of BlocksFinder::FindBlocks (blocksfinder.h:334-433,453-530) as a speculative,
// exactly-validated round engine.
//
// Reference semantics that must hold bit for bit: (i) every seed of a phase of 256 is processed against the `used`
// bits as of the START of its phase; (ii) results are committed strictly in seed order; (iii) a result that fails the
// weak conflict check is re-processed against the live state and committed without a second check.
//
// One phase per launch starves a GPU (256 wavefronts, most of them done in microseconds). Here a ROUND of many phases
// is processed in one launch against the state at the round start. `used` bits only ever go 0 -> 1 and a seed's
// computation is a deterministic function of the bits it read, so a speculative result equals the exact one iff no bit
// inside its footprint (lcb_kernel.h: one position interval per instance ever created) has been set since the round
// snapshot. While walking the round's phases in order, seeds whose footprint intersects the ranges marked since the
// snapshot are recomputed against the exact phase-start state before that phase is committed; the same launch eagerly
// recomputes the invalidated seeds of the next few phases too, and every result carries the EPOCH (launch) that produced
// it: it is valid at its phase iff nothing inside its footprint was marked in that epoch or any later one.
// Conflicting seeds of a phase are re-processed in batches as well: all seeds that currently conflict are launched
// together against the live state, and each of those results is used at its turn iff nothing inside its footprint was
// marked in between (otherwise it is launched again). With world > 1 the round's seeds are dealt round-robin to the
// ranks and the per-seed results + footprints are all-gathered; every rank then runs the identical commit, so all
// ranks hold the same `used` state and block list without further traffic.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "lcb_host.h"

namespace {

// Disjoint, sorted position ranges [lo, hi) marked used since some snapshot.
struct RangeSet {
    std::vector<std::pair<uint64_t, uint64_t>> r;
    void clear() { r.clear(); }
    bool empty() const { return r.empty(); }
    void add(uint64_t lo, uint64_t hi)
    {
        if (hi <= lo) return;
        auto it = std::lower_bound(r.begin(), r.end(), std::make_pair(lo, (uint64_t)0));
        if (it != r.begin() && (it - 1)->second >= lo) --it;
        auto first = it;
        while (it != r.end() && it->first <= hi) { lo = std::min(lo, it->first); hi = std::max(hi, it->second); ++it; }
        it = r.erase(first, it);
        r.insert(it, std::make_pair(lo, hi));
    }
    // any marked position p with lo <= p <= hi ?
    bool hits(uint64_t lo, uint64_t hi) const
    {
        auto it = std::upper_bound(r.begin(), r.end(), std::make_pair(hi, UINT64_MAX));
        if (it == r.begin()) return false;
        --it;
        return it->second > lo;
    }
};

struct Results {                       // per-seed results of one process() call, in call order
    std::vector<uint64_t> off, fpOff;
    std::vector<lcb_instance> inst;
    std::vector<lcb_fp> fp;
};

struct Override {                      // a seed whose result was replaced (recomputed / re-processed)
    std::vector<lcb_instance> inst;
    std::vector<lcb_fp> fp;
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

}  // namespace

void lcb_engine_run(const lcb_graph* g, const lcb_params* p, const lcb_seed* seeds, int64_t nSeeds, LcbProcessor& proc,
                    const LcbEngineConfig& cfg, std::vector<lcb_block>& blocks, LcbEngineStats* stats)
{
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t phase = p->phase_size > 0 ? p->phase_size : 256;
    // Round size adapts between 1 and maxRound phases: speculation pays where few results are invalidated (sparse
    // stretches: many seeds that walk a little and commit nothing) and only adds redundant work where every seed of a
    // region is invalidated by the region's first commit (dense stretches), so the size doubles after a round with few
    // recomputed seeds and halves after one with many.
    int maxRound = cfg.roundPhases;
    if (maxRound <= 0) { const char* e = getenv("LCB_ROUND_PHASES"); maxRound = e && *e ? atoi(e) : 64; }
    if (maxRound < 1) maxRound = 1;
    const char* fixedEnv = getenv("LCB_ROUND_FIXED");
    const bool fixedRound = fixedEnv && *fixedEnv && atoi(fixedEnv) != 0;
    int roundPhases = fixedRound ? maxRound : 1;
    const char* eagerEnv = getenv("LCB_EAGER_PHASES");
    const int eagerPhases = eagerEnv && *eagerEnv ? std::max(0, atoi(eagerEnv)) : 64;
    const int world = cfg.world > 0 ? cfg.world : 1, rank = cfg.rank;
    if (world > 1 && !cfg.allgather) throw LcbError("world > 1 needs an all-gather callback");

    lcb_committer com(g, *p);
    proc.reset();
    LcbEngineStats st;
    st.seeds = nSeeds;
    std::vector<uint64_t> pending;                 // marks not yet applied to the processor's `used` state
    auto flush = [&]() {
        if (pending.empty()) return;
        proc.mark(pending.data(), (int64_t)(pending.size() / 2));
        pending.clear();
    };
    std::vector<RangeSet> epochMarks;              // epochMarks[e]: ranges marked after launch e of this round and before launch e+1
    RangeSet dirtyBatch;                           // marked since the last conflict batch was launched
    auto takeMarks = [&]() {
        for (size_t i = 0; i + 1 < com.marks.size(); i += 2) {
            epochMarks.back().add(com.marks[i], com.marks[i + 1]);
            dirtyBatch.add(com.marks[i], com.marks[i + 1]);
            pending.push_back(com.marks[i]); pending.push_back(com.marks[i + 1]);
        }
        com.marks.clear();
    };

    int64_t portion = nSeeds / 50;                                                       // progressPortion_, blocksfinder.h:509-513
    if (portion == 0) portion = 1;
    if (cfg.progress) std::cout << '[' << std::flush;

    Results mine, round, tmp;
    std::vector<lcb_seed> sub;
    std::vector<unsigned char> sendBuf, recvBuf;
    std::vector<int32_t> ovIdx;                     // per seed of the round: index into overrides or -1
    std::vector<Override> overrides;
    std::vector<uint32_t> epochOf, checkedTo;       // per seed of the round: epoch of its current result / epochs already checked
    std::vector<int32_t> batchIdx;                  // per seed of the phase: conflict-batch result or -1
    std::vector<Override> batch;

    for (int64_t pos = 0; pos < nSeeds;) {
        const int64_t nRound = std::min<int64_t>(nSeeds - pos, (int64_t)roundPhases * phase);
        flush();                                    // processor state == live state at the start of phase `pos`
        epochMarks.assign(1, RangeSet());
        st.rounds++;
        // ---- speculative launch of the whole round (this rank's share) -------------------------------------------
        sub.clear();
        for (int64_t i = rank; i < nRound; i += world) sub.push_back(seeds[pos + i]);
        proc.process(sub.data(), (int64_t)sub.size(), mine.off, mine.inst, mine.fpOff, mine.fp);
        if (world == 1) round = mine;
        else {
            // all-gather: sizes first, then the padded payload
            pack(mine, (int64_t)sub.size(), sendBuf);
            uint64_t myBytes = sendBuf.size();
            std::vector<uint64_t> sizes((size_t)world);
            if (cfg.allgather(cfg.allgatherUser, &myBytes, sizeof(uint64_t), sizes.data())) throw LcbError("all-gather failed");
            const uint64_t maxBytes = *std::max_element(sizes.begin(), sizes.end());
            sendBuf.resize((size_t)maxBytes);
            recvBuf.resize((size_t)maxBytes * world);
            if (cfg.allgather(cfg.allgatherUser, sendBuf.data(), maxBytes, recvBuf.data())) throw LcbError("all-gather failed");
            st.exchanges++;
            // seed i of the round came from rank i % world, local index i / world
            round.off.assign((size_t)nRound + 1, 0); round.fpOff.assign((size_t)nRound + 1, 0);
            std::vector<const unsigned char*> base((size_t)world);
            std::vector<uint64_t> instAt((size_t)world, 0), fpAt((size_t)world, 0), nLocal((size_t)world), instTot((size_t)world, 0);
            for (int r = 0; r < world; r++) {
                base[r] = recvBuf.data() + (size_t)r * maxBytes;
                nLocal[r] = (uint64_t)((nRound - r + world - 1) / world);
                const uint32_t* head = (const uint32_t*)base[r];
                for (uint64_t j = 0; j < nLocal[r]; j++) instTot[r] += head[2 * j];
            }
            uint64_t ti = 0, tf = 0;
            for (int64_t i = 0; i < nRound; i++) {
                const uint32_t* head = (const uint32_t*)base[i % world];
                round.off[i] = ti; round.fpOff[i] = tf;
                ti += head[2 * (i / world)]; tf += head[2 * (i / world) + 1];
            }
            round.off[nRound] = ti; round.fpOff[nRound] = tf;
            round.inst.resize(ti); round.fp.resize(tf);
            for (int64_t i = 0; i < nRound; i++) {
                const int r = (int)(i % world);
                const uint32_t* head = (const uint32_t*)base[r];
                const uint32_t ci = head[2 * (i / world)], cf = head[2 * (i / world) + 1];
                const unsigned char* instBase = base[r] + nLocal[r] * 8;
                const unsigned char* fpBase = instBase + instTot[r] * sizeof(lcb_instance);
                if (ci) memcpy(&round.inst[round.off[i]], instBase + instAt[r] * sizeof(lcb_instance), ci * sizeof(lcb_instance));
                if (cf) memcpy(&round.fp[round.fpOff[i]], fpBase + fpAt[r] * sizeof(lcb_fp), cf * sizeof(lcb_fp));
                instAt[r] += ci; fpAt[r] += cf;
            }
        }
        ovIdx.assign((size_t)nRound, -1);
        overrides.clear();
        epochOf.assign((size_t)nRound, 0);
        checkedTo.assign((size_t)nRound, 0);
        const int64_t recomputedBefore = st.recomputedSeeds;
        // is the current result of seed i still exact, i.e. was nothing inside its footprint marked since its launch?
        auto stillValid = [&](int64_t i) -> bool {
            const lcb_fp* f; size_t nf;
            if (ovIdx[(size_t)i] >= 0) { const Override& o = overrides[(size_t)ovIdx[(size_t)i]]; f = o.fp.data(); nf = o.fp.size(); }
            else { f = round.fp.data() + round.fpOff[i]; nf = (size_t)(round.fpOff[i + 1] - round.fpOff[i]); }
            const uint32_t last = (uint32_t)epochMarks.size() - 1;
            for (uint32_t e = std::max(epochOf[(size_t)i], checkedTo[(size_t)i]); e <= last; e++) {
                if (epochMarks[e].empty()) continue;
                for (size_t k = 0; k < nf; k++) if (epochMarks[e].hits(f[k].lo, f[k].hi)) return false;
            }
            checkedTo[(size_t)i] = last;            // closed epochs need no second look; the open one is re-checked
            return true;
        };

        // ---- walk the round's phases in order -----------------------------------------------------------------------
        for (int64_t ph = 0; ph < nRound; ph += phase) {
            const int64_t n = std::min<int64_t>(phase, nRound - ph);
            // (a) exact phase-start results: recompute what earlier commits may have changed. If this phase has such
            //     seeds, the launch also takes the currently invalid seeds of the next `eager` phases along.
            {
                std::vector<int64_t> which;
                for (int64_t i = ph; i < ph + n; i++) if (!stillValid(i)) which.push_back(i);
                if (!which.empty()) {
                    const int64_t lim = std::min<int64_t>(nRound, ph + n + (int64_t)eagerPhases * phase);
                    for (int64_t i = ph + n; i < lim; i++) if (!stillValid(i)) which.push_back(i);
                    sub.clear();
                    for (int64_t i : which) sub.push_back(seeds[pos + i]);
                    flush();                        // == live state at the start of this phase
                    proc.process(sub.data(), (int64_t)sub.size(), tmp.off, tmp.inst, tmp.fpOff, tmp.fp);
                    st.recomputeLaunches++; st.recomputedSeeds += (int64_t)which.size();
                    epochMarks.emplace_back();      // marks from here on belong to the new epoch
                    const uint32_t epoch = (uint32_t)epochMarks.size() - 1;
                    for (size_t k = 0; k < which.size(); k++) {
                        int32_t& slot = ovIdx[(size_t)which[k]];
                        if (slot < 0) { slot = (int32_t)overrides.size(); overrides.emplace_back(); }
                        Override& o = overrides[(size_t)slot];
                        o.inst.assign(tmp.inst.begin() + tmp.off[k], tmp.inst.begin() + tmp.off[k + 1]);
                        o.fp.assign(tmp.fp.begin() + tmp.fpOff[k], tmp.fp.begin() + tmp.fpOff[k + 1]);
                        epochOf[(size_t)which[k]] = epoch; checkedTo[(size_t)which[k]] = epoch;
                    }
                }
            }
            // (b) ordered commit (blocksfinder.h:372-414) with batched re-processing of conflicts
            batchIdx.assign((size_t)n, -1);
            batch.clear();
            auto resultOf = [&](int64_t i, const lcb_instance*& r, uint64_t& cnt) {
                if (ovIdx[(size_t)i] >= 0) { const Override& o = overrides[(size_t)ovIdx[(size_t)i]]; r = o.inst.data(); cnt = o.inst.size(); }
                else { r = round.inst.data() + round.off[i]; cnt = round.off[i + 1] - round.off[i]; }
            };
            for (int64_t i = ph; i < ph + n; i++) {
                if (cfg.progress && (pos + i) % portion == 0) std::cout << '.' << std::flush;
                const lcb_instance* r; uint64_t cnt;
                resultOf(i, r, cnt);
                if (cnt <= 1) continue;                                                  // blocksfinder.h:375
                if (!com.conflicts(r, cnt)) { com.finalize(r, cnt); takeMarks(); continue; }
                st.failures++;                                                           // blocksfinder.h:406
                bool have = false;
                if (batchIdx[(size_t)(i - ph)] >= 0) {
                    const Override& b = batch[(size_t)batchIdx[(size_t)(i - ph)]];
                    have = true;
                    for (size_t f = 0; f < b.fp.size() && have; f++) if (dirtyBatch.hits(b.fp[f].lo, b.fp[f].hi)) have = false;
                }
                if (!have) {
                    // re-process, against the live state, every seed of the rest of the phase that conflicts right now
                    sub.clear();
                    std::vector<int64_t> which;
                    for (int64_t j = i; j < ph + n; j++) {
                        const lcb_instance* rj; uint64_t cj;
                        resultOf(j, rj, cj);
                        if (cj > 1 && (j == i || com.conflicts(rj, cj))) { which.push_back(j); sub.push_back(seeds[pos + j]); }
                    }
                    flush();
                    dirtyBatch.clear();
                    proc.process(sub.data(), (int64_t)sub.size(), tmp.off, tmp.inst, tmp.fpOff, tmp.fp);
                    st.conflictLaunches++; st.conflictSeeds += (int64_t)which.size();
                    for (size_t k = 0; k < which.size(); k++) {
                        int32_t& slot = batchIdx[(size_t)(which[k] - ph)];
                        if (slot < 0) { slot = (int32_t)batch.size(); batch.emplace_back(); }
                        Override& b = batch[(size_t)slot];
                        b.inst.assign(tmp.inst.begin() + tmp.off[k], tmp.inst.begin() + tmp.off[k + 1]);
                        b.fp.assign(tmp.fp.begin() + tmp.fpOff[k], tmp.fp.begin() + tmp.fpOff[k + 1]);
                    }
                }
                const Override& b = batch[(size_t)batchIdx[(size_t)(i - ph)]];
                if (b.inst.size() > 1) { com.finalize(b.inst.data(), b.inst.size()); takeMarks(); }   // blocksfinder.h:408-411
            }
            com.endPhase();
        }
        pos += nRound;
        if (!fixedRound) {
            const double invalid = (double)(st.recomputedSeeds - recomputedBefore) / (double)nRound;
            if (invalid > 0.25) roundPhases = std::max(1, roundPhases / 2);
            else if (invalid < 0.05) roundPhases = std::min(maxRound, roundPhases * 2);
        }
    }
    flush();
    if (cfg.progress) std::cout << ']' << std::endl;
    blocks = com.blocks;
    st.blocksFound = com.blocksFound;
    st.wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
}
