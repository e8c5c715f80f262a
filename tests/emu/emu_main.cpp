// emu_check — TEST-ONLY: runs the device code of lcb_kernel.h under the wavefront emulator and
// checks it against the CPU oracle (oracle/lcb_oracle.c), seed by seed and end to end. It also
// exercises the product's host code (graph.cpp, bundles.cpp, commit.cpp, output.cpp) without a GPU.
//
//   emu_check <graph.bin> <fasta> <k> <b> <m> <a> <mode> [outdir]
//     mode = seeds-init   every seed against an all-unused table
//            seeds-final  every seed against the oracle's final `used` state
//            find         full phase loop through the product committer + GFF through output.cpp
//            big          like seeds-init but through the global-memory ("big") kernel variant
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <omp.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "emu_runtime.h"
#include "lcb_host.h"
#include "lcb_kernel.h"
#include "lcb_segments.h"
#include "../../oracle/lcb_oracle.h"

extern "C" size_t orc_used_stride(void);

static std::string g_err;
void lcb_set_error(const std::string& m) { g_err = m; }
extern "C" const char* lcb_last_error(void) { return g_err.c_str(); }

namespace {

// One emulated workgroup's private state: workspace slot, work-queue and arena cursors, result buffers.
struct EmuCtx {
    std::vector<uint8_t> slot;
    LcbWork W;
    uint32_t cursor[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<LcbFpOut> fpArena;
    std::vector<LcbSeedOut> out;
    std::vector<LcbSeedCtr> ctr;
    std::vector<uint4> arena;
    std::vector<LcbKSeed> ks;
    std::vector<size_t> which;
};

// A per-position table in the DEVICE's flat index space: the host array itself, or - with the gap hook (EMU_SEG_GAP: unused positions
// between the segments, flat indices beyond 2^32) - a lazily zeroed allocation of which only the segments' pages are ever touched.
template <class T_>
struct DevTable {
    const T_* p = nullptr; void* own = nullptr;
    void set(const T_* host, const LcbSegPlan& pl)
    {
        if (!pl.gap) { p = host; return; }
        own = calloc((size_t)pl.devPositions + 64, sizeof(T_));
        if (!own) throw LcbError("emu: out of memory for a gapped table");
        for (uint32_t sg = 0; sg < pl.nSeg(); sg++) memcpy((T_*)own + pl.segDev[sg], host + pl.segStart[sg], (size_t)(pl.segStart[sg + 1] - pl.segStart[sg]) * sizeof(T_));
        p = (const T_*)own;
    }
    ~DevTable() { free(own); }
};

struct Emu {
    const lcb_graph* g;
    lcb_params p;
    LcbSegPlan plan;                             // EMU_SEG_CAP / EMU_SEG_GAP: many small segments / flat indices beyond 2^32 on a small input
    bool seg = false;
    std::vector<uint2> chrLoHi;
    std::vector<uint32_t> occStart32;
    DevTable<int32_t> dPosId; DevTable<uint32_t> dPosPos, dPosWin; std::vector<uint32_t> winHost; DevTable<uint8_t> dPosCh, dPosRevCh;
    std::vector<uint32_t> used;                  // used: the live bitmap (view 0) over the device's flat index, padded to whole pages, followed by the private pages of the views
    std::vector<uint32_t> viewTab;               // predicted views: page tables (word offset from a live page to the view's copy, 0 = shared)
    size_t usedWords = 0, nPages = 0;
    int nViewsAlloc = 0;
    LcbTables T;
    LcbKParams KP;
    int mode = 0;
    bool big = false;
    std::vector<uint4> occRec;
    std::vector<EmuCtx> ctx;                     // one per host thread
    std::vector<LcbFpOut> fpArena;               // merged results of the last run()
    std::vector<LcbSeedOut> out;
    std::vector<LcbSeedCtr> octr;                // per-seed counters of the last run()
    std::vector<uint4> arena;
    lcb_counters ctr{};
    uint64_t launches = 0, criticalPushes = 0, totalPushes = 0, firstPushes = 0;   // sum over launches of the largest per-seed push count

    Emu(const lcb_graph* graph, const lcb_params& prm, int kernelMode) : g(graph), p(prm), mode(kernelMode), big(kernelMode >= 2)
    {
        const uint64_t segCap = getenv("EMU_SEG_CAP") ? strtoull(getenv("EMU_SEG_CAP"), nullptr, 10) : 0, segGap = getenv("EMU_SEG_GAP") ? strtoull(getenv("EMU_SEG_GAP"), nullptr, 10) : 0;
        // EMU_SEG_MAX=n: at most about n segments whatever the size of the input (the gap hook with more than 2^32 positions costs address space per segment)
        uint64_t segCapEff = segCap;
        if (segCap && getenv("EMU_SEG_MAX")) segCapEff = std::max<uint64_t>(segCap, g->nPos() / (uint64_t)std::max(1, atoi(getenv("EMU_SEG_MAX"))) + 1);
        plan = lcb_plan_segments(*g, segCapEff, segGap);
        seg = plan.nSeg() > 1 || segCap != 0;
        if (seg && getenv("EMU_SEG_VERBOSE")) fprintf(stderr, "emu: %u segments over %zu chromosomes, %llu device positions (gap %llu)\n", plan.nSeg(), (size_t)g->nChr(), (unsigned long long)plan.devPositions, (unsigned long long)plan.gap);
        const size_t pageWords = (size_t)1 << LCB_PAGE_SHIFT;
        usedWords = ((plan.devPositions / 32 + 2) + pageWords - 1) & ~(pageWords - 1);
        nPages = usedWords >> LCB_PAGE_SHIFT;
        used.assign(usedWords, 0);
        viewTab.assign(nPages, 0);
        chrLoHi.resize(g->nChr());
        for (size_t c = 0; c < chrLoHi.size(); c++) chrLoHi[c] = uint2{plan.chrLo[c], plan.chrHi[c]};
        dPosId.set(g->posId.data(), plan); dPosPos.set(g->posPos.data(), plan); dPosCh.set(g->posCh.data(), plan); dPosRevCh.set(g->posRevCh.data(), plan);
        T.chrLoHi = chrLoHi.data(); T.segBase = plan.segDev.data(); T.posId = dPosId.p; T.posPos = dPosPos.p;
        T.posCh = dPosCh.p; T.posRevCh = dPosRevCh.p;
        winHost = lcb_window_table(*g, (uint32_t)p.max_branch); dPosWin.set(winHost.data(), plan); T.posWin = dPosWin.p;
        occStart32.assign(g->occStart.begin(), g->occStart.end());       // (one segment: fewer than 2^32 occurrences)
        T.occStart32 = seg ? nullptr : occStart32.data(); T.occStart64 = seg ? g->occStart.data() : nullptr;
        occRec.resize(g->nPos());
        for (size_t j = 0; j < occRec.size(); j++) {
            const uint64_t q = g->occG[j]; const uint32_t cw = plan.chrWord[g->occChr[j]];
            occRec[j] = uint4{(uint32_t)(q - plan.segStart[cw >> LCB_SEG_SHIFT]), cw, g->posPos[q], (uint32_t)g->posId[q]};
        }
        T.occRec = occRec.data(); T.used = used.data(); T.viewTab = viewTab.data(); T.nPages = (uint32_t)nPages;
        T.nChr = g->nChr(); T.nVertex = g->nVertex; T.nPos = plan.devPositions; T.nSeg = plan.nSeg();
        KP.k = p.k; KP.minBlock = p.min_block; KP.maxBranch = p.max_branch; KP.maxFlank = p.max_flank; KP.depth = p.looking_depth;
        const char* te = getenv("EMU_THREADS");
        int nThreads = te ? atoi(te) : omp_get_max_threads();
        if (nThreads < 1) nThreads = 1;
        ctx.resize((size_t)nThreads);
        for (auto& c : ctx) {
            LcbWork& W = c.W;
            // capacities as the product's device.hip chooses them per kernel variant (compact / wide / big / huge)
            W.pathCap = 65536; W.bodyCap = 32768;
            if (getenv("EMU_PATH_CAP")) { W.pathCap = (uint32_t)atoi(getenv("EMU_PATH_CAP")); W.bodyCap = W.pathCap / 2; }   // tiny path sets: long probe chains, colliding home slots
            W.bestCap = mode == 0 ? LcbCfg<0>::IC : (mode == 1 ? LcbCfg<1>::IC : (mode == 2 ? LcbCfg<2>::IC : 8192));
            W.instCap = mode == 2 ? LcbCfg<2>::IC : (mode == 3 ? 8192 : 0); W.voteCap = mode == 3 ? 65536 : 0;
            W.live = nullptr; W.nLive = nullptr; W.ctr = nullptr;
            LcbSlotLayout L = lcb_slot_layout(W.pathCap, W.bodyCap, W.bestCap, W.instCap, W.voteCap);
            c.slot.assign(L.total, 0);
            int32_t* pk = (int32_t*)(c.slot.data() + L.pKeys);
            for (uint32_t i = 0; i < W.pathCap; i++) pk[i] = LCB_EMPTY_KEY;
            if (mode == 3) { int32_t* vk = (int32_t*)(c.slot.data() + L.vKey); for (uint32_t i = 0; i < W.voteCap; i++) vk[i] = LCB_EMPTY_KEY; }
            W.base = c.slot.data(); W.slotBytes = L.total;
            W.dbg = nullptr; W.cursor = &c.cursor[0]; W.arenaCursor = (unsigned long long*)&c.cursor[2];
            c.arena.resize(1 << 18);
            c.fpArena.resize(1 << 18);
            W.fpCursor = (unsigned long long*)&c.cursor[4];
        }
    }

    // predicted views 1..nViews = live state + the marks whose firstView <= v, copy-on-write over pages like the product's
    // device (device.hip): a page that a mark of view v touches gets a private copy in each of the views v..nViews
    void buildViews(int nViews, const LcbViewMark* marks, int64_t nMarks)
    {
        const size_t pageWords = (size_t)1 << LCB_PAGE_SHIFT, pageBits = pageWords * 32;
        viewTab.assign((size_t)(nViews + 1) * nPages, 0);
        used.resize(usedWords);                                  // drop the private pages of the previous launch
        for (int v = 1; v <= nViews; v++)
            for (int64_t m = 0; m < nMarks; m++) {
                if ((int)marks[m].firstView > v) continue;
                uint64_t mlo, mhi;
                plan.rangeToDev(marks[m].lo, marks[m].hi, mlo, mhi);
                for (uint64_t q = mlo; q < mhi; q++) {
                    const size_t page = q / pageBits;
                    uint32_t& e = viewTab[(size_t)v * nPages + page];
                    if (!e) {
                        const size_t at = used.size();
                        used.resize(at + pageWords);
                        memcpy(&used[at], &used[page * pageWords], pageWords * 4);
                        e = (uint32_t)(at - page * pageWords);
                    }
                    used[page * pageWords + e + (q % pageBits) / 32] |= 1u << (q & 31);
                }
            }
        used.resize(used.size() + pageWords);                    // guard page (a vote walk that leaves its voter's page reads a few words past a private copy)
        T.used = used.data(); T.viewTab = viewTab.data();
        nViewsAlloc = nViews;
    }

    void runCtx(EmuCtx& c)
    {
        LcbWork& W = c.W;
        c.cursor[2] = c.cursor[3] = c.cursor[4] = c.cursor[5] = 0;   // reuse the arenas
        c.out.assign(c.ks.size(), LcbSeedOut{});
        c.ctr.assign(c.ks.size(), LcbSeedCtr{});
        W.ctr = c.ctr.data();
        W.cursorBase = c.cursor[0];
        W.arenaBase = *W.arenaCursor;
        W.fpBase = *W.fpCursor;
        const LcbKSeed* sp = c.ks.data();
        const uint32_t n = (uint32_t)c.ks.size();
        if (!n) return;
        const char* nwEnv = getenv("EMU_NW");
        int nw = nwEnv ? atoi(nwEnv) : 1;
        {   // a seed that overflowed a smaller variant is re-run in THIS mode (runRetry) with the EMU_NW of the case: where this mode has no instantiation with that many
            // wavefronts it runs with one (the fuzz campaign reached this with the small-pool compact variant, two wavefronts, and an overflow into the wide variant)
            const bool ns = getenv("EMU_NOSTATS") != nullptr;
            const bool have = ns ? ((mode == 0 && (nw == 1 || nw == 2)) || (mode == 1 && (nw == 1 || nw == 16)) || (mode == 2 && (nw == 1 || nw == 8 || nw == 16)) || (mode == 3 && (nw == 1 || nw == 8)))
                                 : ((mode == 0 && (nw == 1 || nw == 4)) || (mode == 1 && (nw == 1 || nw == 8 || nw == 16)) || (mode == 2 && (nw == 1 || nw == 4)) || (mode == 3 && (nw == 1 || nw == 4)));
            if (!have && isRetry) nw = 1;
        }
        LcbSeedOut* op = c.out.data(); uint4* ar = c.arena.data(); LcbFpOut* fa = c.fpArena.data();
        const size_t arc = c.arena.size(), fac = c.fpArena.size();
        const bool noStats = getenv("EMU_NOSTATS") != nullptr;     // the shipped instantiation (checkpointed replay, no event counters)
#define EMU_RUN_S(M, ST, NW_, PF, SG) do { if ((NW_) == 1) emu_run_wave(0, [&]() { lcb_process_body<M, ST, 1, PF, SG>(T, KP, sp, n, W, op, ar, arc, fa, fac); }); \
                                       else emu_run_block(0, NW_, [&]() { lcb_process_body<M, ST, NW_, PF, SG>(T, KP, sp, n, W, op, ar, arc, fa, fac); }); } while (0)
#define EMU_RUN(M, ST, NW_, PF) do { if (seg) EMU_RUN_S(M, ST, NW_, PF, true); else EMU_RUN_S(M, ST, NW_, PF, false); } while (0)
        // non-stats = the shipped code path (checkpointed replay, dead-seed early-out); the instrumented variant supplies push counts
        // EMU_COMPACT_SMALL: the compact variant with the small pools (LcbCfg<4>: 128 instances / 512 vote slots; lcb_device_opts.compact_pools)
        const bool small = mode == 0 && getenv("EMU_COMPACT_SMALL") != nullptr;
        if (small && noStats && nw == 1) EMU_RUN(4, false, 1, true);
        else if (small && noStats && nw == 2) EMU_RUN(4, false, 2, true);
        else if (small && !noStats && nw == 1) EMU_RUN(4, true, 1, false);
        else if (small && !noStats && nw == 4) EMU_RUN(4, true, 4, false);
        else
        if (noStats && mode == 0 && nw == 1) EMU_RUN(0, false, 1, true);
        else if (noStats && mode == 0 && nw == 2) EMU_RUN(0, false, 2, true);
        else if (noStats && mode == 1 && nw == 1) EMU_RUN(1, false, 1, true);
        else if (noStats && mode == 1 && nw == 16) EMU_RUN(1, false, 16, true);
        else if (noStats && mode == 2 && nw == 1) EMU_RUN(2, false, 1, true);
        else if (noStats && mode == 2 && nw == 8) EMU_RUN(2, false, 8, true);
        else if (noStats && mode == 2 && nw == 16) EMU_RUN(2, false, 16, true);
        else if (noStats && mode == 3 && nw == 1) EMU_RUN(3, false, 1, true);
        else if (noStats && mode == 3 && nw == 8) EMU_RUN(3, false, 8, true);
        else if (!noStats && mode == 0 && nw == 1) EMU_RUN(0, true, 1, false);
        else if (!noStats && mode == 0 && nw == 4) EMU_RUN(0, true, 4, false);
        else if (!noStats && mode == 1 && nw == 1) EMU_RUN(1, true, 1, false);
        else if (!noStats && mode == 1 && nw == 8) EMU_RUN(1, true, 8, false);
        else if (!noStats && mode == 1 && nw == 16) EMU_RUN(1, true, 16, false);
        else if (!noStats && mode == 2 && nw == 1) EMU_RUN(2, true, 1, false);
        else if (!noStats && mode == 2 && nw == 4) EMU_RUN(2, true, 4, false);
        else if (!noStats && mode == 3 && nw == 1) EMU_RUN(3, true, 1, false);
        else if (!noStats && mode == 3 && nw == 4) EMU_RUN(3, true, 4, false);
        else { fprintf(stderr, "emu: no instantiation for mode %d, EMU_NW=%d, %s\n", mode, nw, noStats ? "no stats" : "stats"); exit(2); }
#undef EMU_RUN
#undef EMU_RUN_S
    }

    // runs the process kernel over the seeds: each host thread emulates ONE wavefront over its share of the seeds
    // (seeds are independent; a thread's work queue feeds it all of its seeds)
    void run(const std::vector<LcbKSeed>& seeds)
    {
        const size_t nT = ctx.size();
        for (auto& c : ctx) { c.ks.clear(); c.which.clear(); }
        for (size_t i = 0; i < seeds.size(); i++) { ctx[i % nT].ks.push_back(seeds[i]); ctx[i % nT].which.push_back(i); }
        for (;;) {
            #pragma omp parallel for schedule(dynamic, 1) num_threads((int)nT)
            for (size_t t = 0; t < nT; t++) runCtx(ctx[t]);
            bool grown = false;                  // a result arena overflowed: enlarge it and run that share again
            for (auto& c : ctx) for (auto& o : c.out) if (o.status == LCB_ST_ARENA_OVF && !grown) { c.arena.resize(c.arena.size() * 4); c.fpArena.resize(c.fpArena.size() * 4); grown = true; }
            if (!grown) break;
        }
        out.assign(seeds.size(), LcbSeedOut{});
        octr.assign(seeds.size(), LcbSeedCtr{});
        arena.clear(); fpArena.clear();
        uint64_t maxPush = 0;
        for (auto& c : ctx)
            for (size_t j = 0; j < c.which.size(); j++) {
                LcbSeedOut o = c.out[j];
                if (o.status == 0) {
                    const uint64_t ao = arena.size(), fo = fpArena.size();
                    arena.insert(arena.end(), c.arena.begin() + o.arenaOff, c.arena.begin() + o.arenaOff + o.nInst);
                    fpArena.insert(fpArena.end(), c.fpArena.begin() + o.fpOff, c.fpArena.begin() + o.fpOff + o.nFp);
                    o.arenaOff = ao; o.fpOff = fo;
                }
                out[c.which[j]] = o;
                if (o.status != 0) continue;     // an overflowed attempt is re-run in a larger mode (runRetry) and counted there
                const LcbSeedCtr& k = c.ctr[j];
                octr[c.which[j]] = k;
                const bool noStats = getenv("EMU_NOSTATS") != nullptr;   // then k is the instrumented variant's profile: c[1] = pushes
                if (!noStats) {
                    ctr.n_walk += k.c[0]; ctr.n_occ += k.c[1]; ctr.n_compat_call += k.c[2]; ctr.n_compat_step += k.c[3];
                    ctr.n_inst_out += k.c[4]; ctr.n_vote += k.c[5]; ctr.n_push += k.c[6]; ctr.n_process += k.c[7];
                }
                const uint64_t pushes = noStats ? k.c[1] : k.c[6];
                maxPush = std::max<uint64_t>(maxPush, pushes); totalPushes += pushes;
            }
        launches++; criticalPushes += maxPush;
        const int pc = getenv("EMU_NOSTATS") ? 1 : 6;
        if (getenv("EMU_LAUNCH_LOG")) fprintf(stderr, "  launch %llu: %zu seeds, longest %llu pushes, first job %llu pushes\n", (unsigned long long)launches, seeds.size(), (unsigned long long)maxPush, (unsigned long long)octr[0].c[pc]);
        firstPushes += out.empty() ? 0 : octr[0].c[pc];
        if (getenv("LCB_ENGINE_DEBUG_JOBS")) for (size_t i = 0; i < out.size(); i++) fprintf(stderr, "   done %zu pushes %llu inst %u\n", i, (unsigned long long)octr[i].c[pc], out[i].nInst);
    }
    // Like the product's retry chain (device.hip): seeds that overflow the LDS capacities of this mode are run again in the next
    // larger mode (small -> medium -> big), against the same `used` views.
    std::unique_ptr<Emu> next;
    bool isRetry = false;          // this emulator runs the seeds that overflowed a smaller variant
    void runRetry(const std::vector<LcbKSeed>& seeds)
    {
        run(seeds);
        std::vector<size_t> again;
        for (size_t i = 0; i < out.size(); i++) if (out[i].status >= LCB_ST_INST_OVF && out[i].status <= LCB_ST_BEST_OVF) again.push_back(i);
        if (again.empty() || mode >= 3) return;
        if (!next) { next.reset(new Emu(g, p, mode + 1)); next->isRetry = true; }
        next->used = used; next->T.used = next->used.data(); next->nViewsAlloc = nViewsAlloc;
        next->viewTab = viewTab; next->T.viewTab = next->viewTab.data();
        std::vector<LcbKSeed> sub;
        for (size_t i : again) sub.push_back(seeds[i]);
        next->runRetry(sub);
        for (size_t k = 0; k < again.size(); k++) {
            LcbSeedOut o = next->out[k];
            if (o.status == 0) {
                const uint64_t ao = arena.size(), fo = fpArena.size();
                arena.insert(arena.end(), next->arena.begin() + o.arenaOff, next->arena.begin() + o.arenaOff + o.nInst);
                fpArena.insert(fpArena.end(), next->fpArena.begin() + o.fpOff, next->fpArena.begin() + o.fpOff + o.nFp);
                o.arenaOff = ao; o.fpOff = fo;
            }
            out[again[k]] = o;
            octr[again[k]] = next->octr[k];
        }
        ctr.n_walk += next->ctr.n_walk; ctr.n_occ += next->ctr.n_occ; ctr.n_compat_call += next->ctr.n_compat_call; ctr.n_compat_step += next->ctr.n_compat_step;
        ctr.n_inst_out += next->ctr.n_inst_out; ctr.n_vote += next->ctr.n_vote; ctr.n_push += next->ctr.n_push; ctr.n_process += next->ctr.n_process;
        memset(&next->ctr, 0, sizeof(next->ctr));
    }
};

struct EmuProcessor : LcbProcessor {
    Emu* emu;
    int views = 0;
    void process(const lcb_seed* sd, const uint32_t* view, int64_t n, std::vector<uint64_t>& off, std::vector<lcb_instance>& inst,
                 std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) override
    {
        std::vector<LcbKSeed> ks;
        for (int64_t i = 0; i < n; i++) {
            const uint32_t v = view ? view[i] : 0u;
            if ((int)v > emu->nViewsAlloc) throw LcbError("seed names a view that was not built");
            ks.push_back(LcbKSeed{sd[i].vid, sd[i].ch, v, 0u});
        }
        if (n) emu->runRetry(ks);
        off.assign((size_t)n + 1, 0); fpOff.assign((size_t)n + 1, 0); inst.clear(); fp.clear();
        for (int64_t i = 0; i < n; i++) {
            const LcbSeedOut& o = emu->out[(size_t)i];
            if (o.status) throw LcbError("emulated kernel overflow");
            off[(size_t)i] = inst.size(); fpOff[(size_t)i] = fp.size();
            for (uint32_t e = 0; e < o.nInst; e++) { const uint4 r = emu->arena[o.arenaOff + e]; inst.push_back(lcb_instance{r.x, r.y, r.z, r.w}); }
            for (uint32_t e = 0; e < o.nFp; e++) { const LcbFpOut r = emu->fpArena[o.fpOff + e]; fp.push_back(lcb_fp{emu->plan.toHost(r.lo), emu->plan.toHost(r.hi)}); }
        }
        off[(size_t)n] = inst.size(); fpOff[(size_t)n] = fp.size();
        if (ctrSink) {       // stats-mode kernels: the per-seed event counters (engine's countEvents)
            ctrSink->assign((size_t)n, lcb_counters{});
            for (int64_t i = 0; i < n && !getenv("EMU_NOSTATS"); i++) {
                const LcbSeedCtr& k = emu->octr[(size_t)i]; lcb_counters& q = (*ctrSink)[(size_t)i];
                q.n_walk = k.c[0]; q.n_occ = k.c[1]; q.n_compat_call = k.c[2]; q.n_compat_step = k.c[3]; q.n_inst_out = k.c[4]; q.n_vote = k.c[5]; q.n_push = k.c[6]; q.n_process = k.c[7];
            }
        }
    }
    void mark(const uint64_t* r, int64_t n) override
    {
        for (int64_t i = 0; i < n; i++) { uint64_t lo, hi; emu->plan.rangeToDev(r[2 * i], r[2 * i + 1], lo, hi); for (uint64_t q = lo; q < hi; q++) emu->used[q >> 5] |= 1u << (q & 31); }
    }
    void reset() override { emu->used.assign(emu->usedWords, 0u); emu->T.used = emu->used.data(); emu->nViewsAlloc = 0; emu->viewTab.assign(emu->nPages, 0); emu->T.viewTab = emu->viewTab.data(); }
    // begin / end (the engine plans a stop's speculative jobs while the results the stop needs are computed): the emulated launch
    // must see the state of the moment of the begin, so the live bitmap is snapshotted there and the seeds run against the snapshot at the end
    std::vector<lcb_seed> begunSeeds; std::vector<uint32_t> begunUsed; bool begunValid = false;
    bool processBegin(const lcb_seed* sd, int64_t n) override
    {
        if (getenv("EMU_NO_EARLY")) return false;   // (the engine then computes the stop's own jobs after the dry run)
        if (begunValid) return false;               // (one call in flight, like the device)
        begunSeeds.assign(sd, sd + n);
        begunUsed.assign(emu->used.begin(), emu->used.begin() + emu->usedWords);
        begunValid = true;
        return true;
    }
    void processEnd(std::vector<uint64_t>& off, std::vector<lcb_instance>& inst, std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) override
    {
        if (!begunValid) throw LcbError("processEnd without begin");
        begunValid = false;
        std::vector<uint32_t> now(emu->used.begin(), emu->used.begin() + emu->usedWords);
        std::copy(begunUsed.begin(), begunUsed.end(), emu->used.begin());
        process(begunSeeds.data(), nullptr, (int64_t)begunSeeds.size(), off, inst, fpOff, fp);
        std::copy(now.begin(), now.end(), emu->used.begin());
    }
    int maxViews() const override { return views; }
    int concurrency() const override { const char* e = getenv("EMU_CONCURRENCY"); return e ? atoi(e) : 16384; }
    // side lanes (EMU_SIDE_LANES=n): a background batch is computed on the spot and handed out job by job, EMU_SIDE_DELAY polls late
    // (EMU_SIDE_LATE=1: computed when first asked for, against the live state of that moment)
    LcbEagerSideLanes side{getenv("EMU_SIDE_LANES") ? atoi(getenv("EMU_SIDE_LANES")) : 0, getenv("EMU_SIDE_DELAY") ? atoi(getenv("EMU_SIDE_DELAY")) : 0, getenv("EMU_SIDE_LATE") != nullptr};
    int sideLanes() const override { return (int)side.lanes.size(); }
    int sideBegin(const lcb_seed* sd, const uint32_t* view, int64_t n, int nViews, const LcbViewMark* marks, int64_t nMarks) override
    {
        if (getenv("EMU_SIDE_CAP")) side.cap = atoll(getenv("EMU_SIDE_CAP"));      // larger batches are refused: their jobs run synchronously
        return side.begin(*this, sd, view, n, nViews, marks, nMarks);
    }
    int sidePoll(int lane, int64_t k, bool wait, std::vector<lcb_instance>& inst, std::vector<lcb_fp>& fp) override { return side.poll(*this, lane, k, wait, inst, fp); }
    void sideRelease(int lane) override { side.release(lane); }
    void buildViews(int nViews, const LcbViewMark* marks, int64_t nMarks) override { emu->buildViews(nViews, marks, nMarks); }
};

// In-process all-gather between the rank threads of the `find-ranks` mode (stands in for ncclAllGather / gloo).
struct EmuExchange {
    std::mutex m; std::condition_variable cv;
    int world = 1, arrived = 0; uint64_t gen = 0;
    std::vector<unsigned char> buf;
    void barrier(std::unique_lock<std::mutex>& lk) { const uint64_t g = gen; if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g; }); }
};
struct EmuRankLink { EmuExchange* ex; int rank; };
int emuAllgather(void* user, const void* send, uint64_t bytes, void* recv)
{
    EmuRankLink* l = (EmuRankLink*)user;
    std::unique_lock<std::mutex> lk(l->ex->m);
    if (l->ex->buf.size() < bytes * l->ex->world) l->ex->buf.resize(bytes * l->ex->world);
    l->ex->barrier(lk);                                   // everyone has sized the buffer
    memcpy(l->ex->buf.data() + bytes * l->rank, send, bytes);
    l->ex->barrier(lk);                                   // everyone has written
    memcpy(recv, l->ex->buf.data(), bytes * l->ex->world);
    l->ex->barrier(lk);                                   // everyone has read
    return 0;
}

int compareSeed(int64_t idx, const lcb_seed& sd, const LcbSeedOut& o, const uint4* arena, const orc_inst* ref, int64_t nRef, int64_t refScore)
{
    bool ok = o.status == 0 && (int64_t)o.nInst == nRef && o.bestScore == refScore;
    for (int64_t i = 0; ok && i < nRef; i++) {
        const uint4 r = arena[o.arenaOff + i];
        ok = r.x == ref[i].chr && r.y == ref[i].front_idx && r.z == ref[i].back_idx && (r.w != 0) == (ref[i].positive != 0);
    }
    if (!ok) {
        fprintf(stderr, "MISMATCH seed %lld vid=%d ch=%c: kernel status=%u n=%u score=%lld | oracle n=%lld score=%lld\n", (long long)idx, sd.vid,
                (char)sd.ch, o.status, o.nInst, (long long)o.bestScore, (long long)nRef, (long long)refScore);
        for (int64_t i = 0; i < nRef || i < (int64_t)o.nInst; i++) {
            if (i < (int64_t)o.nInst && o.status == 0) { const uint4 r = arena[o.arenaOff + i]; fprintf(stderr, "   K %c,%u,%u,%u", r.w ? '+' : '-', r.x, r.y, r.z); }
            else fprintf(stderr, "   K -");
            if (i < nRef) fprintf(stderr, "   | O %c,%u,%u,%u\n", ref[i].positive ? '+' : '-', ref[i].chr, ref[i].front_idx, ref[i].back_idx);
            else fprintf(stderr, "   | O -\n");
            if (i > 12) break;
        }
    }
    return ok ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 8) { fprintf(stderr, "usage: emu_check graph fasta k b m a mode [outdir]\n"); return 2; }
    const char* graph = argv[1]; const char* fasta = argv[2];
    lcb_params p; p.k = atoi(argv[3]); p.max_branch = p.max_flank = atoi(argv[4]); p.min_block = atoi(argv[5]); p.looking_depth = 8; p.phase_size = 256;
    const int a = atoi(argv[6]);
    const std::string mode = argv[7];
    const std::string outDir = argc > 8 ? argv[8] : "/tmp/emu_out";
    try {
        lcb_graph* g = lcb_graph_load_impl(graph, {fasta}, p.k, a, 4);
        std::vector<lcb_seed> seeds;
        lcb_enumerate_seeds_impl(*g, 4, seeds);
        char err[512];
        const char* fa[1] = {fasta};
        orc_graph* og = orc_load(graph, fa, 1, p.k, a, err, sizeof(err));
        if (!og) { fprintf(stderr, "oracle load: %s\n", err); return 1; }
        orc_params op{p.k, p.min_block, p.max_branch, p.max_flank, p.looking_depth};
        const int64_t S = orc_build_bundles(og);
        int bad = 0;
        if (S != (int64_t)seeds.size()) { fprintf(stderr, "seed count differs: %lld vs %zu\n", (long long)S, seeds.size()); bad++; }
        for (int64_t i = 0; i < S && i < (int64_t)seeds.size(); i++) {
            int64_t vid; int32_t ch; uint64_t cnt, rank, rp, rc;
            orc_get_bundle(og, i, &vid, &ch, &cnt, &rank, &rp, &rc);
            const lcb_seed& s = seeds[i];
            if (s.vid != vid || s.ch != ch || s.count != cnt || s.rank != rank || s.resolve_pos != rp || s.resolve_chr != rc) {
                if (bad < 5) fprintf(stderr, "seed %lld differs: product (%d,%d,%llu) oracle (%lld,%d,%llu)\n", (long long)i, s.vid, s.ch,
                                     (unsigned long long)s.count, (long long)vid, ch, (unsigned long long)cnt);
                bad++;
            }
        }
        if (bad) { fprintf(stderr, "FAIL: seed lists differ (%d)\n", bad); return 1; }
        // table parity (flat SoA vs the oracle's per-chromosome arrays)
        for (int64_t c = 0; c < orc_n_chr(og); c++) {
            std::vector<int32_t> id(orc_chr_n_pos(og, c)); std::vector<uint32_t> pos(id.size());
            orc_chr_positions(og, c, id.data(), pos.data());
            if ((int64_t)(g->chrStart[c + 1] - g->chrStart[c]) != (int64_t)id.size()) { fprintf(stderr, "FAIL: chr %lld size\n", (long long)c); return 1; }
            for (size_t i = 0; i < id.size(); i++)
                if (g->posId[g->chrStart[c] + i] != id[i] || g->posPos[g->chrStart[c] + i] != pos[i]) { fprintf(stderr, "FAIL: table differs\n"); return 1; }
        }
        Emu emu(g, p, mode == "huge" ? 3 : (mode == "big" ? 2 : (mode == "medium" ? 1 : 0)));
        if (mode == "seeds-init" || mode == "seeds-final" || mode == "big" || mode == "medium" || mode == "huge") {
            orc_counters octr; memset(&octr, 0, sizeof(octr));
            if (mode == "seeds-final") {
                orc_block* ob = nullptr; orc_stats st;
                orc_find_blocks(og, &op, &ob, &st, nullptr);
                orc_free_blocks(ob);
                const size_t stride = orc_used_stride();
                for (int64_t c = 0; c < orc_n_chr(og); c++) {
                    const uint8_t* u = orc_chr_used(og, c);
                    for (int64_t i = 0; i < orc_chr_n_pos(og, c); i++)
                        if (u[(size_t)i * stride]) { const uint64_t q = emu.plan.toDev(g->chrStart[c] + i); emu.used[q >> 5] |= 1u << (q & 31); }
                }
            }
            if (getenv("EMU_ONLY")) { const lcb_seed one = seeds[atoi(getenv("EMU_ONLY"))]; seeds.assign(1, one); }
            if (getenv("EMU_LIMIT") && (size_t)atoi(getenv("EMU_LIMIT")) < seeds.size()) seeds.resize((size_t)atoi(getenv("EMU_LIMIT")));   // the heavy seeds come first
            std::vector<LcbKSeed> ks;
            for (auto& s : seeds) ks.push_back(LcbKSeed{s.vid, s.ch, 0u, 0u});
            emu.runRetry(ks);
            std::vector<orc_inst> ref(1 << 16);
            for (size_t i = 0; i < seeds.size(); i++) {
                int64_t score = 0;
                const int64_t n = orc_process_seed(og, &op, seeds[i].vid, seeds[i].ch, ref.data(), (int64_t)ref.size(), &score, &octr);
                bad += compareSeed((int64_t)i, seeds[i], emu.out[i], emu.arena.data(), ref.data(), n, score);
                if (getenv("EMU_FP_CHECK") && emu.out[i].status == 0) {
                    // Completeness of the footprint (rule (2) of the engine): with EVERY unused position outside the kernel's
                    // footprint intervals set to used, Process() must still give this result - it never read those bits as 0.
                    const LcbSeedOut& o = emu.out[i];
                    std::vector<uint8_t> keep((size_t)g->nPos(), 0);
                    for (uint32_t e = 0; e < o.nFp; e++) { const LcbFpOut f = emu.fpArena[o.fpOff + e]; for (uint64_t q = emu.plan.toHost(f.lo); q <= emu.plan.toHost(f.hi) && q < keep.size(); q++) keep[q] = 1; }
                    const size_t stride = orc_used_stride();
                    std::vector<uint8_t*> flipped;
                    for (int64_t c = 0; c < orc_n_chr(og); c++) {
                        uint8_t* u = orc_chr_used(og, c);
                        const uint64_t base = g->chrStart[c];
                        for (int64_t q = 0; q < orc_chr_n_pos(og, c); q++) if (!keep[(size_t)(base + q)] && !u[(size_t)q * stride]) { u[(size_t)q * stride] = 1; flipped.push_back(&u[(size_t)q * stride]); }
                    }
                    int64_t score2 = 0;
                    const int64_t n2 = orc_process_seed(og, &op, seeds[i].vid, seeds[i].ch, ref.data(), (int64_t)ref.size(), &score2, nullptr);
                    for (uint8_t* u : flipped) *u = 0;
                    if (compareSeed((int64_t)i, seeds[i], o, emu.arena.data(), ref.data(), n2, score2)) { fprintf(stderr, "FAIL: the footprint of seed %zu does not cover what it read (%zu positions outside it were set)\n", i, flipped.size()); bad++; }
                }
                {
                    static orc_counters prev; static int shown = 0;
                    const uint64_t dc = octr.n_compat_call - prev.n_compat_call, ds = octr.n_compat_step - prev.n_compat_step;
                    if ((dc != emu.octr[i].c[2] || ds != emu.octr[i].c[3]) && shown < 5 && getenv("EMU_CTR_DEBUG")) {
                        shown++;
                        fprintf(stderr, "ctr seed %zu vid=%d ch=%c: kernel ccall=%llu cstep=%llu | oracle ccall=%llu cstep=%llu (occ k=%llu o=%llu)\n", i, seeds[i].vid,
                                (char)seeds[i].ch, (unsigned long long)emu.octr[i].c[2], (unsigned long long)emu.octr[i].c[3], (unsigned long long)dc,
                                (unsigned long long)ds, (unsigned long long)emu.octr[i].c[1], (unsigned long long)(octr.n_occ - prev.n_occ));
                    }
                    prev = octr;
                }
                if (bad > 8) break;
            }
            fprintf(stderr, "%s: %zu seeds, %d mismatches, %llu cross-lane ops\n", mode.c_str(), seeds.size(), bad, (unsigned long long)emu_collective_count());
            fprintf(stderr, "counters kernel: walk=%llu occ=%llu ccall=%llu cstep=%llu inst=%llu vote=%llu push=%llu\n", (unsigned long long)emu.ctr.n_walk,
                    (unsigned long long)emu.ctr.n_occ, (unsigned long long)emu.ctr.n_compat_call, (unsigned long long)emu.ctr.n_compat_step,
                    (unsigned long long)emu.ctr.n_inst_out, (unsigned long long)emu.ctr.n_vote, (unsigned long long)emu.ctr.n_push);
            fprintf(stderr, "counters oracle: walk=%llu occ=%llu ccall=%llu cstep=%llu inst=%llu vote=%llu push=%llu\n", (unsigned long long)octr.n_walk,
                    (unsigned long long)octr.n_occ, (unsigned long long)octr.n_compat_call, (unsigned long long)octr.n_compat_step,
                    (unsigned long long)octr.n_inst_out, (unsigned long long)octr.n_vote, (unsigned long long)octr.n_push);
            if (!bad && !getenv("EMU_NOSTATS") && (emu.ctr.n_walk != octr.n_walk || emu.ctr.n_occ != octr.n_occ || emu.ctr.n_compat_call != octr.n_compat_call ||
                         emu.ctr.n_compat_step != octr.n_compat_step || emu.ctr.n_inst_out != octr.n_inst_out)) { fprintf(stderr, "FAIL: event counters differ\n"); bad++; }
        } else if (mode == "find") {
            // the product's speculative round engine (engine.cpp) over the emulated kernel, for several round sizes
            orc_block* ob = nullptr; orc_stats st;
            orc_counters fctr; memset(&fctr, 0, sizeof(fctr));
            const int64_t nb = orc_find_blocks(og, &op, &ob, &st, &fctr);
            std::vector<lcb_block> blocks;
            const char* rp = getenv("EMU_ROUNDS");
            std::vector<int> rounds = rp ? std::vector<int>{atoi(rp)} : std::vector<int>{1, 3, 64};
            const char* vp = getenv("EMU_VIEWS");
            for (int R : rounds) {
                EmuProcessor proc; proc.emu = &emu;
                proc.views = vp ? atoi(vp) : (R == 3 ? 0 : (R == 1 ? 2 : 64));   // no views / view starvation / plenty
                emu.launches = emu.criticalPushes = emu.totalPushes = emu.firstPushes = 0;
                LcbEngineConfig cfg; cfg.roundPhases = R;
                auto envInt = [](const char* n) { const char* e = getenv(n); return e && *e ? atoi(e) : 0; };
                cfg.roundFixed = envInt("LCB_ROUND_FIXED") != 0; cfg.maxJobs = envInt("LCB_MAX_JOBS");
                if (getenv("LCB_PREDICT_F")) cfg.predictF = std::max(1, envInt("LCB_PREDICT_F"));
                if (getenv("LCB_EAGER_PHASES")) cfg.eagerPhases = envInt("LCB_EAGER_PHASES") ? envInt("LCB_EAGER_PHASES") : -1;
                if (getenv("LCB_LAZY_SPAN")) cfg.lazySpan = envInt("LCB_LAZY_SPAN") ? envInt("LCB_LAZY_SPAN") : -1;
                if (getenv("LCB_SPARSE_ROUNDS")) cfg.sparseRounds = atoi(getenv("LCB_SPARSE_ROUNDS"));   // -1 / 0 / 1 (lcb_hooks.sparse_rounds)
                cfg.countEvents = !getenv("EMU_NOSTATS");   // stats-mode kernels: the engine sums the events of exactly the reference's Process() calls (host commit only)
                LcbEngineStats es;
                lcb_engine_run(g, &p, seeds.data(), (int64_t)seeds.size(), proc, cfg, blocks, &es);
                int diffs = 0;
                if (nb != (int64_t)blocks.size() || st.blocks_found != es.blocksFound || st.failures != es.failures) diffs++;
                for (int64_t i = 0; i < nb && i < (int64_t)blocks.size(); i++)
                    if (ob[i].id != blocks[i].id || ob[i].chr != blocks[i].chr || ob[i].start != blocks[i].start || ob[i].end != blocks[i].end) diffs++;
                fprintf(stderr, "find R=%d: %zu seeds, blocks %zu/%lld found %lld/%lld failures %lld/%lld rounds %lld recompute %lld launches (%lld seeds) conflict %lld launches (%lld seeds) diffs %d\n",
                        R, seeds.size(), blocks.size(), (long long)nb, (long long)es.blocksFound, (long long)st.blocks_found, (long long)es.failures,
                        (long long)st.failures, (long long)es.rounds, (long long)es.recomputeLaunches, (long long)es.recomputedSeeds,
                        (long long)es.conflictLaunches, (long long)es.conflictSeeds, diffs);
                fprintf(stderr, "       side lanes: %lld batches, %lld jobs, %lld taken | early critical launches %lld | lazy seeds %lld\n",
                        (long long)es.sideBatches, (long long)es.sideJobs, (long long)es.sideTaken, (long long)es.earlyCritical, (long long)es.lazySeeds);
                if (getenv("EMU_EXPECT_EARLY") && es.recomputeLaunches > 0 && es.earlyCritical == 0) { fprintf(stderr, "early critical launches expected but none happened\n"); return 1; }
                fprintf(stderr, "       views %d: built %lld, job results used %lld | launches %llu, critical path %llu pushes (first jobs %llu), total %llu pushes\n", proc.views,
                        (long long)es.viewsBuilt, (long long)es.jobsUsed, (unsigned long long)emu.launches, (unsigned long long)emu.criticalPushes,
                        (unsigned long long)emu.firstPushes, (unsigned long long)emu.totalPushes);
                if (cfg.countEvents) {
                    const lcb_counters& e = es.events;
                    // n_compat_step counts the steps of Compatible's `used` walk up to the first used bit; a speculative result that is
                    // exact (no bit of its footprint changed) may still have walked an incompatible gap against an older bitmap, where
                    // the first used bit sits further away: same outcome, a few more steps. Everything else is equal by construction.
                    const double dstep = (double)e.n_compat_step - (double)fctr.n_compat_step;
                    const bool same = e.n_walk == fctr.n_walk && e.n_occ == fctr.n_occ && e.n_compat_call == fctr.n_compat_call &&
                                      dstep >= 0 && dstep <= 1e-2 * (double)fctr.n_compat_step + 8 &&
                                      e.n_inst_out == fctr.n_inst_out && e.n_vote == fctr.n_vote && e.n_push == fctr.n_push && e.n_process == fctr.n_process;
                    fprintf(stderr, "       events walk=%llu occ=%llu ccall=%llu cstep=%llu inst=%llu process=%llu | oracle walk=%llu occ=%llu ccall=%llu cstep=%llu inst=%llu process=%llu %s\n",
                            (unsigned long long)e.n_walk, (unsigned long long)e.n_occ, (unsigned long long)e.n_compat_call, (unsigned long long)e.n_compat_step, (unsigned long long)e.n_inst_out,
                            (unsigned long long)e.n_process, (unsigned long long)fctr.n_walk, (unsigned long long)fctr.n_occ, (unsigned long long)fctr.n_compat_call,
                            (unsigned long long)fctr.n_compat_step, (unsigned long long)fctr.n_inst_out, (unsigned long long)fctr.n_process, same ? "equal" : "DIFFERENT");
                    if (!same) { fprintf(stderr, "FAIL: the engine's event totals differ from the oracle's FindBlocks\n"); diffs++; }
                }
                bad += diffs;
            }
            int64_t nTrim = 0; double cov = 0;
            lcb_generate_output_impl(*g, p.min_block, blocks.data(), (int64_t)blocks.size(), st.blocks_found, outDir, false, 0, &nTrim, &cov);
            double ocov = 0;
            const std::string od2 = outDir + "_oracle";
            const int64_t ont = orc_generate_output(og, p.min_block, ob, nb, st.blocks_found, od2.c_str(), &ocov, err, sizeof(err));
            if (ont != nTrim) { fprintf(stderr, "FAIL: trimmed %lld vs %lld\n", (long long)nTrim, (long long)ont); bad++; }
            printf("Blocks found: %lld\nCoverage: %.2f\n", (long long)nTrim, cov);
        } else if (mode == "find-ranks") {
            // The multi-rank engine with REAL footprints and predicted views: EMU_RANKS rank threads, each with its own emulated
            // device (its own `used` views), launches dealt to the ranks, results all-gathered in process, identical commit.
            orc_block* ob = nullptr; orc_stats st;
            const int64_t nb = orc_find_blocks(og, &op, &ob, &st, nullptr);
            const int world = getenv("EMU_RANKS") ? atoi(getenv("EMU_RANKS")) : 2;
            const int R = getenv("EMU_ROUNDS") ? atoi(getenv("EMU_ROUNDS")) : 64;
            EmuExchange ex; ex.world = world;
            std::vector<std::vector<lcb_block>> blocksOf((size_t)world);
            std::vector<LcbEngineStats> statsOf((size_t)world);
            std::vector<std::string> errOf((size_t)world);
            std::vector<std::unique_ptr<Emu>> emus;
            for (int r = 0; r < world; r++) emus.emplace_back(new Emu(g, p, 0));
            std::vector<std::thread> th;
            for (int r = 0; r < world; r++)
                th.emplace_back([&, r]() {
                    try {
                        EmuProcessor proc; proc.emu = emus[(size_t)r].get();
                        proc.views = getenv("EMU_VIEWS") ? atoi(getenv("EMU_VIEWS")) : 64;
                        EmuRankLink link{&ex, r};
                        LcbEngineConfig cfg; cfg.roundPhases = R; cfg.rank = r; cfg.world = world;
                        if (getenv("LCB_LAZY_SPAN")) cfg.lazySpan = atoi(getenv("LCB_LAZY_SPAN")) ? atoi(getenv("LCB_LAZY_SPAN")) : -1;
                        if (getenv("LCB_SPARSE_ROUNDS")) cfg.sparseRounds = atoi(getenv("LCB_SPARSE_ROUNDS"));   // -1 / 0 / 1 (lcb_hooks.sparse_rounds)
                        cfg.allgather = emuAllgather; cfg.allgatherUser = &link;
                        lcb_engine_run(g, &p, seeds.data(), (int64_t)seeds.size(), proc, cfg, blocksOf[(size_t)r], &statsOf[(size_t)r]);
                    } catch (std::exception& e) { errOf[(size_t)r] = e.what(); }
                });
            for (auto& t : th) t.join();
            for (int r = 0; r < world; r++) {
                if (!errOf[(size_t)r].empty()) { fprintf(stderr, "rank %d: error: %s\n", r, errOf[(size_t)r].c_str()); bad++; continue; }
                const auto& blocks = blocksOf[(size_t)r]; const auto& es = statsOf[(size_t)r];
                int diffs = 0;
                if (nb != (int64_t)blocks.size() || st.blocks_found != es.blocksFound || st.failures != es.failures) diffs++;
                for (int64_t i = 0; i < nb && i < (int64_t)blocks.size(); i++)
                    if (ob[i].id != blocks[i].id || ob[i].chr != blocks[i].chr || ob[i].start != blocks[i].start || ob[i].end != blocks[i].end) diffs++;
                fprintf(stderr, "find-ranks rank %d/%d: blocks %zu/%lld failures %lld/%lld rounds %lld job launches %lld (%lld jobs, %lld used) views %lld exchanges %lld (collectives %lld) | side lanes: %lld batches, %lld jobs, %lld taken, %lld void, %lld failed | diffs %d\n", r, world,
                        blocks.size(), (long long)nb, (long long)es.failures, (long long)st.failures, (long long)es.rounds, (long long)es.recomputeLaunches,
                        (long long)es.recomputedSeeds, (long long)es.jobsUsed, (long long)es.viewsBuilt, (long long)es.exchanges, (long long)es.collectives, (long long)es.sideBatches, (long long)es.sideJobs,
                        (long long)es.sideTaken, (long long)es.sideVoid, (long long)es.sideFailed, diffs);
                if (es.exchanges == 0) { fprintf(stderr, "FAIL: no exchange happened\n"); bad++; }
                if (getenv("EMU_SIDE_LANES") && es.recomputeLaunches > 2 && es.sideBatches == 0) { fprintf(stderr, "FAIL: side lanes asked for but no batch ran in the background\n"); bad++; }
                bad += diffs;
            }
        } else { fprintf(stderr, "unknown mode\n"); return 2; }
        return bad ? 1 : 0;
    } catch (std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
