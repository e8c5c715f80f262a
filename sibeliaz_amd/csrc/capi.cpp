// capi.cpp — the extern "C" surface declared in include/lcb.h: argument checks, exception -> return
// code translation, thread-local error text. No logic lives here.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lcb_device.h"
#include "lcb_host.h"

namespace {
thread_local std::string g_error;
}

void lcb_set_error(const std::string& msg) { g_error = msg; }

#define LCB_TRY try {
#define LCB_CATCH(ret)                                          \
    }                                                           \
    catch (std::exception & e) { g_error = e.what(); return ret; } \
    catch (...) { g_error = "unknown error"; return ret; }

namespace {
// The structs of the ABI are allocated by the caller: one built against another header would be misread field by field.
void checkAbi(const lcb_hooks* hooks, const lcb_device_opts* opts)
{
    if (hooks && hooks->abi != LCB_ABI_VERSION) throw LcbError("lcb_hooks.abi is " + std::to_string(hooks->abi) + ", this library has LCB_ABI_VERSION " + std::to_string(LCB_ABI_VERSION));
    if (opts && opts->abi != (uint32_t)LCB_ABI_VERSION) throw LcbError("lcb_device_opts.abi is " + std::to_string(opts->abi) + ", this library has LCB_ABI_VERSION " + std::to_string(LCB_ABI_VERSION));
}

LcbEngineConfig tuningOf(const lcb_hooks* hooks)
{
    checkAbi(hooks, nullptr);
    LcbEngineConfig cfg;
    if (hooks) {
        cfg.roundPhases = hooks->round_phases; cfg.progress = hooks->progress != 0;
        cfg.roundFixed = hooks->round_fixed != 0; cfg.eagerPhases = hooks->eager_phases; cfg.maxViews = hooks->max_views;
        cfg.maxJobs = hooks->max_jobs; cfg.predictF = hooks->predict_f; cfg.exchangeAlways = hooks->exchange_always != 0; cfg.countEvents = hooks->count_events != 0; cfg.syncJobs = hooks->sync_jobs != 0; cfg.lazySpan = hooks->lazy_span; cfg.sparseRounds = hooks->sparse_rounds;
    }
    return cfg;
}
int giveBlocks(std::vector<lcb_block>& v, lcb_block** blocks, int64_t* n_blocks)
{
    *blocks = (lcb_block*)malloc((v.size() ? v.size() : 1) * sizeof(lcb_block));
    if (!*blocks) throw LcbError("out of memory");
    if (!v.empty()) memcpy(*blocks, v.data(), v.size() * sizeof(lcb_block));
    *n_blocks = (int64_t)v.size();
    return LCB_OK;
}
}  // namespace

#define LCB_NEED(cond, what) do { if (!(cond)) throw LcbError(std::string(what) + ": null argument"); } while (0)

extern "C" {

const char* lcb_last_error(void) { return g_error.c_str(); }
const char* lcb_version(void) { return "sibeliaz_amd 0.4 (gfx950)"; }
int lcb_abi_version(void) { return LCB_ABI_VERSION; }
void lcb_free(void* p) { free(p); }

lcb_graph* lcb_graph_load(const char* junction_file, const char* const* fasta_files, int n_fasta, int k, int abundance, int threads)
{
    LCB_TRY
    if (!junction_file || !fasta_files || n_fasta <= 0) throw LcbError("lcb_graph_load: missing input files");
    if (k <= 0 || (k % 2) == 0) throw LcbError("value of K must be odd");
    std::vector<std::string> fa(fasta_files, fasta_files + n_fasta);
    return lcb_graph_load_impl(junction_file, fa, k, abundance, threads);
    LCB_CATCH(nullptr)
}
void lcb_graph_free(lcb_graph* g) { delete g; }
int64_t lcb_graph_n_chr(const lcb_graph* g) { return g->nChr(); }
int64_t lcb_graph_n_pos(const lcb_graph* g) { return (int64_t)g->nPos(); }
int64_t lcb_graph_n_vertices(const lcb_graph* g) { return g->nVertex; }
int64_t lcb_graph_chr_len(const lcb_graph* g, int64_t chr) { return (int64_t)g->seq[(size_t)chr].size(); }
int64_t lcb_graph_chr_n_pos(const lcb_graph* g, int64_t chr) { return (int64_t)(g->chrStart[(size_t)chr + 1] - g->chrStart[(size_t)chr]); }
const char* lcb_graph_chr_name(const lcb_graph* g, int64_t chr) { return g->chrName[(size_t)chr].c_str(); }
const uint64_t* lcb_graph_chr_start(const lcb_graph* g) { return g->chrStart.data(); }
const int32_t* lcb_graph_pos_id(const lcb_graph* g) { return g->posId.data(); }
const uint32_t* lcb_graph_pos_pos(const lcb_graph* g) { return g->posPos.data(); }

int64_t lcb_enumerate_seeds(const lcb_graph* g, int threads, lcb_seed** out)
{
    LCB_TRY
    LCB_NEED(g && out, "lcb_enumerate_seeds");
    std::vector<lcb_seed> v;
    lcb_enumerate_seeds_impl(*g, threads, v);
    *out = (lcb_seed*)malloc((v.size() ? v.size() : 1) * sizeof(lcb_seed));
    if (!*out) throw LcbError("out of memory");
    if (!v.empty()) memcpy(*out, v.data(), v.size() * sizeof(lcb_seed));
    return (int64_t)v.size();
    LCB_CATCH(-1)
}

lcb_device* lcb_device_create(const lcb_graph* g, const lcb_params* p, int device_ordinal)
{
    LCB_TRY
    if (!g || !p) throw LcbError("lcb_device_create: null argument");
    return lcb_device_create_impl(g, p, device_ordinal);
    LCB_CATCH(nullptr)
}
lcb_device* lcb_device_create_ex(const lcb_graph* g, const lcb_params* p, int device_ordinal, const lcb_device_opts* opts)
{
    LCB_TRY
    if (!g || !p) throw LcbError("lcb_device_create_ex: null argument");
    checkAbi(nullptr, opts);
    return lcb_device_create_impl(g, p, device_ordinal, opts);
    LCB_CATCH(nullptr)
}
int lcb_device_mode_seeds(lcb_device* d, int64_t counts[4])
{
    LCB_TRY
    if (!d || !counts) throw LcbError("lcb_device_mode_seeds: null argument");
    lcb_device_mode_seeds_impl(d, counts);
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}
int lcb_device_mode_time(lcb_device* d, double ms[4], int64_t launches[4])
{
    LCB_TRY
    if (!d || !ms || !launches) throw LcbError("lcb_device_mode_time: null argument");
    lcb_device_mode_time_impl(d, ms, launches);
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}
void lcb_device_destroy(lcb_device* d) { lcb_device_destroy_impl(d); }
int lcb_device_reset_used(lcb_device* d) { LCB_TRY LCB_NEED(d, "lcb_device_reset_used"); lcb_device_reset_used_impl(d); return LCB_OK; LCB_CATCH(LCB_ERR) }
int lcb_device_mark_used(lcb_device* d, const uint64_t* ranges, int64_t n) { LCB_TRY LCB_NEED(d && (ranges || n == 0), "lcb_device_mark_used"); lcb_device_mark_used_impl(d, ranges, n); return LCB_OK; LCB_CATCH(LCB_ERR) }
int lcb_device_set_used(lcb_device* d, const uint32_t* words, int64_t n_words) { LCB_TRY LCB_NEED(d && words, "lcb_device_set_used"); lcb_device_set_used_impl(d, words, n_words); return LCB_OK; LCB_CATCH(LCB_ERR) }
int lcb_device_set_stats_mode(lcb_device* d, int on) { LCB_TRY LCB_NEED(d, "lcb_device_set_stats_mode"); lcb_device_set_stats_impl(d, on != 0); return LCB_OK; LCB_CATCH(LCB_ERR) }
int lcb_device_hbm_triad(lcb_device* d, uint64_t bytes, int reps, double* gb_per_s)
{
    LCB_TRY
    LCB_NEED(d && gb_per_s, "lcb_device_hbm_triad");
    *gb_per_s = lcb_device_hbm_triad_impl(d, bytes, reps > 0 ? reps : 1);
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}
int lcb_device_kernel_time(lcb_device* d, double* ms, int64_t* launches) { LCB_TRY LCB_NEED(d, "lcb_device_kernel_time"); lcb_device_kernel_time_impl(d, ms, launches); return LCB_OK; LCB_CATCH(LCB_ERR) }

int lcb_process_seeds(lcb_device* d, const lcb_seed* seeds, int64_t n, uint64_t* offsets, lcb_instance* inst, uint64_t inst_cap,
                      int64_t* best_score, lcb_counters* ctr)
{
    LCB_TRY
    LCB_NEED(d && (seeds || n == 0) && offsets && (inst || inst_cap == 0), "lcb_process_seeds");
    std::vector<uint64_t> off;
    std::vector<lcb_instance> res;
    lcb_device_process_impl(d, seeds, n, off, res, best_score, ctr);
    memcpy(offsets, off.data(), off.size() * sizeof(uint64_t));
    if (res.size() > inst_cap) throw LcbError("lcb_process_seeds: inst_cap too small (needed count is in offsets[n])");
    if (!res.empty()) memcpy(inst, res.data(), res.size() * sizeof(lcb_instance));
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}

int lcb_process_seeds_fp(lcb_device* d, const lcb_seed* seeds, int64_t n, uint64_t* offsets, lcb_instance* inst, uint64_t inst_cap,
                         uint64_t* fp_offsets, uint64_t* fp, uint64_t fp_cap)
{
    LCB_TRY
    LCB_NEED(d && (seeds || n == 0) && offsets && fp_offsets && (inst || inst_cap == 0) && (fp || fp_cap == 0), "lcb_process_seeds_fp");
    std::vector<uint64_t> off, fpOff;
    std::vector<lcb_instance> res;
    std::vector<lcb_fp> fps;
    lcb_device_process_impl(d, seeds, n, off, res, nullptr, nullptr, &fpOff, &fps);
    memcpy(offsets, off.data(), off.size() * sizeof(uint64_t));
    memcpy(fp_offsets, fpOff.data(), fpOff.size() * sizeof(uint64_t));
    if (res.size() > inst_cap || fps.size() > fp_cap) throw LcbError("lcb_process_seeds_fp: inst_cap / fp_cap too small (needed counts are in offsets[n] / fp_offsets[n])");
    if (!res.empty()) memcpy(inst, res.data(), res.size() * sizeof(lcb_instance));
    static_assert(sizeof(lcb_fp) == 16, "footprint intervals are pairs of uint64");
    if (!fps.empty()) memcpy(fp, fps.data(), fps.size() * sizeof(lcb_fp));
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}

lcb_committer* lcb_committer_create(const lcb_graph* g, const lcb_params* p)
{
    LCB_TRY
    return new lcb_committer(g, *p);
    LCB_CATCH(nullptr)
}
void lcb_committer_free(lcb_committer* c) { delete c; }
int lcb_committer_commit_phase(lcb_committer* c, const lcb_seed* seeds, int64_t n, const uint64_t* offsets, const lcb_instance* inst,
                               lcb_reprocess_fn fn, void* user)
{
    LCB_TRY
    c->commitPhase(seeds, n, offsets, inst, fn, user);
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}
int64_t lcb_committer_take_marks(lcb_committer* c, uint64_t* ranges, int64_t cap)
{
    const int64_t have = (int64_t)(c->marks.size() / 2);
    const int64_t n = have < cap ? have : cap;
    if (n > 0) memcpy(ranges, c->marks.data(), (size_t)n * 2 * sizeof(uint64_t));
    c->marks.erase(c->marks.begin(), c->marks.begin() + 2 * n);
    return n;
}
int64_t lcb_committer_n_blocks(const lcb_committer* c) { return (int64_t)c->blocks.size(); }
const lcb_block* lcb_committer_blocks(const lcb_committer* c) { return c->blocks.data(); }
int64_t lcb_committer_blocks_found(const lcb_committer* c) { return c->blocksFound; }
int64_t lcb_committer_failures(const lcb_committer* c) { return c->failures; }
const uint32_t* lcb_committer_used_words(const lcb_committer* c, int64_t* n_words)
{
    if (n_words) *n_words = (int64_t)c->used.size();
    return c->used.data();
}

namespace {
// A per-seed engine supplied by the caller through C callbacks (no footprints: one all-covering interval per seed, so
// every speculative result is conservatively re-computed after any commit).
struct CallbackProcessor : LcbProcessor {
    const lcb_hooks* h;
    explicit CallbackProcessor(const lcb_hooks* hooks) : h(hooks) {}
    void process(const lcb_seed* seeds, const uint32_t*, int64_t n, std::vector<uint64_t>& off, std::vector<lcb_instance>& inst,
                 std::vector<uint64_t>& fpOff, std::vector<lcb_fp>& fp) override      // no predicted views: maxViews() == 0
    {
        off.assign((size_t)n + 1, 0);
        uint64_t cap = inst.size() < 4096 ? 4096 : inst.size();
        for (;;) {
            inst.resize(cap);
            const int rc = n ? h->process(h->engine_user, seeds, n, off.data(), inst.data(), cap) : 0;
            if (rc == 0) break;
            if (rc == 1 && off[(size_t)n] > cap) { cap = off[(size_t)n]; continue; }
            throw LcbError("process callback failed");
        }
        inst.resize(off[(size_t)n]);
        fpOff.resize((size_t)n + 1);
        fp.assign((size_t)n, lcb_fp{0u, UINT64_MAX});
        for (int64_t i = 0; i <= n; i++) fpOff[(size_t)i] = (uint64_t)i;
    }
    void mark(const uint64_t* ranges, int64_t n) override { if (h->mark(h->engine_user, ranges, n)) throw LcbError("mark callback failed"); }
    void reset() override { if (h->reset(h->engine_user)) throw LcbError("reset callback failed"); }
};
}  // namespace

int lcb_find_blocks_ex(const lcb_graph* g, lcb_device* d, const lcb_params* p, const lcb_seed* seeds, int64_t n_seeds,
                       const lcb_hooks* hooks, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats)
{
    LCB_TRY
    LCB_NEED(g && p && (seeds || n_seeds == 0) && blocks && n_blocks, "lcb_find_blocks_ex");
    LcbEngineConfig cfg;
    checkAbi(hooks, nullptr);
    if (hooks) {
        cfg.rank = hooks->rank; cfg.world = hooks->world > 0 ? hooks->world : 1;
        cfg.allgather = hooks->allgather; cfg.allgatherUser = hooks->allgather_user;
        cfg.roundPhases = hooks->round_phases; cfg.progress = hooks->progress != 0;
        cfg.roundFixed = hooks->round_fixed != 0; cfg.eagerPhases = hooks->eager_phases; cfg.maxViews = hooks->max_views;
        cfg.maxJobs = hooks->max_jobs; cfg.predictF = hooks->predict_f; cfg.exchangeAlways = hooks->exchange_always != 0; cfg.countEvents = hooks->count_events != 0; cfg.syncJobs = hooks->sync_jobs != 0; cfg.lazySpan = hooks->lazy_span; cfg.sparseRounds = hooks->sparse_rounds;
    }
    std::vector<lcb_block> v;
    if (d) lcb_find_blocks_impl(g, d, p, seeds, n_seeds, cfg, v, stats);
    else {
        if (!hooks || !hooks->process || !hooks->mark || !hooks->reset) throw LcbError("lcb_find_blocks_ex: no device and no engine callbacks");
        CallbackProcessor proc(hooks);
        LcbEngineStats es;
        lcb_engine_run(g, p, seeds, n_seeds, proc, cfg, v, &es);
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->seeds = n_seeds; stats->blocks_found = es.blocksFound; stats->failures = es.failures; stats->wall_ms = es.wallMs;
            stats->rounds = es.rounds; stats->recompute_launches = es.recomputeLaunches; stats->recomputed_seeds = es.recomputedSeeds;
            stats->conflict_launches = es.conflictLaunches; stats->conflict_seeds = es.conflictSeeds; stats->exchanges = es.exchanges;
            stats->jobs_used = es.jobsUsed; stats->views_built = es.viewsBuilt; stats->over_predicted = es.overPredicted;
            stats->process_ms = es.processMs; stats->plan_ms = es.planMs; stats->events = es.events;
            stats->side_batches = es.sideBatches; stats->side_jobs = es.sideJobs; stats->side_taken = es.sideTaken; stats->side_void = es.sideVoid; stats->side_failed = es.sideFailed;
            stats->early_critical = es.earlyCritical; stats->lazy_seeds = es.lazySeeds; stats->host_dead = es.hostDead; stats->collectives = es.collectives;
        }
    }
    *blocks = (lcb_block*)malloc((v.size() ? v.size() : 1) * sizeof(lcb_block));
    if (!*blocks) throw LcbError("out of memory");
    if (!v.empty()) memcpy(*blocks, v.data(), v.size() * sizeof(lcb_block));
    *n_blocks = (int64_t)v.size();
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}

int lcb_find_blocks(const lcb_graph* g, lcb_device* d, const lcb_params* p, const lcb_seed* seeds, int64_t n_seeds, int progress,
                    lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats)
{
    LCB_TRY
    LCB_NEED(g && d && p && (seeds || n_seeds == 0) && blocks && n_blocks, "lcb_find_blocks");
    std::vector<lcb_block> v;
    LcbEngineConfig cfg;
    cfg.progress = progress != 0;
    lcb_find_blocks_impl(g, d, p, seeds, n_seeds, cfg, v, stats);
    *blocks = (lcb_block*)malloc((v.size() ? v.size() : 1) * sizeof(lcb_block));
    if (!*blocks) throw LcbError("out of memory");
    if (!v.empty()) memcpy(*blocks, v.data(), v.size() * sizeof(lcb_block));
    *n_blocks = (int64_t)v.size();
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}

int lcb_comm_unique_id(unsigned char id[LCB_COMM_ID_BYTES]) { LCB_TRY LCB_NEED(id, "lcb_comm_unique_id"); lcb_comm_unique_id_impl(id); return LCB_OK; LCB_CATCH(LCB_ERR) }
lcb_comm* lcb_comm_create(lcb_device* d, const unsigned char id[LCB_COMM_ID_BYTES], int rank, int world)
{
    LCB_TRY
    LCB_NEED(d && id, "lcb_comm_create");
    return lcb_comm_create_impl(lcb_device_ordinal_impl(d), id, rank, world);
    LCB_CATCH(nullptr)
}
void lcb_comm_destroy(lcb_comm* c) { lcb_comm_destroy_impl(c); }
int lcb_find_blocks_comm(const lcb_graph* g, lcb_device* d, lcb_comm* c, const lcb_params* p, const lcb_seed* seeds, int64_t n_seeds,
                         const lcb_hooks* hooks, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats)
{
    LCB_TRY
    LCB_NEED(g && d && c && p && (seeds || n_seeds == 0) && blocks && n_blocks, "lcb_find_blocks_comm");
    LcbEngineConfig cfg = tuningOf(hooks);
    lcb_comm_fill_config(c, cfg);
    std::vector<lcb_block> v;
    lcb_find_blocks_impl(g, d, p, seeds, n_seeds, cfg, v, stats);
    return giveBlocks(v, blocks, n_blocks);
    LCB_CATCH(LCB_ERR)
}
int lcb_find_blocks_gpus(const lcb_graph* g, const int* device_ordinals, int n_devices, const lcb_params* p, const lcb_device_opts* opts,
                         const lcb_seed* seeds, int64_t n_seeds, const lcb_hooks* hooks, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats)
{
    LCB_TRY
    LCB_NEED(g && device_ordinals && p && (seeds || n_seeds == 0) && blocks && n_blocks, "lcb_find_blocks_gpus");
    std::vector<lcb_block> v;
    checkAbi(nullptr, opts);
    lcb_find_blocks_gpus_impl(g, device_ordinals, n_devices, p, opts, seeds, n_seeds, tuningOf(hooks), v, stats);
    return giveBlocks(v, blocks, n_blocks);
    LCB_CATCH(LCB_ERR)
}

struct lcb_gpus { lcb_gpus_impl* impl; };
lcb_gpus* lcb_gpus_create(const lcb_graph* g, const int* device_ordinals, int n_devices, const lcb_params* p, const lcb_device_opts* opts, int always_comm)
{
    LCB_TRY
    if (!g || !device_ordinals || !p) throw LcbError("lcb_gpus_create: null argument");
    checkAbi(nullptr, opts);
    return new lcb_gpus{lcb_gpus_create_impl(g, device_ordinals, n_devices, p, opts, always_comm != 0)};
    LCB_CATCH(nullptr)
}
int lcb_gpus_find_blocks(lcb_gpus* m, const lcb_seed* seeds, int64_t n_seeds, const lcb_hooks* hooks, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats)
{
    LCB_TRY
    LCB_NEED(m && m->impl && (seeds || n_seeds == 0) && blocks && n_blocks, "lcb_gpus_find_blocks");
    std::vector<lcb_block> v;
    lcb_gpus_find_blocks_impl(m->impl, seeds, n_seeds, tuningOf(hooks), v, stats);
    return giveBlocks(v, blocks, n_blocks);
    LCB_CATCH(LCB_ERR)
}
void lcb_gpus_destroy(lcb_gpus* m)
{
    if (!m) return;
    try { lcb_gpus_destroy_impl(m->impl); } catch (...) {}
    delete m;
}

int lcb_generate_output(const lcb_graph* g, int64_t min_block, const lcb_block* blocks, int64_t n_blocks, int64_t blocks_found,
                        const char* out_dir, int gen_seq, int64_t chunks, int64_t* n_trimmed, double* coverage)
{
    LCB_TRY
    LCB_NEED(g && (blocks || n_blocks == 0), "lcb_generate_output");
    lcb_generate_output_impl(*g, min_block, blocks, n_blocks, blocks_found, out_dir ? out_dir : "", gen_seq != 0, chunks, n_trimmed, coverage);
    return LCB_OK;
    LCB_CATCH(LCB_ERR)
}

}  // extern "C"
