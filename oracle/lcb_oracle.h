/* lcb_oracle.h — CPU restatement of SibeliaZ-LCB's BlocksFinder hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (sibeliaz_amd/, include/, the sibeliaz-lcb
 * executable) may include, link or call this. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * Parity pinning: this restatement is validated against outputs of the real reference compiled
 * here (oracle/_ref, see oracle/Makefile `ref`) — blocks_coords.gff, the sorted seed list, the
 * pre-trim block instances and per-seed Process() results — committed under tests/golden/.
 */
#ifndef LCB_ORACLE_H
#define LCB_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_graph orc_graph;

typedef struct {
    int64_t k;              /* sibeliaz.cpp:45-51 */
    int64_t min_block;      /* -m, blocksfinder.h:458 */
    int64_t max_branch;     /* -b, blocksfinder.h:459 */
    int64_t max_flank;      /* = -b, sibeliaz.cpp:136 */
    int64_t looking_depth;  /* 8, sibeliaz.cpp:137 */
} orc_params;

typedef struct {            /* one Path::Instance of a per-seed result (path.h:53-181) */
    int32_t positive;       /* strand of Front()/Back() */
    uint32_t chr;
    uint32_t front_idx;
    uint32_t back_idx;
} orc_inst;

typedef struct {            /* BlockInstance (blocksfinder.h:29-51) */
    int32_t id;             /* signed block id */
    uint64_t chr, start, end;
} orc_block;

typedef struct {            /* reference-semantics event counters, SURVEY.md §8(d) */
    uint64_t n_walk;        /* iterations of the look-ahead loop, blocksfinder.h:722-756 */
    uint64_t n_occ;         /* occurrences visited at path.h:38, :446, :515 */
    uint64_t n_compat_call; /* calls of Path::Compatible, path.h:380 */
    uint64_t n_compat_step; /* iterations of the `used` walk, path.h:387-393 */
    uint64_t n_inst_out;    /* instances in the final per-seed results */
    uint64_t n_vote;        /* calls of MostPopularVertex */
    uint64_t n_push;        /* successful PointPushBack/Front */
    uint64_t n_process;     /* Process() calls */
} orc_counters;

typedef struct {
    int64_t blocks_found;   /* blocksFound_ */
    int64_t failures;       /* failure_ (re-processes, blocksfinder.h:406) */
    int64_t seeds;          /* bundle_.size() */
} orc_stats;

/* JunctionStorage::Init (junctionstorage.h:572-650). Returns NULL and fills err on failure. */
orc_graph* orc_load(const char* graph_file, const char* const* fasta, int n_fasta, int64_t k,
                    int64_t abundance, char* err, size_t err_len);
void orc_free(orc_graph* g);

int64_t orc_n_chr(const orc_graph* g);
int64_t orc_n_vertices(const orc_graph* g);           /* vertex_.size() = max|id|+1 */
int64_t orc_chr_len(const orc_graph* g, int64_t chr);
int64_t orc_chr_n_pos(const orc_graph* g, int64_t chr);
const char* orc_chr_name(const orc_graph* g, int64_t chr);
/* copies position_[chr] (id, pos) into caller arrays of orc_chr_n_pos entries */
void orc_chr_positions(const orc_graph* g, int64_t chr, int32_t* id, uint32_t* pos);
/* `used` flags of position_[chr] (one byte per entry); writable, for differential tests */
uint8_t* orc_chr_used(orc_graph* g, int64_t chr);
/* diagnostics (tests/emu/engine_model): pushes made before the first read of a watched `used` byte (value 3), per thread; -1 = none was read */
void orc_watch_begin(void);
int64_t orc_watch_end(int64_t* pushes);
void orc_reset_used(orc_graph* g);

/* Seed enumeration + sort, blocksfinder.h:461-503,517. Returns S. */
int64_t orc_build_bundles(orc_graph* g);
void orc_get_bundle(const orc_graph* g, int64_t i, int64_t* vid, int32_t* ch, uint64_t* count,
                    uint64_t* rank, uint64_t* resolve_pos, uint64_t* resolve_chr);

/* ProcessVertex::Process (blocksfinder.h:228-310) for one seed against the current `used`
 * state. Returns the number of instances of the result (bestInstance.size()); at most `cap`
 * are written to out. */
int64_t orc_process_seed(orc_graph* g, const orc_params* p, int64_t vid, int32_t ch,
                         orc_inst* out, int64_t cap, int64_t* best_score, orc_counters* ctr);

/* Same as above but appends a line-per-event trace to a file (debug aid for kernel diffs). */
void orc_set_trace(const char* file);

/* Reusable per-thread finder that also reports a seed's footprints (per instance ever created: the range of flat positions,
 * first position of the chromosome + index, whose `used` bits were read as 0; chr = -1 marks flat coordinates). Not part of the
 * reference: support for tests / models of the round engine. */
typedef struct { int64_t chr, lo, hi; } orc_fp;
typedef struct orc_worker orc_worker;
orc_worker* orc_worker_new(orc_graph* g, const orc_params* p);
void orc_worker_free(orc_worker* w);
int64_t orc_worker_process(orc_worker* w, int64_t vid, int32_t ch, orc_inst* out, int64_t cap, int64_t* best_score, orc_counters* ctr,
                           orc_fp* fp, int64_t fp_cap, int64_t* n_fp);
int64_t orc_worker_path_vertices(const orc_worker* w, int64_t* out, int64_t cap);   /* |id| of every vertex ever in the last call's path */

/* FindBlocks phase loop + ordered commit (blocksfinder.h:334-433,453-530): fills *out with the
 * pre-trim blocksInstance_ in commit order (malloc'ed, caller frees with orc_free_blocks). */
int64_t orc_find_blocks(orc_graph* g, const orc_params* p, orc_block** out, orc_stats* st, orc_counters* ctr);
void orc_free_blocks(orc_block* b);

/* GenerateOutput (blocksfinder.h:605-670) + ListBlocksIndicesGFF (blocksfinder.cpp:141-174).
 * Writes <out_dir>/blocks_coords.gff. Returns the number of trimmed blocks or -1. */
int64_t orc_generate_output(const orc_graph* g, int64_t min_block, const orc_block* blocks, int64_t n,
                            int64_t blocks_found, const char* out_dir, double* coverage, char* err, size_t err_len);

/* libstdc++ std::sort (introsort) restated; exposed so tests can pin it against known outputs. */
void orc_introsort(void* base, size_t n, size_t size, int (*less)(const void*, const void*, void*), void* ctx);

#ifdef __cplusplus
}
#endif
#endif
