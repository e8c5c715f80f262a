/* lcb.h — C ABI of the MI355X-native locally-collinear-block finder (libsibeliaz_amd.so).
 *
 * The reference (SibeliaZ-LCB, C++11/OpenMP) has no plugin or FFI surface: its only in-process seam
 * is BlocksFinder::FindBlocks (blocksfinder.h:453) -> blocksInstance_ (blocksfinder.h:921) ->
 * GenerateOutput (blocksfinder.h:605) over a JunctionStorage (junctionstorage.h:116-698). This
 * header is that seam as plain C: opaque handles, plain pointers and sizes, integer return codes
 * (0 = ok, <0 = error, message via lcb_last_error()), no exceptions and no torch types across
 * the boundary. Each entry point cites the reference interface it replaces. INTEGRATION.md shows
 * the binding a reference maintainer would add.
 *
 * Threading: one host thread per lcb_device; lcb_graph is immutable after load and may be shared.
 * The hot path (lcb_process_seeds) runs only on an MI355X; there is no CPU fallback — without a GPU
 * lcb_device_create fails loudly.
 */
#ifndef LCB_H
#define LCB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCB_OK 0
#define LCB_ERR (-1)

typedef struct lcb_graph lcb_graph;         /* replaces Sibelia::JunctionStorage (junctionstorage.h:116-698) */
typedef struct lcb_device lcb_device;       /* tables + `used` bitmap resident in HBM, per-wavefront workspaces */
typedef struct lcb_committer lcb_committer; /* ordered commit state of ProcessVertex::operator() (blocksfinder.h:372-427) */

typedef struct {
    int32_t k;             /* sibeliaz.cpp:45-51  (-k, odd) */
    int32_t min_block;     /* sibeliaz.cpp:61-67  (-m) -> FindBlocks minBlockSize */
    int32_t max_branch;    /* sibeliaz.cpp:53-59  (-b) -> FindBlocks maxBranchSize */
    int32_t max_flank;     /* = -b, sibeliaz.cpp:136 */
    int32_t looking_depth; /* 8, sibeliaz.cpp:137 */
    int32_t phase_size;    /* 256, blocksfinder.h:519 — part of the semantics */
} lcb_params;

typedef struct {           /* BlocksFinder::Bundle (blocksfinder.h:182-209) */
    int32_t vid;           /* signed vertex id */
    int32_t ch;            /* next character of the seed edge */
    uint64_t count;
    uint64_t rank;
    uint64_t resolve_pos;
    uint64_t resolve_chr;
} lcb_seed;

typedef struct {           /* one Path::Instance of a per-seed result (path.h:53-181) */
    uint32_t chr;
    uint32_t front_idx;    /* Front().GetIndex() */
    uint32_t back_idx;     /* Back().GetIndex() */
    uint32_t positive;     /* Front().IsPositiveStrand() */
} lcb_instance;

typedef struct {           /* Sibelia::BlockInstance (blocksfinder.h:29-51) */
    int32_t id;            /* signed block id */
    uint32_t chr;
    uint64_t start;
    uint64_t end;
} lcb_block;

typedef struct {           /* reference-semantics event counters; only filled in stats mode */
    uint64_t n_walk;        /* iterations of the look-ahead loop, blocksfinder.h:722-756 */
    uint64_t n_occ;         /* occurrences visited at path.h:38, :446, :515 */
    uint64_t n_compat_call; /* calls of Path::Compatible, path.h:380 */
    uint64_t n_compat_step; /* iterations of its `used` walk, path.h:387-393 */
    uint64_t n_inst_out;    /* instances in final per-seed results */
    uint64_t n_vote;        /* MostPopularVertex calls */
    uint64_t n_push;        /* successful PointPushBack/Front */
    uint64_t n_process;     /* Process() calls */
} lcb_counters;

typedef struct {
    int64_t seeds;          /* bundle_.size() */
    int64_t blocks_found;   /* blocksFound_ */
    int64_t failures;       /* failure_ : ordered-commit conflicts that were re-processed (blocksfinder.h:406) */
    int64_t launches;       /* kernel launches */
    int64_t big_retries;    /* seeds re-run with global-memory workspaces after an LDS capacity overflow */
    double kernel_ms;       /* sum of the hipEvent-timed kernel durations over the device's streams (see kernel_busy_ms) */
    double wall_ms;         /* wall time of the phase loop */
    int64_t rounds;             /* speculative multi-phase launches */
    int64_t recompute_launches; /* job launches: after a stop of the ordered commit, every seed the dry run expects to need a new result */
    int64_t recomputed_seeds;   /* ... jobs in those launches */
    int64_t conflict_launches;  /* job launches whose stop was a conflicting seed waiting for its re-processed result (blocksfinder.h:406-411) */
    int64_t conflict_seeds;     /* jobs that re-process a (predicted) conflict against the (predicted) live state */
    int64_t exchanges;          /* variable-size exchanges between the ranks (multi-rank): a launch's results, published background results, agreements */
    int64_t jobs_used;          /* job results that passed the exact validation and were committed from */
    int64_t views_built;        /* predicted `used` views materialised on the device */
    int64_t over_predicted;     /* validations that failed because a predicted mark did not come true */
    double process_ms;          /* wall time inside launches + result gathering (kernel_ms is the device part of it) */
    double plan_ms;             /* wall time of the dry runs */
    lcb_counters events;        /* lcb_hooks.count_events: the reference-semantics event counts of the whole FindBlocks (phase-start
                                   Process() of every seed + the re-Process() of every commit conflict), else zero */
    int64_t side_batches;       /* asynchronous job batches (side lanes): a stop waits only for the results it cannot go on without */
    int64_t side_jobs;          /* ... their jobs (also counted in recomputed_seeds) */
    int64_t side_taken;         /* ... results taken when the commit reached their seed */
    int64_t side_void;          /* ... jobs dropped: a mark of their view did not come true, superseded, or the round ended */
    int64_t side_failed;        /* ... jobs that ended without a result (stopped, or needed another kernel variant) */
    int64_t early_critical;     /* stops whose own jobs were computed while the host planned the rest */
    double kernel_busy_ms;      /* UNION of the hipEvent-timed kernel intervals of all streams: the time the GPU was busy with process kernels
                                   (kernel_ms is their SUM; the side lanes' kernels run beside the synchronous ones, so the sum can exceed the pass) */
    double kernel_side_ms;      /* the part of kernel_ms that ran on the side lanes' streams */
    int64_t lazy_seeds;         /* seeds in the lazy tails of the rounds: no speculative launch, their phase-start results are background jobs (lcb_hooks.lazy_span) */
    int64_t collectives;        /* all-gathers issued for the exchanges: one where every rank's buffer is small, else two */
    int64_t host_dead;          /* results settled on the host without the device: no unused occurrence of the seed's vertex carries its character (lcb_hooks.sparse_rounds) */
} lcb_stats;

/* Message of the last failing call on this thread. */
const char* lcb_last_error(void);
/* Library version string. */
const char* lcb_version(void);
/* Layout version of the structs of this header (lcb_stats, lcb_hooks, lcb_device_opts): they are allocated by the caller, so a caller
 * built against another LCB_ABI_VERSION must not call in. lcb_abi_version() returns the library's. */
#define LCB_ABI_VERSION 6
int lcb_abi_version(void);

/* ---- graph: JunctionStorage::Init (junctionstorage.h:572-650), junctionapi.h:80-98, streamfastaparser.cpp:28-92 */
lcb_graph* lcb_graph_load(const char* junction_file, const char* const* fasta_files, int n_fasta,
                          int k, int abundance, int threads);
void lcb_graph_free(lcb_graph* g);
int64_t lcb_graph_n_chr(const lcb_graph* g);          /* GetChrNumber, junctionstorage.h:522 */
int64_t lcb_graph_n_pos(const lcb_graph* g);          /* total junction occurrences kept (P) */
int64_t lcb_graph_n_vertices(const lcb_graph* g);     /* GetVerticesNumber, junctionstorage.h:557 */
int64_t lcb_graph_chr_len(const lcb_graph* g, int64_t chr);     /* GetChrSequence(chr).size() */
int64_t lcb_graph_chr_n_pos(const lcb_graph* g, int64_t chr);   /* GetChrVerticesCount, junctionstorage.h:537 */
const char* lcb_graph_chr_name(const lcb_graph* g, int64_t chr);/* GetChrDescription, junctionstorage.h:532 */
/* SoA views (flat position index g = chr_start[chr] + idx); valid while the graph lives. */
const uint64_t* lcb_graph_chr_start(const lcb_graph* g);        /* [n_chr+1] */
const int32_t* lcb_graph_pos_id(const lcb_graph* g);            /* [n_pos] Position::id,  junctionstorage.h:142 */
const uint32_t* lcb_graph_pos_pos(const lcb_graph* g);          /* [n_pos] Position::pos, junctionstorage.h:143 */

/* ---- seeds: bundle enumeration + std::sort (blocksfinder.h:461-503,517). *out is malloc'ed; free with lcb_free. */
int64_t lcb_enumerate_seeds(const lcb_graph* g, int threads, lcb_seed** out);
void lcb_free(void* p);

/* ---- device: one MI355X. device_ordinal is the HIP device index. */
lcb_device* lcb_device_create(const lcb_graph* g, const lcb_params* p, int device_ordinal);
/* Tuning knobs of a device; a zero field means "default". Results never depend on them (tests sweep them). The struct is allocated by
 * the caller: `abi` must hold the LCB_ABI_VERSION of the header it was compiled against (a mismatch is rejected, not misread). */
typedef struct {
    uint32_t abi;            /* = LCB_ABI_VERSION */
    uint32_t compact_slots;  /* workgroups (= seeds in flight) of the compact kernel variant; default 5 per CU (8 with the small pools, compact_pools) */
    uint32_t wide_slots;     /* ... of the wide variant; default 1 per CU */
    uint32_t big_slots;      /* ... of the big variant (index in LDS, instance fields in HBM); default 1 per CU */
    uint32_t huge_slots;     /* ... of the huge variant (all per-path state in HBM); default 1 per 4 CUs */
    uint32_t path_cap;       /* INITIAL path vertex set capacity of a compact slot (power of two; HBM-resident); default 32768.
                                Grows x4 on demand (a path of few instances that overflows it) up to path_cap_max */
    uint32_t wide_path_cap;  /* ... of a wide slot (the set lives in LDS); default and maximum 8192 */
    uint32_t max_views;      /* predicted `used` views kept behind the live bitmap; default 256 (within 2 GiB) */
    uint32_t batch;          /* seeds per launch; default 65536 */
    uint32_t wide_threshold; /* calls with at most this many seeds start in the wide variant; default 2 * wide_slots */
    uint32_t start_mode;     /* 0 = automatic; 1 / 2 / 3 / 4 = every seed starts in the compact / wide / big / huge variant */
    uint32_t screen_min;     /* launches of at least this many seeds are screened first; default 2048 */
    uint32_t path_cap_max;   /* largest compact path set (a seed that needs more goes to the big variant); default 1 << 20 */
    uint32_t arena;          /* INITIAL capacity of the pinned result arena of a launch, in instances (and footprint intervals);
                                default 1 << 22; enlarged x4 when a launch fills it (its unlucky seeds run again) */
    uint32_t side_lanes;     /* asynchronous job batches that can be in flight beside the synchronous launches (own streams, buffers,
                                workspace slots and predicted views each); default 4; 0xFFFFFFFF = none */
                             /* (the stream of the synchronous launches - the results the commit waits for - has the highest HIP stream priority,
                                the side lanes' streams the lowest: speculation that fills the machine does not delay a needed result) */
    /* Positions on the device are (segment, 32-bit offset) pairs; a segment is a run of whole chromosomes (the reference bounds a
     * chromosome by 2^32, junctionstorage.h:120-151, not the input). Test hooks - results never depend on them: */
    uint64_t seg_cap;        /* most junction occurrences per segment; default 2^32 - 2^20. Non-zero: the segment-aware kernels run even
                                if everything fits one segment (small values cut a small input into many segments) */
    uint32_t side_big_cap;   /* most jobs of one background batch that run in the big variant (seeds known to need it: one workgroup per CU for tens of
                                milliseconds each); the others get no result there - the commit computes them when it needs them. Default: the big-variant slots of a
                                lane (one wave); 0xFFFFFFFF = no cap */
    uint32_t compact_pools;  /* pools of the compact variant: 1 = 256 instances / 1 024 vote slots (5 workgroups per CU), 2 = 128 / 512 (8 per CU: more seeds in
                                flight where paths have few instances); 0 = chosen by the input (small pools if the typical occurrence belongs to a vertex
                                with at most 20 occurrences; back to the large ones if more than an eighth of the live seeds outgrow them) */
    uint64_t seg_gap;        /* unused positions between two segments of the device tables (0 = none): with 2^32 the flat indices of a
                                small input exceed 32 bits, i.e. every 64-bit address computation of the kernels is exercised */
} lcb_device_opts;
lcb_device* lcb_device_create_ex(const lcb_graph* g, const lcb_params* p, int device_ordinal, const lcb_device_opts* opts);
/* Seeds handed to the compact / wide / big / huge kernel variant since the device was created (a seed that overflows
 * one variant is counted again in the next). */
int lcb_device_mode_seeds(lcb_device* d, int64_t counts[4]);
/* hipEvent-timed kernel time (ms) and launches of the compact / wide / big / huge variant since the device was created, over all of its
 * streams (measurement: bench.py's roofline.per_kernel; the background batches of the side lanes count when they retire). */
int lcb_device_mode_time(lcb_device* d, double ms[4], int64_t launches[4]);
void lcb_device_destroy(lcb_device* d);
/* `used` bits (Position::used, junctionstorage.h:144) live in HBM as a bitmap over g. */
int lcb_device_reset_used(lcb_device* d);
/* Set bits [lo, hi) for each of n ranges given as pairs (lo, hi) of flat position indices. */
int lcb_device_mark_used(lcb_device* d, const uint64_t* ranges, int64_t n);
/* Replace the bitmap wholesale (n_words = ceil(n_pos / 32)). */
int lcb_device_set_used(lcb_device* d, const uint32_t* words, int64_t n_words);
/* 1: collect lcb_counters in the kernels (slower; used for the roofline's algorithmic bytes). */
int lcb_device_set_stats_mode(lcb_device* d, int on);

/* THE HOT PATH: ProcessVertex::Process (blocksfinder.h:228-310) for a batch of seeds, each against the
 * device's current `used` state, one seed per workgroup (wavefront 0 runs the seed, the others share its look-ahead votes). offsets has n+1 entries; the instances of seed
 * i are inst[offsets[i] .. offsets[i+1]). Returns LCB_OK, or LCB_ERR (e.g. inst_cap too small: the needed
 * capacity is then in offsets[n]). best_score and ctr may be NULL. */
int lcb_process_seeds(lcb_device* d, const lcb_seed* seeds, int64_t n, uint64_t* offsets,
                      lcb_instance* inst, uint64_t inst_cap, int64_t* best_score, lcb_counters* ctr);
/* The same, also returning every seed's FOOTPRINT: intervals [lo, hi] of flat positions (pairs of uint64) that cover every position
 * whose `used` bit the seed's computation read as 0 - what makes the engine's speculation exact (a result stays valid as long as
 * no bit inside its footprint has been set since). fp_offsets has n+1 entries; interval j of seed i is fp[2*j], fp[2*j+1] for j in
 * fp_offsets[i] .. fp_offsets[i+1]. On LCB_ERR for lack of room the needed capacities are in offsets[n] / fp_offsets[n]. */
int lcb_process_seeds_fp(lcb_device* d, const lcb_seed* seeds, int64_t n, uint64_t* offsets, lcb_instance* inst, uint64_t inst_cap,
                         uint64_t* fp_offsets, uint64_t* fp, uint64_t fp_cap);
/* Measured HBM rate of this GPU: STREAM triad over three arrays of `bytes` each, GB/s (the roofline's measured peak). */
int lcb_device_hbm_triad(lcb_device* d, uint64_t bytes, int reps, double* gb_per_s);
/* hipEvent-timed duration (ms) and launch count of kernels since the last call (reset on read). */
int lcb_device_kernel_time(lcb_device* d, double* ms, int64_t* launches);

/* ---- ordered commit: thread-0 section of ProcessVertex::operator() + Finalize (blocksfinder.h:312-332,372-427).
 * Host-side, usable without a GPU (multi-GPU ranks run it redundantly on all-gathered results). */
typedef int (*lcb_reprocess_fn)(void* user, const lcb_seed* seed, lcb_instance* out, uint64_t cap, uint64_t* n_out);
lcb_committer* lcb_committer_create(const lcb_graph* g, const lcb_params* p);
void lcb_committer_free(lcb_committer* c);
/* Commits one phase's results in seed order. Conflicting seeds are re-processed through fn against the live
 * state (fn must first apply lcb_committer_take_marks to its device). */
int lcb_committer_commit_phase(lcb_committer* c, const lcb_seed* seeds, int64_t n, const uint64_t* offsets,
                               const lcb_instance* inst, lcb_reprocess_fn fn, void* user);
/* Ranges (lo, hi pairs over g) marked used since the previous call; returns the number of ranges written
 * (at most cap) and leaves the rest queued. */
int64_t lcb_committer_take_marks(lcb_committer* c, uint64_t* ranges, int64_t cap);
int64_t lcb_committer_n_blocks(const lcb_committer* c);          /* blocksInstance_.size() */
const lcb_block* lcb_committer_blocks(const lcb_committer* c);   /* pre-trim, commit order */
int64_t lcb_committer_blocks_found(const lcb_committer* c);
int64_t lcb_committer_failures(const lcb_committer* c);
const uint32_t* lcb_committer_used_words(const lcb_committer* c, int64_t* n_words);

/* ---- BlocksFinder::FindBlocks on one GPU (blocksfinder.h:453-530): phase loop, kernel launches, ordered
 * commit. *blocks is malloc'ed (pre-trim blocksInstance_ in commit order); free with lcb_free. */
int lcb_find_blocks(const lcb_graph* g, lcb_device* d, const lcb_params* p, const lcb_seed* seeds, int64_t n_seeds,
                    int progress, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats);

/* ---- the same with hooks: multi-rank operation and/or a caller-supplied per-seed engine.
 * rank/world + allgather: the round's seeds are dealt round-robin to the ranks, per-seed results and footprints are
 * all-gathered (the callback gathers a fixed-size buffer: recv holds world * bytes), every rank commits identically.
 * If dev is NULL the process/mark/reset callbacks stand in for the device (used by the CPU tests of this logic). */
typedef int (*lcb_allgather_cb)(void* user, const void* send, uint64_t bytes, void* recv);
typedef int (*lcb_process_cb)(void* user, const lcb_seed* seeds, int64_t n, uint64_t* offsets /* n+1 */, lcb_instance* inst,
                              uint64_t inst_cap);   /* returns 0, or 1 with the needed capacity in offsets[n] */
typedef int (*lcb_mark_cb)(void* user, const uint64_t* ranges, int64_t n);
typedef int (*lcb_reset_cb)(void* user);
typedef struct {
    int32_t abi;                /* = LCB_ABI_VERSION (the struct is allocated by the caller: a mismatch is rejected, not misread) */
    int32_t rank, world;
    lcb_allgather_cb allgather;
    void* allgather_user;
    lcb_process_cb process;     /* only when dev == NULL */
    lcb_mark_cb mark;
    lcb_reset_cb reset;
    void* engine_user;
    int32_t round_phases;       /* phases per speculative round (upper bound of the adaptive size); 0 = default 256 */
    int32_t progress;
    /* engine tuning, 0 = default; results never depend on these (tests sweep them) */
    int32_t round_fixed;        /* 1: every round has round_phases phases (no adaptation) */
    int32_t eager_phases;       /* phases a dry run plans ahead; default 256; -1 = none */
    int32_t max_views;          /* predicted views used per job launch; default: all the device has; -1 = none */
    int32_t max_jobs;           /* a dry run stops planning beyond this many jobs; default: the device's seeds in flight */
    int32_t predict_f;          /* how a dry run predicts the re-processed result of a conflicting seed: 1 nothing,
                                   2 the still-free instances of its phase-start result, 3 (default) a stale re-processed result if any, else as 2 */
    int32_t exchange_always;    /* 1: a single rank still packs / all-gathers / unpacks every launch (tests of the exchange path) */
    int32_t count_events;       /* 1: fill lcb_stats.events (the device must be in stats mode; one rank) */
    int32_t sync_jobs;          /* 1: do not use the device's side lanes - every job of a stop's plan runs in one synchronous launch
                                   (the round-2 engine; for A/B runs and tests). With side lanes the results a stop cannot go on without
                                   are launched BEFORE the dry run that plans the rest of the stop's jobs (measured: profiles/r04/ab_first.txt) */
    int32_t lazy_span;          /* a round spans at least this many phases: the phases beyond the (adaptive) size of its speculative launch get their
                                   phase-start results as background jobs against predicted views, planned while the commit works through the stops of the
                                   phases in front of them. Default 8; -1 = off (a round is exactly its speculative launch) */
    int32_t sparse_rounds;      /* round 6. 0 (default): the host settles the results it can settle itself - a seed none of whose occurrences is unused with its
                                   character has an empty result against every later state too (lcb_stats.host_dead) - instead of asking the device. 1: also
                                   sparse speculative launches: seeds sorted next to each other lie next to each other in the genome (Bundle::operator<,
                                   blocksfinder.h:195-208), so a round launches only the seeds of the FIRST phase of every such cluster and spans as many
                                   phases as that takes; the others are settled on the host when their phase starts, or computed then (measured: slower on
                                   every test shape, neutral at Gbp scale - an experiment, not a default). -1: neither */
} lcb_hooks;
int lcb_find_blocks_ex(const lcb_graph* g, lcb_device* d, const lcb_params* p, const lcb_seed* seeds, int64_t n_seeds,
                       const lcb_hooks* hooks, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats);

/* ---- multi-GPU: launches of the round engine dealt to one rank per MI355X, per-seed results and footprints all-gathered
 * with RCCL (ncclAllGather over xGMI) from the C++ host; every rank runs the identical ordered commit and returns the same
 * blocks. The reference has no counterpart (OpenMP only); tables and the `used` bitmap are replicated per GPU. */
typedef struct lcb_comm lcb_comm;
#define LCB_COMM_ID_BYTES 128
/* One process per GPU: rank 0 calls lcb_comm_unique_id and ships the bytes to the other ranks (MPI, torch.distributed,
 * a file ...); then every rank creates its communicator on its device (collective: ncclCommInitRank). */
int lcb_comm_unique_id(unsigned char id[LCB_COMM_ID_BYTES]);
lcb_comm* lcb_comm_create(lcb_device* d, const unsigned char id[LCB_COMM_ID_BYTES], int rank, int world);
void lcb_comm_destroy(lcb_comm* c);
/* lcb_find_blocks_ex on rank c->rank of c->world; hooks (may be NULL) only supplies the engine tuning fields. */
int lcb_find_blocks_comm(const lcb_graph* g, lcb_device* d, lcb_comm* c, const lcb_params* p, const lcb_seed* seeds, int64_t n_seeds,
                         const lcb_hooks* hooks, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats);
/* One process, one host thread + device per listed GPU (what `sibeliaz-lcb` does with LCB_GPUS=N). */
int lcb_find_blocks_gpus(const lcb_graph* g, const int* device_ordinals, int n_devices, const lcb_params* p, const lcb_device_opts* opts,
                         const lcb_seed* seeds, int64_t n_seeds, const lcb_hooks* hooks, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats);

/* The same with a persistent handle: devices are created, the tables uploaded and RCCL initialised ONCE; every
 * lcb_gpus_find_blocks is one pass of the phase loop over all the GPUs of the set (what bench.py --gpus N times, and what
 * sibeliaz-lcb uses with LCB_GPUS=N). always_comm != 0: a set of one GPU still goes through the RCCL exchange path (tests). */
typedef struct lcb_gpus lcb_gpus;
lcb_gpus* lcb_gpus_create(const lcb_graph* g, const int* device_ordinals, int n_devices, const lcb_params* p, const lcb_device_opts* opts, int always_comm);
int lcb_gpus_find_blocks(lcb_gpus* m, const lcb_seed* seeds, int64_t n_seeds, const lcb_hooks* hooks, lcb_block** blocks, int64_t* n_blocks, lcb_stats* stats);
void lcb_gpus_destroy(lcb_gpus* m);

/* ---- GenerateOutput (blocksfinder.h:605-670): trimming, blocks_coords.gff (blocksfinder.cpp:141-174) and,
 * if gen_seq, the <out_dir>/<i>.tmp chunk files (blocksfinder.h:533-582). */
int lcb_generate_output(const lcb_graph* g, int64_t min_block, const lcb_block* blocks, int64_t n_blocks,
                        int64_t blocks_found, const char* out_dir, int gen_seq, int64_t chunks,
                        int64_t* n_trimmed, double* coverage);

#ifdef __cplusplus
}
#endif
#endif
