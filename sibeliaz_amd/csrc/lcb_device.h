// lcb_device.h — one MI355X: tables + `used` bitmap resident in HBM, per-wavefront workspaces,
// kernel launches (device.hip).
#ifndef LCB_DEVICE_H
#define LCB_DEVICE_H

#include <cstdint>
#include <vector>

#include "lcb_host.h"

struct lcb_device_impl;

struct lcb_device {
    lcb_device_impl* impl;
};

lcb_device* lcb_device_create_impl(const lcb_graph* g, const lcb_params* p, int ordinal, const lcb_device_opts* opts = nullptr);
void lcb_device_destroy_impl(lcb_device* d);
void lcb_device_reset_used_impl(lcb_device* d);
void lcb_device_mark_used_impl(lcb_device* d, const uint64_t* ranges, int64_t n);
void lcb_device_set_used_impl(lcb_device* d, const uint32_t* words, int64_t nWords);
void lcb_device_set_stats_impl(lcb_device* d, bool on);
// Processes seeds[0..n): fills offsets[n+1] and inst (resized), bestScore (optional, n entries), ctr (optional, accumulated).
void lcb_device_process_impl(lcb_device* d, const lcb_seed* seeds, int64_t n, std::vector<uint64_t>& offsets,
                             std::vector<lcb_instance>& inst, int64_t* bestScore, lcb_counters* ctr,
                             std::vector<uint64_t>* fpOffsets = nullptr, std::vector<lcb_fp>* fp = nullptr,
                             const uint32_t* view = nullptr,    // view[i]: `used` view of seed i (null = the live state)
                             std::vector<lcb_counters>* perSeedCtr = nullptr);   // stats mode: the counters of every seed
// The same for a call whose first launch overlaps with host work: begin enqueues it against the live state of this moment (false:
// not applicable, use the synchronous call), end waits and completes it.
bool lcb_device_process_begin_impl(lcb_device* d, const lcb_seed* seeds, int64_t n);
void lcb_device_process_end_impl(lcb_device* d, std::vector<uint64_t>& offsets, std::vector<lcb_instance>& inst, std::vector<uint64_t>& fpOffsets,
                                 std::vector<lcb_fp>& fp);
// Predicted `used` views 1..nViews = live state + the marks with firstView <= v (engine.cpp).
void lcb_device_build_views_impl(lcb_device* d, int nViews, const LcbViewMark* marks, int64_t nMarks);
int lcb_device_max_views_impl(lcb_device* d);
// asynchronous job batches (LcbProcessor::side*, lcb_host.h)
int lcb_device_side_lanes_impl(lcb_device* d);
int lcb_device_side_begin_impl(lcb_device* d, const lcb_seed* seeds, const uint32_t* view, int64_t n, int nViews, const LcbViewMark* marks, int64_t nMarks);
int lcb_device_side_poll_impl(lcb_device* d, int lane, int64_t k, bool wait, std::vector<lcb_instance>& inst, std::vector<lcb_fp>& fp);
void lcb_device_side_release_impl(lcb_device* d, int lane);
double lcb_device_hbm_triad_impl(lcb_device* d, uint64_t bytes, int reps);
int lcb_device_concurrency_impl(lcb_device* d);          // seeds in flight in the compact variant
void lcb_device_mode_seeds_impl(lcb_device* d, int64_t out[4]);
void lcb_device_mode_time_impl(lcb_device* d, double ms[4], int64_t launches[4]);
// since the last call: sum of the hipEvent-timed kernel durations over all streams, launches, union of the kernels' intervals (the time
// the GPU was busy with them: side-lane kernels overlap the synchronous ones), and the part of the sum that ran on the side lanes
void lcb_device_kernel_time_impl(lcb_device* d, double* ms, int64_t* launches, double* busyMs = nullptr, double* sideMs = nullptr);
int64_t lcb_device_big_retries_impl(lcb_device* d);
void lcb_find_blocks_impl(const lcb_graph* g, lcb_device* d, const lcb_params* p, const lcb_seed* seeds, int64_t nSeeds,
                          const LcbEngineConfig& cfg, std::vector<lcb_block>& blocks, lcb_stats* stats);

// comm.hip — RCCL all-gather between the ranks of the round engine
struct lcb_comm;
void lcb_comm_unique_id_impl(unsigned char* id128);
lcb_comm* lcb_comm_create_impl(int ordinal, const unsigned char* id128, int rank, int world);
void lcb_comm_destroy_impl(lcb_comm* c);
void lcb_comm_fill_config(lcb_comm* c, LcbEngineConfig& cfg);
int lcb_device_ordinal_impl(lcb_device* d);
void lcb_find_blocks_gpus_impl(const lcb_graph* g, const int* ordinals, int n, const lcb_params* p, const lcb_device_opts* opts,
                               const lcb_seed* seeds, int64_t nSeeds, LcbEngineConfig cfg, std::vector<lcb_block>& blocks, lcb_stats* stats);

// a persistent set of GPUs of this node driven from one process (one host thread per GPU inside every pass)
struct lcb_gpus_impl;
lcb_gpus_impl* lcb_gpus_create_impl(const lcb_graph* g, const int* ordinals, int n, const lcb_params* p, const lcb_device_opts* opts, bool alwaysComm);
void lcb_gpus_find_blocks_impl(lcb_gpus_impl* m, const lcb_seed* seeds, int64_t nSeeds, LcbEngineConfig cfg, std::vector<lcb_block>& blocks, lcb_stats* stats);
void lcb_gpus_destroy_impl(lcb_gpus_impl* m);
int lcb_gpus_count_impl(const lcb_gpus_impl* m);

#endif
