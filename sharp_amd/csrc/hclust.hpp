// hclust.hpp -- batched get_opt_hclust (R/get_opt_hclust.R:33-244) on the GPU: distance build,
// agglomerative tree, cutree for every candidate k, median silhouette, CH index, model selection.
#pragma once
#include <memory>
#include <functional>
#include <vector>

#include "common.hpp"

namespace sharp {

struct HcParams {
    int hmethod = 1;          // stats::hclust method table, 1 = ward.D
    int N_cluster = 0;        // 0 = NULL: choose k automatically
    int minN = 2, maxN = 40;
    double sil_thre = 0.35;
    double height_Ntimes = 2.0;
    // where the call sits in the run, for the decision log only (sharp_decision_log): level 0 = base clustering of one random projection of
    // one fold (getrowColor), 1 = a fold's wMetaC, 2 = the sMetaC across a block's folds, 3 = SHARP_unlimited's sMetaC across blocks,
    // -1 = a direct call of the entry point
    int dec_level = -1, dec_block = 0, dec_k = 0, dec_fold = 0;
};

// The decision log ("tests must log the best-vs-second-best margin of every model-selection decision"): one row per
// get_opt_hclust call while it is on (sharp_decision_log(1) or SHARP_DECISION_LOG=1), the same row the CPU checker of tests/ writes:
//  0 level  1 block  2 k (projection)  3 fold  4 n  5 branch (0 median silhouette, 1 CH, 2 height gap, 3 N.cluster given)  6 chosen number of
//  clusters  7 exact ties at the deciding maximum (R/get_opt_hclust.R:162-168 picks the middle one; which.max the first, :194-195)
//  8 the deciding maximum (msil for branch 0 / 3, CH for 1 / 2)  9 the largest value strictly below it (NaN: none)
//  10 max(msil) - sil.thre: the distance to the silhouette / CH switch (:194)
//  11 height rule (:196-210), when CH's first level won: (gap / ((height.Ntimes - 1) * height)) of the deciding step (branch 2: > 1) or its
//     maximum over the last ten merges (branch 1: <= 1); NaN otherwise
//  12 sMetaC's two-cluster override (R/sMetaC.R:139-148): the number of clusters of the column taken instead, 0 = not applied
//  13 number of candidate levels
constexpr int kDecisionCols = 14;
bool decision_log_on();
void decision_log_set(bool on);                 // (clears the log)
void decision_log_add(const double *row);
void decision_log_override(int level, int block, int k_taken);
int decision_log_fetch(double *rows, int cap_rows);     // rows sorted by (level, block, k, fold); returns the number of rows held
void decision_row(const HcParams &prm, int n, int kmin, int nk, const double *msil, const double *CH, const double *height, int oind,
                  int branch, double *row);

// One clustering problem of a batch.  d_mat is DEVICE memory: n x p row-major with leading dimension
// ld (feature rows), or an n x n symmetric similarity (symmetric = true, p = n).
struct HcTask {
    const double *d_mat = nullptr;
    long long ld = 0;
    int n = 0, p = 0;
    bool symmetric = false;
    HcParams prm;             // per-task (sMetaC adjusts minN/maxN per call)
};

struct HcResult {
    int rc = 0;               // SHARP_OK or SHARP_WARN_RANGE
    std::vector<int> f;       // chosen labels, 1-based, numbered by first appearance
    std::vector<int> v;       // n x nk column-major (only when want_v)
    std::vector<double> msil, CHind, height;
    double maxsil = 0;
    int optN = 0, nk = 0, branch = 0;
};

// Observations per clustering task.  Up to kHcLdsMaxN (sequential kernel) / 4096 (bulk-synchronous kernel) the agglomeration state
// of a task lives in its workgroup's LDS; beyond, up to kHcMaxN, the same kernels keep it in global memory (the cross-block sMetaC
// of SHARP_unlimited over thousands of block-level clusters, R/SHARP_unlimited.R:163; the per-block sMetaC of a block of hundreds of
// folds).  kHcMaxN: the silhouette medians are sorted in LDS (8 B per observation, rounded up to a power of two) and indices are 15-bit.
constexpr int kHcLdsMaxN = 7168;
constexpr int kHcMaxN = 16384;

// progress (optional; only used when the batch runs as pipelined chunks): called with the number of leading tasks whose results in
// `out` are final, at moments when the device has later chunks' agglomeration to work on; it may itself call get_opt_hclust_batch
// (that nested batch gets buffers and agglomeration scratch of its own).
// prepare (optional; pipelined chunks only): called on the calling thread before a chunk's work is enqueued, with the number of leading
// tasks whose inputs that chunk needs; it enqueues whatever still has to produce them (on the caller's stream) and returns the event behind
// it, which the chunk's stream waits for.  Without it every chunk waits for the caller's stream as it stands at the call.
void get_opt_hclust_batch(const std::vector<HcTask> &tasks, bool want_v, std::vector<HcResult> &out,
                          const std::function<void(size_t)> *progress = nullptr,
                          const std::function<hipEvent_t(size_t)> *prepare = nullptr);

// One-shot hook for the NEXT get_opt_hclust_batch call of this device slot that runs as pipelined chunks: fn(ev) is called once, on the
// host, right after the LAST chunk's agglomeration has been enqueued; ev is recorded behind that agglomeration.  For work that needs
// none of the batch's results and should share the chip with the last chunk's statistics rather than with an agglomeration (the
// ensemble mean of SHARP_large: one HBM-bound pass over E).  Returns through hc_after_last_agglomeration_fired() whether it ran.
void hc_release_workspaces();   // the clustering workspaces of the calling thread's slot (sharp_trim; the device is idle)
void hc_set_after_last_agglomeration(std::function<void(hipEvent_t)> fn);
bool hc_after_last_agglomeration_fired();

// The same batch in two halves, for a caller that overlaps the front of its NEXT block with the tail of the current one
// (SHARP_unlimited with several blocks per GPU): hc_prefetch_begin uploads the descriptors and enqueues the row preparation and the
// distance GEMM on the stream that is current at the call (StreamScope) into a workspace slot of its own (slot 0 / 1, alternating);
// hc_prefetch_finish makes the main stream wait for that, runs the agglomeration and the statistics and fetches the results --
// exactly what get_opt_hclust_batch does for a batch of one chunk.  Only for batches of at most one task per CU.
struct HcPrefetch;
bool hc_prefetch_possible(const std::vector<HcTask> &tasks);
std::shared_ptr<HcPrefetch> hc_prefetch_begin(std::vector<HcTask> tasks, int slot);
// (optional, before hc_prefetch_finish) enqueues the agglomeration alone and returns an event recorded behind it on the main stream:
// what the caller makes the NEXT block's front wait for, so that it runs under the statistics and the host-bound tail, not beside
// the HBM-bound agglomeration
hipEvent_t hc_prefetch_agglomerate(HcPrefetch &P);
void hc_prefetch_stamp_block(HcPrefetch &P, int block);
void hc_prefetch_finish(HcPrefetch &P, bool want_v, std::vector<HcResult> &out);

}  // namespace sharp
