// hclust.hpp -- batched get_opt_hclust (R/get_opt_hclust.R:33-244) on the GPU: distance build,
// agglomerative tree, cutree for every candidate k, median silhouette, CH index, model selection.
#pragma once
#include <vector>

#include "common.hpp"

namespace sharp {

struct HcParams {
    int hmethod = 1;          // stats::hclust method table, 1 = ward.D
    int N_cluster = 0;        // 0 = NULL: choose k automatically
    int minN = 2, maxN = 40;
    double sil_thre = 0.35;
    double height_Ntimes = 2.0;
};

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

constexpr int kHcMaxN = 7168;   // LDS-resident nearest-neighbour state per task

void get_opt_hclust_batch(const std::vector<HcTask> &tasks, bool want_v, std::vector<HcResult> &out);

}  // namespace sharp
