// meta.hpp -- wMetaC (R/wMetaC.R:15-226) and sMetaC (R/sMetaC.R:17-210) on the GPU.
#pragma once
#include <vector>

#include "hclust.hpp"

namespace sharp {

// ---- wMetaC: one ensemble per task (a fold of SHARP_large, or the whole data in SHARP_small)
struct WmTask {
    const int *nC = nullptr;   // HOST, N x C column-major labels; equality within a column is all that matters
    int N = 0, C = 0;
    HcParams prm;              // prm.N_cluster = enN.cluster
};
struct WmResult {
    int rc = 0;
    std::vector<int> finalC;   // meta-cluster id per cell (the number R keeps as a string)
    int ncl = 0;
    std::vector<double> x0;    // N x ncl column-major soft matrix (want_x0)
    // intermediates (want_debug)
    std::vector<double> w1, S;
    std::vector<int> tf;
    int allC = 0;
};
void wmetac_batch(const std::vector<WmTask> &tasks, bool want_x0, bool want_debug, std::vector<WmResult> &out);

// ---- sMetaC
// Per-label mean rows of a device-resident n x p matrix (R/sMetaC.R:58-63).  uid[i] in [0, nC) is the
// first-appearance index of cell i's label.  d_means: nC x p device buffer (row-major).
// row_of_cell (optional): cell i lives in row row_of_cell[i] of d_E; sums still run in ascending cell order.
void cluster_means_dev(const double *d_E, long long ld, int n, int p, const std::vector<int> &uid, int nC, double *d_means,
                       const int *row_of_cell = nullptr);

struct SmResult {
    int rc = 0;
    std::vector<int> tf;       // meta id per unique input label (first-appearance order)
    double maxsil = 0;
    int optN = 0;
};
// The part of sMetaC after the centroids: S = cor(centroids), k-range adjustment by ncells,
// get_opt_hclust(S), second-best override (R/sMetaC.R:67-151).  d_means: nC x p on the device.
SmResult smetac_from_means(const double *d_means, int nC, int p, long long ncells, HcParams prm);

// enE = sum_k E_k ; viE = enE / K   (R/SHARP.R:398,416,634,750).  d_E: n x (K*p), d_viE: n x p
void ensemble_mean_dev(const double *d_E, long long ldE, int n, int p, int K, double *d_viE);

// first-appearance renumbering of arbitrary int labels: uid[i] in [0, nuniq)
int first_appearance_ids(const int *labels, long long n, std::vector<int> &uid);

}  // namespace sharp
