// dotc.hip -- the entry points of include/sharp_hip.h once more in R's .C() calling convention: every argument a
// pointer (R hands .C() copies of its vectors: double* for numeric, int* for integer/logical), void return, the status in the last
// argument.  With these the reference's exported functions (NAMESPACE:3-28) can call the library from plain R --
//     .C("sharp_C_SHARP", as.double(scExp), nrow(scExp), as.double(ncol(scExp)), ..., status = integer(1))
// -- with no glue compiled against R.h; r/sharp_hip.R holds those R lines, r/sharp_glue.c the .Call shim that avoids .C()'s copies.
// Conventions on top of sharp_hip.h's:
//   - a dimension that can exceed 2^31 - 1 (cells) travels as double*, since R has no 64-bit integer;
//   - R cannot pass NULL: an optional output is a buffer of length >= 1 plus a bit in *want (documented per function);
//   - "missing" scalar arguments are 0 (integers) / negative (sil.thre) like in sharp_hip.h;
//   - *status receives what the plain entry point returns (0, warning bits, or an error code: then sharp_C_last_error()
//     gives the text for R's stop()).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

namespace {

inline long long as_ll(const double *v) { return v ? static_cast<long long>(std::llround(*v)) : 0; }

// R/get_opt_hclust.R:76-83: flashmark = TRUE takes flashClust(d, "ward"), the ward.D criterion, and the test
// `hmethod == "ward.D" || "ward.D2"` that guards it is an error in R for every other method (reference quirk 6)
int flashmark_method(int flashmark, int hmethod, int *status) {
    if (!flashmark) return hmethod;
    const int hm = hmethod > 0 ? hmethod : 1;
    if (hm != 1) {
        sharp::set_error("invalid 'y' type in 'x || y' (flashmark = TRUE with hmethod other than \"ward.D\": R/get_opt_hclust.R:79)");
        *status = SHARP_ERR_ARG;
        return -1;
    }
    return 1;
}

}  // namespace

extern "C" {

void sharp_C_init(int *device, int *status) { *status = sharp_init(*device); }
void sharp_C_shutdown(int *status) { *status = sharp_shutdown(); }
void sharp_C_trim(int *status) { *status = sharp_trim(); }
void sharp_C_reload_options(int *status) { *status = sharp_reload_options(); }
void sharp_C_device_count(int *count, int *status) { *status = sharp_device_count(count); }

/* .C("sharp_C_last_error", msg = paste(rep(" ", 1024), collapse = ""), len = 1024L)$msg : the text is copied into the caller's string */
void sharp_C_last_error(char **msg, int *len) {
    const char *e = sharp_last_error();
    if (!msg || !msg[0] || !len || *len <= 0) return;
    std::strncpy(msg[0], e ? e : "", static_cast<size_t>(*len));
    msg[0][*len - 1] = '\0';
}

/* ---- a1: ranM / ranM2 / RPmat's projector (R/ranM.R:11-33, R/ranM2.R:11-35, R/RPmat.R:14-31) */
void sharp_C_projector_create(int *m, int *p, int *K, double *seeds, int *handle, int *status) {
    *status = sharp_projector_create(*m, *p, *K, seeds, handle);
}
void sharp_C_projector_destroy(int *handle, int *status) { *status = sharp_projector_destroy(*handle); }
/* nnz: in = capacity of gene / col / sign (0: only the count is wanted), out = the number of non-zeros of projector *k */
void sharp_C_projector_triplets(int *handle, int *k, int *gene, int *col, int *sign, double *nnz, int *status) {
    long long nn = 0;
    *status = sharp_projector_triplets(*handle, *k, nullptr, nullptr, nullptr, &nn);
    const long long cap = as_ll(nnz);
    *nnz = static_cast<double>(nn);
    if (*status != SHARP_OK || cap <= 0) return;
    if (cap < nn) { sharp::set_error("sharp_C_projector_triplets: buffers too small"); *status = SHARP_ERR_ARG; return; }
    std::vector<signed char> s(static_cast<size_t>(nn));
    *status = sharp_projector_triplets(*handle, *k, gene, col, s.data(), &nn);
    for (long long i = 0; i < nn; ++i) sign[i] = s[static_cast<size_t>(i)];
}

/* ---- a2: projmat of RPmat (R/RPmat.R:32; R/SHARP.R:343-345,363,579-585).  X: m x n column-major as R holds it; E: n x (K p) row-major
 * = R's (K p) x n matrix */
void sharp_C_project(int *proj, double *X, int *m, int *n, int *log_flag, double *E, int *status) {
    *status = sharp_project(*proj, X, *m, *n, static_cast<long long>(*m), *log_flag, E);
}

/* ---- a3-a5: get_opt_hclust (R/get_opt_hclust.R:33-244).  mat: n x p ROW-major, i.e. t(mat) of the R matrix.
 * want: bit 0 v (n x nk column-major; room for n * (min(maxN, n - 1) - minN + 1)), bit 1 msil/CHind, bit 2 height (n - 1) */
void sharp_C_get_opt_hclust(double *mat, int *n, int *p, int *hmethod, int *N_cluster, int *minN, int *maxN, double *sil_thre,
                            double *height_Ntimes, int *flashmark, int *f, int *v, double *msil, double *CHind, double *maxsil,
                            double *height, int *optN, int *nk, int *branch, int *want, int *status) {
    const int hm = flashmark_method(*flashmark, *hmethod, status);
    if (hm < 0) return;
    const int w = *want;
    *status = sharp_get_opt_hclust(mat, *n, *p, hm, *N_cluster, *minN, *maxN, *sil_thre, *height_Ntimes, f, (w & 1) ? v : nullptr,
                                   (w & 2) ? msil : nullptr, (w & 2) ? CHind : nullptr, maxsil, (w & 4) ? height : nullptr, optN, nk, branch);
}

/* ---- a6: getrowColor (R/getrowColor.R:17-121): rowColor[i] = index into colorL */
void sharp_C_getrowColor(double *E, int *n, int *p, int *hmethod, int *indN_cluster, int *minN, int *maxN, double *sil_thre,
                         double *height_Ntimes, int *flashmark, int *rowColor, double *maxsil, int *status) {
    const int hm = flashmark_method(*flashmark, *hmethod, status);
    if (hm < 0) return;
    *status = sharp_getrowColor(E, *n, *p, hm, *indN_cluster, *minN, *maxN, *sil_thre, *height_Ntimes, rowColor, maxsil);
}

/* ---- a9: wMetaC (R/wMetaC.R:15-226).  nC: N x C column-major integer labels (apply(nC, 2, function(x) match(x, unique(x)))).
 * want bit 0: x0 (N x *ncl column-major; room for N * (maxN + 2)) */
void sharp_C_wMetaC(int *nC, int *N, int *C, int *hmethod, int *enN_cluster, int *minN, int *maxN, double *sil_thre,
                    double *height_Ntimes, int *finalC, double *x0, int *ncl, int *want, int *status) {
    *status = sharp_wMetaC(nC, *N, *C, *hmethod, *enN_cluster, *minN, *maxN, *sil_thre, *height_Ntimes, finalC, (*want & 1) ? x0 : nullptr,
                           ncl, nullptr, nullptr, nullptr, nullptr);
}

/* ---- a10: sMetaC (R/sMetaC.R:17-210).  labels: match(rerowColor, unique(rerowColor)); sE1: n x p ROW-major (t(sE1) of the R matrix);
 * n as double.  tf: room for n entries (the first *nC are written) */
void sharp_C_sMetaC(int *labels, double *sE1, double *n, int *p, int *hmethod, int *finalN_cluster, int *minN, int *maxN,
                    double *sil_thre, double *height_Ntimes, int *finalColor, int *tf, int *nC, int *status) {
    *status = sharp_sMetaC(labels, sE1, as_ll(n), *p, *hmethod, *finalN_cluster, *minN, *maxN, *sil_thre, *height_Ntimes, finalColor, tf, nC);
}

/* ---- a7, a8, a12: SHARP / SHARP_small / SHARP_large (R/SHARP.R:44-318, 339-454, 478-851) on a prepared matrix.
 * X: genes x cells column-major (as.double(scExp)); n as double.  want: bit 0 viE (n x p row-major = R's p x n; room for n * p with
 * p = reduced.ndim or ceiling(log2(n)/0.04)), bit 1 x0 (n x *x0_cols column-major, room for n * *x0_cap_cols).
 * info[5]: n_pred, x0_cols, p_used, K_used, path (0 SHARP_small, 1 SHARP_large). */
void sharp_C_SHARP(double *X, int *m, double *n, int *ensize_K, int *reduced_ndim, int *base_ncells, int *partition_ncells, int *hmethod,
                   int *N_cluster, int *enpN_cluster, int *indN_cluster, int *minN, int *maxN, double *sil_thre, double *height_Ntimes,
                   int *flashmark, int *log_flag, int *projector, double *rN_seed, int *pred, double *viE, double *x0, int *x0_cap_cols,
                   int *info, int *want, int *status) {
    const int hm = flashmark_method(*flashmark, *hmethod, status);
    if (hm < 0) return;
    const int w = *want;
    *status = sharp_SHARP(X, *m, as_ll(n), static_cast<long long>(*m), *ensize_K, *reduced_ndim, *base_ncells, *partition_ncells, hm,
                          *N_cluster, *enpN_cluster, *indN_cluster, *minN, *maxN, *sil_thre, *height_Ntimes, *log_flag, *projector,
                          *rN_seed, pred, &info[0], (w & 1) ? viE : nullptr, (w & 2) ? x0 : nullptr, *x0_cap_cols, &info[1], &info[2],
                          &info[3], &info[4]);
}
/* the same for a Matrix::dgCMatrix: colptr = scExp@p, rowidx = scExp@i, val = scExp@x */
void sharp_C_SHARP_csc(int *colptr, int *rowidx, double *val, int *m, double *n, int *ensize_K, int *reduced_ndim, int *base_ncells,
                       int *partition_ncells, int *hmethod, int *N_cluster, int *enpN_cluster, int *indN_cluster, int *minN, int *maxN,
                       double *sil_thre, double *height_Ntimes, int *flashmark, int *log_flag, int *projector, double *rN_seed, int *pred,
                       double *viE, double *x0, int *x0_cap_cols, int *info, int *want, int *status) {
    const int hm = flashmark_method(*flashmark, *hmethod, status);
    if (hm < 0) return;
    const int w = *want;
    *status = sharp_SHARP_csc(colptr, rowidx, val, *m, as_ll(n), *ensize_K, *reduced_ndim, *base_ncells, *partition_ncells, hm, *N_cluster,
                              *enpN_cluster, *indN_cluster, *minN, *maxN, *sil_thre, *height_Ntimes, *log_flag, *projector, *rN_seed, pred,
                              &info[0], (w & 1) ? viE : nullptr, (w & 2) ? x0 : nullptr, *x0_cap_cols, &info[1], &info[2], &info[3], &info[4]);
}
/* allrpinfo of the last SHARP_small run (R/SHARP.R:350-387,446): dims[3] = n, K, p; want bit 0: enrp (n x K column-major colour
 * indices = the rowColor of every random projection), bit 1: indE (n x (K p) row-major: projection k is columns [k p, (k+1) p)) */
void sharp_C_last_rpinfo(int *dims, int *enrp, double *indE, int *want, int *status) {
    *status = sharp_last_rpinfo(&dims[0], &dims[1], &dims[2], (*want & 1) ? enrp : nullptr, (*want & 2) ? indE : nullptr);
}

/* ---- a11: SHARP_unlimited (R/SHARP_unlimited.R:29-242).  Xcat: the blocks one after the other, each m x ncb[b] column-major
 * (unlist(lapply(scExp, as.double))); ncb as doubles.  want bit 0: viE (ncells x p row-major, p = ceiling(log2(ncells)/0.04)).
 * info[2]: n_pred, p_used */
void sharp_C_SHARP_unlimited(double *Xcat, int *nblocks, double *ncb, int *m, int *ensize_K, int *N_cluster, int *minN, int *maxN,
                             double *rN_seed, int *pred, double *viE, int *info, int *want, int *status) {
    const int B = *nblocks;
    if (B < 1) { sharp::set_error("No expression data is provided!"); *status = SHARP_ERR_ARG; return; }
    std::vector<const double *> ptrs(static_cast<size_t>(B));
    std::vector<long long> nc(static_cast<size_t>(B));
    long long off = 0;
    for (int b = 0; b < B; ++b) { ptrs[b] = Xcat + off * (*m); nc[b] = as_ll(ncb + b); off += nc[b]; }
    *status = sharp_SHARP_unlimited_view(ptrs.data(), nc.data(), B, *m, *ensize_K, *N_cluster, *minN, *maxN, *rN_seed, pred, &info[0],
                                         &info[1], (*want & 1) ? viE : nullptr);
}
/* arms the device-side view reduction for the NEXT sharp_C_SHARP_unlimited* call (sharp_unlimited_view_dim; R/SHARP_unlimited.R:216-228): its viE
 * buffer is then ncells x *kdim */
void sharp_C_unlimited_view_dim(int *kdim, int *status) { *status = sharp_unlimited_view_dim(*kdim); }
void sharp_C_decision_log(int *enable, int *status) { *status = sharp_decision_log(*enable); }
void sharp_C_last_decisions(double *rows, int *cap_rows, int *n_rows, int *status) { *status = sharp_last_decisions(rows, *cap_rows, n_rows); }
/* the same on several GPUs (sharp_SHARP_unlimited_multi): devices = integer vector of device indices, block b on devices[b mod ndev] */
void sharp_C_SHARP_unlimited_multi(double *Xcat, int *nblocks, double *ncb, int *m, int *ensize_K, int *N_cluster, int *minN, int *maxN,
                                   double *rN_seed, int *devices, int *ndevices, int *pred, double *viE, int *info, int *want, int *status) {
    const int B = *nblocks;
    if (B < 1) { sharp::set_error("No expression data is provided!"); *status = SHARP_ERR_ARG; return; }
    std::vector<const double *> ptrs(static_cast<size_t>(B));
    std::vector<long long> nc(static_cast<size_t>(B));
    long long off = 0;
    for (int b = 0; b < B; ++b) { ptrs[b] = Xcat + off * (*m); nc[b] = as_ll(ncb + b); off += nc[b]; }
    *status = sharp_SHARP_unlimited_multi(ptrs.data(), nc.data(), B, *m, *ensize_K, *N_cluster, *minN, *maxN, *rN_seed, devices, *ndevices,
                                          pred, &info[0], &info[1], (*want & 1) ? viE : nullptr);
}
/* a list of dgCMatrix blocks (R/SHARP_unlimited.R:125-135 hands each block to SHARP() as it is; R/SHARP.R:343-345,579 take a dgCMatrix):
 * pcat = the blocks' @p one after the other (ncb[b] + 1 ints each), icat / xcat = their @i / @x one after the other.  *ndevices = 0: the
 * caller's GPU (or SHARP_DEVICES); else block b on devices[b mod *ndevices]. */
void sharp_C_SHARP_unlimited_csc(int *pcat, int *icat, double *xcat, int *nblocks, double *ncb, int *m, int *ensize_K, int *N_cluster,
                                 int *minN, int *maxN, double *rN_seed, int *devices, int *ndevices, int *pred, double *viE, int *info,
                                 int *want, int *status) {
    const int B = *nblocks;
    if (B < 1) { sharp::set_error("No expression data is provided!"); *status = SHARP_ERR_ARG; return; }
    std::vector<const int *> cp(static_cast<size_t>(B)), ri(static_cast<size_t>(B));
    std::vector<const double *> vx(static_cast<size_t>(B));
    std::vector<long long> nc(static_cast<size_t>(B));
    long long poff = 0, eoff = 0;
    for (int b = 0; b < B; ++b) {
        nc[b] = as_ll(ncb + b);
        cp[b] = pcat + poff; ri[b] = icat + eoff; vx[b] = xcat + eoff;
        eoff += static_cast<long long>(cp[b][nc[b]]) - cp[b][0];
        poff += nc[b] + 1;
    }
    *status = sharp_SHARP_unlimited_csc_multi(cp.data(), ri.data(), vx.data(), nc.data(), B, *m, *ensize_K, *N_cluster, *minN, *maxN, *rN_seed,
                                              *ndevices > 0 ? devices : nullptr, *ndevices, pred, &info[0], &info[1], (*want & 1) ? viE : nullptr);
}
/* SHARP_unlimited2 (R/SHARP_unlimited2.R:29-292); same block layout */
void sharp_C_SHARP_unlimited2(double *Xcat, int *nblocks, double *ncb, int *m, int *ensize_K, int *reduced_ndim, int *partition_ncells,
                              int *hmethod, int *N_cluster, int *enpN, int *indN, int *minN, int *maxN, double *sil_thre,
                              double *height_Ntimes, int *flag, double *rN_seed, int *pred, double *viE, int *info, int *want, int *status) {
    const int B = *nblocks;
    if (B < 1) { sharp::set_error("No expression data is provided!"); *status = SHARP_ERR_ARG; return; }
    std::vector<const double *> ptrs(static_cast<size_t>(B));
    std::vector<long long> nc(static_cast<size_t>(B));
    long long off = 0;
    for (int b = 0; b < B; ++b) { ptrs[b] = Xcat + off * (*m); nc[b] = as_ll(ncb + b); off += nc[b]; }
    *status = sharp_SHARP_unlimited2(ptrs.data(), nc.data(), B, *m, *ensize_K, *reduced_ndim, *partition_ncells, *hmethod, *N_cluster, *enpN,
                                     *indN, *minN, *maxN, *sil_thre, *height_Ntimes, *flag, *rN_seed, pred, &info[0], &info[1],
                                     (*want & 1) ? viE : nullptr);
}
/* the merge step of a sharded run (R/SHARP_unlimited.R:163-183) on gathered centroid tables; counts and ncells as doubles */
void sharp_C_unlimited_merge(double *means, double *counts, int *nC, int *p, double *ncells, int *N_cluster, int *minN, int *maxN,
                             int *final_id, int *n_final, int *status) {
    std::vector<long long> cn(static_cast<size_t>(std::max(*nC, 0)));
    for (int q = 0; q < *nC; ++q) cn[q] = as_ll(counts + q);
    *status = sharp_unlimited_merge(means, cn.data(), *nC, *p, as_ll(ncells), *N_cluster, *minN, *maxN, final_id, n_final);
}

/* ---- get_marker_genes' per-gene pass (R/get_marker_genes.R:120-152); out: m x 5 row-major */
void sharp_C_marker_genes(double *X, int *m, double *n, int *label, int *n_cluster, double *theta, int *ng, double *out, int *status) {
    *status = sharp_marker_genes(X, *m, as_ll(n), static_cast<long long>(*m), label, *n_cluster, *theta, *ng, out);
}

}  // extern "C"
