/*
 * sharp_hip.h -- C ABI of libsharp_hip.so: the MI355X (gfx950) implementation of
 * SHARP's ensemble random-projection + meta-clustering hot path.
 *
 * The reference (shibiaowan/SHARP) is pure R with no FFI (NAMESPACE has no
 * useDynLib), so the drop-in boundary is the set of exported R functions
 * (SURVEY.md 8b).  Each entry point below replaces the body of one of them and
 * cites it; INTEGRATION.md shows the R glue (.C / .Call) and the ctypes binding.
 *
 * Conventions
 *   - plain C, pointers + sizes only; no R, torch or C++ types in any signature.
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from sharp_last_error() (reference convention: stop("..."),
 *     R/SHARP.R:52-54,171-176; R/get_opt_hclust.R:91-99).  Nothing aborts.
 *   - the caller owns every host buffer; the library owns device memory and the
 *     state behind integer handles.
 *   - matrices: X is genes x cells, COLUMN-major (a cell is a contiguous m-vector),
 *     exactly R's layout.  E / viE are cells x p ROW-major (= R's p x n column-major
 *     projmat before the transpose at R/SHARP.R:363,583).  Label matrices (enrp, v)
 *     are column-major like R.
 *   - cluster ids are 1-based like R.  "NULL" integer arguments are passed as 0.
 *   - hmethod codes follow stats::hclust's method table:
 *       1 ward.D  2 single  3 complete  4 average  5 mcquitty  6 median  7 centroid  8 ward.D2
 *   - functions ending in _dev take DEVICE pointers (HBM-resident data, e.g. from
 *     hipMalloc or a torch tensor's data_ptr()) and run on the library's stream.
 *   - single-threaded callers (R's main thread); the library never calls back.
 *   - the `sharp_C_*` entry points at the end of this header are the same calls in R's .C()
 *     convention (every argument a pointer, void return, int* status out), so the reference's
 *     R functions can call the library with no glue compiled against R.h (r/sharp_hip.R);
 *     r/sharp_glue.c is the .Call shim that avoids .C()'s argument copies.
 */
#ifndef SHARP_HIP_H
#define SHARP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SHARP_OK 0
#define SHARP_ERR 1          /* generic failure, see sharp_last_error()           */
#define SHARP_ERR_ARG 2      /* argument rejected (what R's stop() checks reject) */
#define SHARP_ERR_NO_DEVICE 3
#define SHARP_WARN_RANGE 16  /* model selection ran off the candidate range (reference quirk 8) */
#define SHARP_WARN_NA_VOTE 32/* wMetaC single-cluster fallback met a cell with one vote value (quirk 11) */

/* ---- runtime -------------------------------------------------------------- */
const char *sharp_last_error(void);
int sharp_version(void);
int sharp_device_count(int *count);
/* Select the GPU and create the library's stream/workspace.  Idempotent per device. */
int sharp_init(int device);
int sharp_shutdown(void);
int sharp_synchronize(void);
/* The library reads its tuning switches (SHARP_* environment variables: kernel forms, chunk sizes, pipeline depths -- none changes a
 * result) ONCE, at its first use; this re-reads them, for a host that changes its environment afterwards (the tests do). */
int sharp_reload_options(void);
/* Per-kernel HIP-event timing on the library's stream (used by bench.py's roofline). */
int sharp_profile_enable(int on);
int sharp_profile_reset(void);
/* name: e.g. "rp_scatter"; returns total milliseconds and launch count since reset. */
int sharp_profile_get(const char *name, double *total_ms, long long *launches);
/* writes a '\n'-separated "name total_ms launches" table into buf (NUL-terminated). */
int sharp_profile_dump(char *buf, int buflen);

/* ---- a1: ranM / ranM2 / projector half of RPmat ---------------------------- */
/* R/ranM.R:11-33, R/ranM2.R:11-35, R/RPmat.R:14-31.
 * Builds K sparse ternary projectors R_k (m x p) with R's own RNG stream:
 * set.seed(seeds[k]); sample(c(sqrt(s),0,-sqrt(s)), m*p, TRUE, c(1/2s,1-1/s,1/2s));
 * byrow fill.  seeds[k] must be integer-valued (50 + rN.seed + k at the call sites
 * R/SHARP.R:360,545, R/SHARP_unlimited.R:101); a non-integer seed (the reference's
 * 0.5 "unseeded" sentinel) draws a seed from the OS entropy source.
 * The projectors are kept on the device as gene-major packed row lists. */
int sharp_projector_create(int m, int p, int K, const double *seeds, int *handle);
int sharp_projector_destroy(int handle);
int sharp_projector_info(int handle, int *m, int *p, int *K, long long *nnz_total);
/* ranM() drop-in: the k-th (0-based) projector as COO triplets sorted row-major
 * (gene, then column): gene[i], col[i] 0-based, sign[i] = +1/-1; value = sign*sqrt(sqrt(m)).
 * Call with gene == NULL to get *nnz only. */
int sharp_projector_triplets(int handle, int k, int *gene, int *col, signed char *sign, long long *nnz);

/* ---- a2: RP matmul  E1 = t( 1/sqrt(p) * t(R_k) %*% log2(X+1) ) -------------- */
/* R/RPmat.R:32, R/SHARP.R:343-345,363,569-571,579-585.
 * Host variant: X double, m x n column-major with leading dimension ld (>= m).
 * E: n x (K*p) row-major, component k*p + c = projector k, column c.
 * X is kept on the device as fp32 when every value is fp32-exact, else as fp64 (sharp_x_storage(),
 * DESIGN.md "Numerics").  log_flag: 1 = log2(x+1) (flag TRUE), 0 = raw. */
int sharp_project(int proj, const double *X, int m, int n, long long ld, int log_flag, double *E);
/* Device variant: dX fp32 (m x n, leading dimension ld elements, 16-byte aligned columns
 * when ld % 4 == 0), dE fp64 n x ldE row-major (ldE >= K*p). */
int sharp_project_dev(int proj, const float *dX, int m, int n, long long ld, int log_flag,
                      double *dE, long long ldE);
/* The same for an fp64 block that is already resident: TPM / CPM-like doubles (the reference's own example data, README.md:88,114;
 * R/SHARP.R:110-117 computes log2(X + 1) and the projection in double), which the fp32 form would perturb by 6e-8 relative.
 * dX must be 16-byte aligned with an even ld >= m and hold finite values (one pass over it finds max |x| and checks that).  The
 * *_dev64 entry points below take the same kind of block. */
int sharp_project_dev64(int proj, const double *dX, int m, int n, long long ld, int log_flag,
                        double *dE, long long ldE);

/* ---- a3-a5: get_opt_hclust --------------------------------------------------- */
/* R/get_opt_hclust.R:33-244.  mat: n x p ROW-major feature rows, or an n x n symmetric similarity
 * (detected like isSymmetric(): square and all.equal(mat, t(mat), 100*eps)); then d = 1 - mat.
 * Feature rows: t(scale(t(mat))), d = 1 - cor(t(mat)) (fp64 MFMA), hclust(d, method), cutree for
 * k = minN..min(maxN, n-1), median silhouette, get_CH(.., "1-corr"), and the reference's choice rule
 * (middle arg-max of msil; CH when max(msil) <= sil_thre; height gap when CH picks the first k).
 * N_cluster: 0 = NULL, else a single cutree(k = N_cluster).
 * Outputs (caller-allocated; any of v, msil, CHind, maxsil, height, optN, nk, branch may be NULL):
 *   f[n] chosen labels (1-based, numbered by first appearance); v[n * nk] column-major;
 *   msil[nk], CHind[nk]; height[n-1]; branch: 0 silhouette, 1 CH, 2 height gap.
 * Returns SHARP_OK, SHARP_WARN_RANGE (choice clamped into the candidate range, reference quirk 8:
 * R would raise "subscript out of bounds") or an error code. */
int sharp_get_opt_hclust(const double *mat, int n, int p, int hmethod, int N_cluster, int minN, int maxN,
                         double sil_thre, double height_Ntimes, int *f, int *v, double *msil, double *CHind,
                         double *maxsil, double *height, int *optN, int *nk, int *branch);

/* ---- a6: getrowColor ----------------------------------------------------------- */
/* ---- the decision log -------------------------------------------------
 * Every choice of a number of clusters on the path is an arg-max over doubles compared with `==` (R/get_opt_hclust.R:162-168: the middle one of
 * the exact ties of the median silhouette; :194-195: which.max(CHind) when max(msil) <= sil.thre; :196-210: the height-gap rule;
 * R/sMetaC.R:139-148: the two-cluster override).  sharp_decision_log(1) (or SHARP_DECISION_LOG=1 in the environment) makes every
 * get_opt_hclust call of the process leave one row of SHARP_DECISION_COLS doubles; sharp_decision_log(0) stops and clears.
 * sharp_last_decisions copies up to cap_rows rows, sorted by (level, block, k, fold), and returns the number held in *n_rows.  Row:
 *  [0] level: 0 base clustering of one projection of one fold (getrowColor, R/SHARP.R:366,592), 1 a fold's wMetaC (R/wMetaC.R:98-99),
 *      2 the sMetaC across a block's folds (R/SHARP.R:754), 3 SHARP_unlimited's sMetaC across blocks (R/SHARP_unlimited.R:163), -1 a direct call
 *  [1] block (index in the SHARP_unlimited list; 0 for SHARP())  [2] k (projection, 0-based)  [3] fold (0-based)  [4] observations
 *  [5] branch: 0 median silhouette, 1 CH, 2 height gap, 3 N.cluster given   [6] chosen number of clusters
 *  [7] exact ties at the deciding maximum  [8] that maximum (msil: branch 0 / 3, CH: 1 / 2)  [9] largest value strictly below it (NaN: none)
 *  [10] max(msil) - sil.thre  [11] height rule, when CH's first level won: gap / ((height.Ntimes - 1) * height) of the deciding step
 *       (branch 2: > 1) or its maximum over the last ten merges (branch 1: <= 1); NaN otherwise
 *  [12] sMetaC's two-cluster override: the number of clusters of the column taken instead (0: not applied)  [13] candidate levels
 * oracle/sharp_oracle.c writes the same rows (oracle_decision_log / oracle_last_decisions): tests compare the two logs entry by entry. */
#define SHARP_DECISION_COLS 14
int sharp_decision_log(int enable);
int sharp_last_decisions(double *rows /* cap_rows x SHARP_DECISION_COLS */, int cap_rows, int *n_rows);

/* R/getrowColor.R:17-121.  rowColor[i] in 1..40 is the index into the reference's colorL table
 * (cluster j > 40 wraps and collides exactly like :59-68); height_Ntimes <= 0 -> 1 (:28-30). */
int sharp_getrowColor(const double *E, int n, int p, int hmethod, int indN_cluster, int minN, int maxN,
                      double sil_thre, double height_Ntimes, int *rowColor, double *maxsil);

/* ---- a9: wMetaC ------------------------------------------------------------------ */
/* R/wMetaC.R:15-226 (+ getA :242-283, getss :299-311, getnewk :313-320).
 * nC: N x C column-major integer labels (the reference's strings only ever compare for equality
 * within a column).  finalC[N]: meta-cluster id per cell (the number R stores as a string; ties in the
 * vote go to the id whose decimal string sorts first, like names(sort(table(d), decreasing=TRUE)[1])).
 * x0 (optional): N x *ncl column-major soft matrix (caller buffer of N * min(maxN, allC-1) doubles).
 * Optional intermediates for stage-wise checks: w1_out[N], S_out[allC*allC], tf_out[allC], *allC_out.
 * Returns SHARP_OK, or warning bits SHARP_WARN_RANGE / SHARP_WARN_NA_VOTE, or an error code. */
int sharp_wMetaC(const int *nC, int N, int C, int hmethod, int enN_cluster, int minN, int maxN, double sil_thre,
                 double height_Ntimes, int *finalC, double *x0, int *ncl, double *w1_out, double *S_out, int *allC_out,
                 int *tf_out);

/* ---- a10: sMetaC ----------------------------------------------------------------- */
/* R/sMetaC.R:17-210.  labels[n]: integers, equal <=> same label string; sE1: n x p row-major.
 * finalColor[n]: meta id per cell; tf_out[nC] (optional): meta id per unique label in
 * first-appearance order; `folds` is unused by the reference and has no counterpart here. */
int sharp_sMetaC(const int *labels, const double *sE1, long long n, int p, int hmethod, int finalN_cluster, int minN,
                 int maxN, double sil_thre, double height_Ntimes, int *finalColor, int *tf_out, int *nC_out);

/* ---- a7, a8, a12: SHARP / SHARP_small / SHARP_large -------------------------------- */
/* R/SHARP.R:44-318 (front door), :339-454 (SHARP_small), :478-851 (SHARP_large), for a matrix that
 * already went through the host-side preparation (dedupe, prep, CPM: the Python/R glue does those).
 * Arguments <= 0 (sil_thre < 0) take the reference defaults: ensize_K 15 (small) / 5 (large),
 * reduced_ndim ceiling(log2(n)/0.2^2), base_ncells 5000, partition_ncells 2000, hmethod ward.D,
 * minN 2, maxN max(40, ceiling(n/5000)), sil_thre 0.35, height_Ntimes 2.  *_cluster: 0 = NULL.
 * log_flag: the `flag` of R/SHARP.R:211-228 (1 = log2(x+1)).  projector: handle of a shared `rM` list
 * (sharp_projector_create) or 0 to draw it from rN_seed (seeds 50 + rN_seed + k); rN_seed = 0.5 is the
 * reference's "not reproducible" sentinel.  n < base_ncells runs SHARP_small, else SHARP_large
 * (with N_cluster given and n < base_ncells the reference's reroute at :181-191 applies).
 * Outputs: pred[n] (1..*n_pred, numbered by first appearance); viE (optional) n x p row-major;
 * x0 (optional) n x *x0_cols column-major with room for x0_cap_cols columns; *path 0 small / 1 large. */
int sharp_SHARP_dev(const float *dX, int m, long long n, long long ld, int ensize_K, int reduced_ndim, int base_ncells,
                    int partition_ncells, int hmethod, int N_cluster, int enpN_cluster, int indN_cluster, int minN, int maxN,
                    double sil_thre, double height_Ntimes, int log_flag, int projector, double rN_seed, int *pred,
                    int *n_pred, double *viE, double *x0, int x0_cap_cols, int *x0_cols, int *p_used, int *K_used, int *path);
/* a resident fp64 block (see sharp_project_dev64) */
int sharp_SHARP_dev64(const double *dX, int m, long long n, long long ld, int ensize_K, int reduced_ndim, int base_ncells,
                      int partition_ncells, int hmethod, int N_cluster, int enpN_cluster, int indN_cluster, int minN, int maxN,
                      double sil_thre, double height_Ntimes, int log_flag, int projector, double rN_seed, int *pred,
                      int *n_pred, double *viE, double *x0, int x0_cap_cols, int *x0_cols, int *p_used, int *K_used, int *path);
/* host matrix: X double, m x n column-major (ld >= m); kept in HBM as fp32 when that is exact, else as fp64 (sharp_x_storage) */
int sharp_SHARP(const double *X, int m, long long n, long long ld, int ensize_K, int reduced_ndim, int base_ncells,
                int partition_ncells, int hmethod, int N_cluster, int enpN_cluster, int indN_cluster, int minN, int maxN,
                double sil_thre, double height_Ntimes, int log_flag, int projector, double rN_seed, int *pred,
                int *n_pred, double *viE, double *x0, int x0_cap_cols, int *x0_cols, int *p_used, int *K_used, int *path);

/* How the most recent host-matrix entry point (sharp_SHARP, sharp_SHARP_csc, sharp_SHARP_unlimited*, sharp_project) stored its block
 * in HBM: 32 = fp32, chosen when every value survives the round trip through float (counts, UMI data: half the bytes of the one pass
 * over X); 64 = fp64 (TPM / CPM-like doubles), so that log2(X + 1) and the projection see the numbers the reference computes with
 * (R/SHARP.R:110-117,343-345).  0: nothing uploaded yet.  The environment variable SHARP_X_STORAGE = fp32 | fp64 forces the choice. */
int sharp_x_storage(void);
/* what a value of the most recently uploaded host block was on PCIe: 8 or 16 (counts: every value an integer in 0 .. 255 / 0 .. 65535, sent
 * as unsigned integers of that width and stored as fp32), 32 (other fp32-exact values), 64 (doubles).  A sparse block also sends 16-bit row
 * indices when it has at most 65 536 genes: 3 bytes per non-zero for typical count data against the 12 of R's dgCMatrix slots
 * (R/SHARP_unlimited.R:125-143 hands the blocks over as they are). */
int sharp_x_wire(void);

/* allrpinfo of the most recent call that took the SHARP_small path (R/SHARP.R:350-387,446: per random projection k its tag, the
 * rowColor of every cell, N.cluster and indE = the projected matrix): enrp n x K column-major colour indices (1..40, the index into the
 * reference's colorL), indE n x (K p) row-major with projection k in columns [k p, (k+1) p).  Any output may be NULL (sizes only).
 * Valid until the next sharp_SHARP* call; an error if the last call took the SHARP_large path (the reference returns no allrpinfo there). */
int sharp_last_rpinfo(int *n, int *K, int *p, int *enrp, double *indE);

/* Releases the resident copy of the last host matrix, the pinned staging buffers that sharp_SHARP / sharp_SHARP_csc keep
 * between calls and the projections of the last batched sharp_SHARP_unlimited window; the worker and helper slots that in-process
 * multi-GPU runs and batched windows have left behind also give back their clustering workspaces (the analogue of R's gc() after a
 * run; no reference counterpart). */
int sharp_trim(void);

/* Sparse input: the reference takes whatever `log2(scExp + 1)` and `%*%` accept (R/SHARP.R:343-345,579), which includes the
 * Matrix package's dgCMatrix -- the usual container of scRNA-seq counts.  colptr = @p (n + 1 ints), rowidx = @i (0-based),
 * val = @x; canonical CSC (no duplicated entries).  Only the non-zeros cross PCIe; the dense block (fp32 or fp64 like the dense
 * entry points choose, sharp_x_storage) is built on the device, so results are bit-identical to the dense entry points.
 * sharp_csc_to_dense_dev fills a caller-owned device block (m x n fp32, column stride ld >= m; values narrowed to fp32) for the
 * *_dev entry points. */
int sharp_csc_to_dense_dev(const int *colptr, const int *rowidx, const double *val, int m, long long n, float *dX, long long ld);
/* The same for a block that is ALREADY PACKED and ALREADY IN DEVICE MEMORY: d_colptr n + 1 int64, d_idx row indices of idx_bits (16 or 32),
 * d_val values of val_bits (8 / 16: unsigned integers, 32: float, 64: double) -- what a block file of the compact format holds
 * (sharp_amd/blocks.py; the directory-of-partitions input of R/SHARP_unlimited3.R:59-62,103-105), DMA'd as it is.  dX: m x n, column stride
 * ld, fp32 or (dx_is_f64) fp64; zeroed and filled on the library's stream. */
int sharp_csc_packed_expand_dev(const long long *d_colptr, const void *d_idx, int idx_bits, const void *d_val, int val_bits, int m,
                                long long n, void *dX, long long ld, int dx_is_f64);
int sharp_SHARP_csc(const int *colptr, const int *rowidx, const double *val, int m, long long n, int ensize_K, int reduced_ndim,
                    int base_ncells, int partition_ncells, int hmethod, int N_cluster, int enpN_cluster, int indN_cluster,
                    int minN, int maxN, double sil_thre, double height_Ntimes, int log_flag, int projector, double rN_seed,
                    int *pred, int *n_pred, double *viE, double *x0, int x0_cap_cols, int *x0_cols, int *p_used, int *K_used,
                    int *path);

/* ---- a11: SHARP_unlimited ---------------------------------------------------------- */
/* R/SHARP_unlimited.R:29-242.  Whole call on one GPU: blocks are device (or host) matrices sharing m
 * genes; p = ceiling(log2(sum ncb)/0.04); shared projectors; per-block SHARP(); cross-block sMetaC on
 * the per-(block, cluster) centroid means of viE; < 10-cell merge; ids by decreasing size. */
int sharp_SHARP_unlimited_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                              int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred,
                              int *p_used);
int sharp_SHARP_unlimited(const double *const *X_blocks, const long long *ncb, int nblocks, int m, int ensize_K,
                          int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred, int *p_used);
/* The same over several GPUs of this node, inside the calling process: the serial block loop of R/SHARP_unlimited.R:125-163
 * is dealt out, block b to devices[b mod ndevices]; one host thread per device (each with its own context, streams and workspaces) builds
 * the projectors -- a pure function of m, p and the seeds, :97-104 -- and clusters its blocks; p comes from the GLOBAL cell count (:65-66);
 * the per-(block, cluster) centroid tables (a few hundred rows x p doubles per block: all sMetaC uses of E1, R/sMetaC.R:58-63) meet in
 * host memory, the final sMetaC / small-cluster merge / size-ordered relabel (:163-183) run once, on devices[0], and every block's labels
 * are mapped through the result.  Labels identical to sharp_SHARP_unlimited on one GPU.  A device may be named more than once (several
 * slots on one GPU: the tests), and successive calls may name different lists (a worker's context is found by device, not by position).
 * rN_seed must be a seed (0.5, the unseeded sentinel, would give every device different projectors).  Each GPU has a second host thread
 * that uploads block b + ndevices while block b is clustered.  sharp_SHARP_unlimited / _view themselves take this path when the
 * environment names several devices (SHARP_DEVICES=0,1,2,...) and the call is seeded; an unseeded call stays on the caller's GPU.
 * sharp_trim() / sharp_shutdown() release what every worker context keeps between calls, on every GPU. */
int sharp_SHARP_unlimited_multi(const double *const *X_blocks, const long long *ncb, int nblocks, int m, int ensize_K,
                                int N_cluster, int minN, int maxN, double rN_seed, const int *devices, int ndevices,
                                int *pred, int *n_pred, int *p_used, double *viE /* ncells x p row-major, or NULL */);
/* A list of SPARSE blocks (R/SHARP_unlimited.R:125-135 hands each block to SHARP() as it is, and log2(scExp + 1) / %*% take a
 * Matrix::dgCMatrix, R/SHARP.R:343-345,579): colptr[b] = block b's @p (ncb[b] + 1 ints), rowidx[b] = @i (0-based), val[b] = @x.  Only
 * a block's non-zeros cross PCIe (8 bytes each for count data); csc_expand_kernel builds the dense block on its GPU, the very block the
 * dense entry builds, so labels and viE are those of sharp_SHARP_unlimited_multi on the same values.  Block b + W is uploaded (second
 * host thread per GPU, own stream and pinned staging, two resident copies in rotation) while block b is clustered.  devices = NULL /
 * ndevices = 0 (and sharp_SHARP_unlimited_csc): the caller's GPU, or the SHARP_DEVICES list. */
int sharp_SHARP_unlimited_csc(const int *const *colptr, const int *const *rowidx, const double *const *val, const long long *ncb,
                              int nblocks, int m, int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred,
                              int *n_pred, int *p_used, double *viE /* or NULL */);
int sharp_SHARP_unlimited_csc_multi(const int *const *colptr, const int *const *rowidx, const double *const *val, const long long *ncb,
                                    int nblocks, int m, int ensize_K, int N_cluster, int minN, int maxN, double rN_seed,
                                    const int *devices, int ndevices, int *pred, int *n_pred, int *p_used, double *viE /* or NULL */);
/* Blocks ALREADY resident on the GPUs of the device list (the foreach loop of R/SHARP_unlimited.R:125-163 over data that never leaves
 * HBM): block b is an m x ncb[b] column-major matrix (column stride ldb[b]; fp32, or fp64 where is_f64[b] != 0: 16-byte aligned, even
 * stride) on devices[device_of_block[b]].  Every GPU's worker runs its blocks in list order, each block's RP stage and first distance
 * matrices prepared under the previous block's tail.  is_f64 = NULL: all fp32. */
int sharp_SHARP_unlimited_multi_dev(const void *const *dX_blocks, const int *is_f64, const long long *ncb, const long long *ldb,
                                    const int *device_of_block, int nblocks, int m, int ensize_K, int N_cluster, int minN, int maxN,
                                    double rN_seed, const int *devices, int ndevices, int *pred, int *n_pred, int *p_used,
                                    double *viE /* or NULL */);
/* What the most recent in-process multi-device call did when: one row per block -- worker, block, upload start, upload end, clustering
 * start, clustering end (seconds since the call began; the upload columns are 0 for a resident block).  rows: room for cap_rows x 6. */
int sharp_multi_timeline(double *rows, int cap_rows, int *nrows);
/* The same with the viewflag output (R/SHARP_unlimited.R:153,216-228): viE = the blocks' ensemble-mean projections E1,
 * ncells x p row-major in block order (NULL: not wanted).  The 50-dimension reduction the reference applies above 1e5
 * cells is one more sharp_project() call on this matrix (host side: sharp_amd/api.py). */
int sharp_SHARP_unlimited_view(const double *const *X_blocks, const long long *ncb, int nblocks, int m, int ensize_K,
                               int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred, int *p_used,
                               double *viE);
int sharp_SHARP_unlimited_view_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                                   int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred,
                                   int *p_used, double *viE);
/* viewflag above 1e5 cells (R/SHARP_unlimited.R:216-228): enresults$viE is not E1 but 1/sqrt(kdim) * E1 %*% ranM2(p, kdim, seed), kdim = 50.
 * Every block's product is taken on its GPU as soon as the block's viE exists and the caller's viE receives ncells x kdim doubles
 * (row-major) instead of ncells x p.  sharp_SHARP_unlimited_viewk_dev carries kdim as an argument.  For the other entries that take a viE
 * buffer sharp_unlimited_view_dim(kdim) arms it for the NEXT SHARP_unlimited call OF THE CALLING THREAD (one-shot and thread-local: a call
 * made by another thread never takes it; the call that takes it keeps kdim and its seed as state of its own, so overlapping calls of one
 * process do not see each other).  Arm it immediately before the call; kdim = 0 disarms.  ONE z0 per call: seed 50 + rN.seed + ensize.K + 1
 * (the expression of :222 reads an undefined `k`, an R error whenever rN.seed is given: taken as the next seed of the projector sequence),
 * or, for an unseeded call, one random integer seed drawn when the call begins and used for every block, helper thread and device (:219-225
 * draw z0 once and multiply all of E1 by it). */
int sharp_unlimited_view_dim(int kdim);
int sharp_SHARP_unlimited_viewk_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                                    int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred,
                                    int *p_used, int view_dim, double *viE /* ncells x (view_dim > 0 ? view_dim : p) row-major */);
/* The same split for one-block-per-GPU sharding: every rank runs its blocks through
 * sharp_unlimited_block_dev (block labels 1..*n_clusters by first appearance, the cluster means of viE,
 * n_clusters x p row-major into `means` with room for cap_rows rows, and the cluster sizes), the
 * (means, counts) tables are all-gathered in block order, and every rank calls sharp_unlimited_merge,
 * which returns the final 1-based id of each gathered (block, cluster) row. */
int sharp_unlimited_block_dev(const float *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                              double rN_seed, int *pred, int *n_clusters, double *means, int cap_rows, long long *counts);
int sharp_unlimited_block_dev64(const double *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                                double rN_seed, int *pred, int *n_clusters, double *means, int cap_rows, long long *counts);
/* Several resident blocks of ONE rank in one call: what sharp_unlimited_block_dev returns for each of them (labels back to back in
 * `pred`, n_clusters[b] rows of `means` / `counts` per block, back to back; cap_rows: room for all of them), with the base clustering
 * of all blocks as one pipelined batch and the blocks' tails on helper threads (R/SHARP_unlimited.R:125-149: the loop over blocks a rank
 * owns; 17 instead of 25 ms per 50 000-cell block).  is_f64[b] != 0: block b holds doubles (NULL: all fp32). */
int sharp_unlimited_blocks_dev(const void *const *dX_blocks, const int *is_f64, const long long *ncb, const long long *ldb, int nblocks, int m,
                               int p, int projector, int ensize_K, double rN_seed, int *pred, int *n_clusters, double *means, int cap_rows,
                               long long *counts);
/* SHARP_unlimited2 (R/SHARP_unlimited2.R:29-292, with SHARP_fpart :297-544): log10 instead of log2, E1 rounded to one
 * decimal before the base clustering (maxN.cluster = 40 there), and a single sMetaC over the per-fold ensemble clusters of
 * all blocks.  flag: log-transform (the reference's testlog decision); viE: ncells x p row-major E1 or NULL.  0 / negative
 * parameters take the reference defaults (:39-69). */
int sharp_SHARP_unlimited2(const double *const *X_blocks, const long long *ncb, int nblocks, int m, int ensize_K,
                           int reduced_ndim, int partition_ncells, int hmethod, int N_cluster, int enpN, int indN, int minN,
                           int maxN, double sil_thre, double height_Ntimes, int flag, double rN_seed, int *pred, int *n_pred,
                           int *p_used, double *viE);
int sharp_SHARP_unlimited2_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                               int ensize_K, int reduced_ndim, int partition_ncells, int hmethod, int N_cluster, int enpN,
                               int indN, int minN, int maxN, double sil_thre, double height_Ntimes, int flag, double rN_seed,
                               int *pred, int *n_pred, int *p_used, double *viE);
/* The block step with the log flag of the per-block SHARP() call (SHARP_unlimited3 leaves it to testlog,
 * R/SHARP_unlimited3.R:122) and, optionally, the block's viE rows (nb x p row-major, host; NULL: not wanted). */
int sharp_unlimited_block_view_dev(const float *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                                   double rN_seed, int flag, int *pred, int *n_clusters, double *means, int cap_rows,
                                   long long *counts, double *viE);
/* The same with the view reduction as arguments (a caller that runs its blocks one call at a time, e.g. a rank of the sharded run): viE
 * receives nb x view_dim doubles (view_dim = 0: nb x p); view_seed: the integer seed of the RUN's z0 = ranM2(p, view_dim, view_seed),
 * the same for every block and rank (50 + rN.seed + ensize.K + 1, or one random integer per run when the run is unseeded). */
int sharp_unlimited_block_viewk_dev(const float *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                                    double rN_seed, int flag, int *pred, int *n_clusters, double *means, int cap_rows,
                                    long long *counts, int view_dim, double view_seed, double *viE);
/* ... and for a resident fp64 block (values fp32 cannot hold: 16-byte aligned, even leading dimension) */
int sharp_unlimited_block_viewk_dev64(const double *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                                      double rN_seed, int flag, int *pred, int *n_clusters, double *means, int cap_rows,
                                      long long *counts, int view_dim, double view_seed, double *viE);
int sharp_unlimited_merge(const double *means, const long long *counts, int nC, int p, long long ncells, int N_cluster,
                          int minN, int maxN, int *final_id, int *n_final);
/* One-shot hint for a caller that runs its blocks one call at a time (a rank of the sharded run with several blocks per GPU): the block
 * of the call AFTER the next one (same genes, projector, parameters), already resident in HBM.  The next sharp_unlimited_block_view_dev
 * call enqueues that block's projection and distance matrices on a side stream under its own host-bound tail (what
 * sharp_SHARP_unlimited_dev does between its blocks); the call after it finds them done.  NULL clears the hint.  Results do not change. */
int sharp_unlimited_next_block_dev(const float *dX_next, long long nb_next, long long ld_next);

/* ---- get_marker_genes (R/get_marker_genes.R:25-264), the per-gene pass :120-152 ---------------- */
/* X genes x cells column-major (host, leading dimension ld) or dX fp32 on the device; label[n] in 1..n_cluster
 * (y$pred_clusters after the match() of :96-99).  out: m x 5 row-major = auc, icluster, pvalue (before p.adjust),
 * sparsity, FC for every gene; genes with sparsity <= theta get (0, 0, 1, sparsity, 0) like :146-149.  ng: how many
 * of the clusters with the highest mean rank are tried (:118).  Filtering, Holm adjustment and ordering (:153-187)
 * are host work (sharp_amd/api.py). */
int sharp_marker_genes(const double *X, int m, long long n, long long ld, const int *label, int n_cluster, double theta, int ng,
                       double *out);
int sharp_marker_genes_dev(const float *dX, int m, long long n, long long ld, const int *label, int n_cluster, double theta, int ng,
                           double *out);
/* The same per-gene pass over the cells of a LIST of blocks, as get_marker_genes_unlimited runs it on the list SHARP_unlimited
 * clustered (R/get_marker_genes_unlimited.R:95-118: a gene's values across every block, one rank over all cells; ng = 1 there) and as
 * get_marker_genes_unlimited2 runs it (R/get_marker_genes_unlimited2.R:152-190, ng = min(10, N.cluster)).  label: the labels of all
 * cells in block order.  _dev: resident fp32 blocks (m x ncb[b], column stride ldb[b]); _csc: dgCMatrix blocks on the host
 * (colptr[b] = @p, rowidx[b] = @i, val[b] = @x) -- the stored entries go over as they are and are scattered straight into the
 * per-gene lists of non-zero cells, no dense block is built.  Values are compared as fp32, like sharp_marker_genes. */
int sharp_marker_genes_blocks_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                                  const int *label, int n_cluster, double theta, int ng, double *out);
int sharp_marker_genes_blocks_csc(const int *const *colptr, const int *const *rowidx, const double *const *val, const long long *ncb,
                                  int nblocks, int m, const int *label, int n_cluster, double theta, int ng, double *out);

/* ---- synthetic inputs (bench / tests; not part of the reference) ------------ */
/* Counter-based generator, value = f(seed, gene, cell): bit-identical to
 * oracle_synth_value().  Fills dX (fp32, m x ncell column-major, leading dim ld). */
int sharp_synth_fill_dev(unsigned seed, int m, long long cell0, int ncell, int G, int nmark,
                         float *dX, long long ld);
int sharp_synth_labels(unsigned seed, long long cell0, int ncell, int G, int *labels);

/* Test entry for the fp64 MFMA GEMM kernels behind the correlation distance and the per-cluster sums (tests/test_linalg_gpu.py):
 * C (M x N row-major) = sum_k At[k][i] * Bt[k][j], At K x M and Bt K x N row-major host arrays.  epilogue: 0 plain, 1 = 1 - clamp(v)
 * with a zero diagonal (correlation distance), 2 = clamp(v) with a unit diagonal; symmetric: Bt is ignored (C = At^T At, upper
 * triangle computed and mirrored); fast: the 128 x 128-tile kernel on zero-padded copies, else the generic 64 x 64 kernel. */
int sharp_gemm_tn_f64(const double *At, const double *Bt, double *C, int M, int N, int K, int epilogue, int symmetric, int fast);

/* ---- device memory helpers for non-torch hosts (R glue, tests) -------------- */
int sharp_dev_alloc(long long bytes, void **dptr);
int sharp_dev_free(void *dptr);
int sharp_dev_upload(void *dptr, const void *host, long long bytes);
int sharp_dev_download(void *host, const void *dptr, long long bytes);

/* ---- the same entry points in R's .C() calling convention (sharp_amd/csrc/dotc.hip) -------------------------------------------
 * What R's  .C("sharp_C_SHARP", as.double(scExp), nrow(scExp), as.double(ncol(scExp)), ..., status = integer(1))  binds: every
 * argument is a pointer into a vector R owns (double* for numeric, int* for integer / logical, char** for character), the return is
 * void and *status receives what the plain entry point returns.  Dimensions that can exceed 2^31 - 1 (numbers of cells) travel as
 * double.  R cannot pass NULL, so optional outputs are buffers of length >= 1 selected by bits of *want.  "Missing" arguments are 0
 * (negative for sil.thre) as above.  `flashmark` (R/get_opt_hclust.R:76-83): TRUE takes flashClust(d, "ward") = the ward.D criterion;
 * with any other hmethod the reference's test `hmethod == "ward.D" || "ward.D2"` is an R error, reproduced as SHARP_ERR_ARG.
 * r/sharp_hip.R holds the R side of every one of these. */
void sharp_C_init(int *device, int *status);
void sharp_C_shutdown(int *status);
void sharp_C_trim(int *status);
void sharp_C_reload_options(int *status);   /* sharp_reload_options(): after Sys.setenv() of a SHARP_* switch in a session that has already called sharp_C_init */
void sharp_C_device_count(int *count, int *status);
void sharp_C_last_error(char **msg, int *len);                     /* copies the message into the caller's string of *len bytes */
/* R/ranM.R:11-33, R/ranM2.R:11-35, R/RPmat.R:14-31 */
void sharp_C_projector_create(int *m, int *p, int *K, double *seeds, int *handle, int *status);
void sharp_C_projector_destroy(int *handle, int *status);
/* nnz: in = capacity of gene/col/sign (0: count only), out = non-zeros of projector *k; sign[i] = +1 / -1 */
void sharp_C_projector_triplets(int *handle, int *k, int *gene, int *col, int *sign, double *nnz, int *status);
/* R/RPmat.R:32: X m x n column-major, E n x (K p) row-major */
void sharp_C_project(int *proj, double *X, int *m, int *n, int *log_flag, double *E, int *status);
/* R/get_opt_hclust.R:33-244; mat n x p ROW-major; want: 1 v, 2 msil + CHind, 4 height */
void sharp_C_get_opt_hclust(double *mat, int *n, int *p, int *hmethod, int *N_cluster, int *minN, int *maxN, double *sil_thre,
                            double *height_Ntimes, int *flashmark, int *f, int *v, double *msil, double *CHind, double *maxsil,
                            double *height, int *optN, int *nk, int *branch, int *want, int *status);
/* R/getrowColor.R:17-121 */
void sharp_C_getrowColor(double *E, int *n, int *p, int *hmethod, int *indN_cluster, int *minN, int *maxN, double *sil_thre,
                         double *height_Ntimes, int *flashmark, int *rowColor, double *maxsil, int *status);
/* R/wMetaC.R:15-226; want: 1 x0 (room for N * (maxN + 2)) */
void sharp_C_wMetaC(int *nC, int *N, int *C, int *hmethod, int *enN_cluster, int *minN, int *maxN, double *sil_thre,
                    double *height_Ntimes, int *finalC, double *x0, int *ncl, int *want, int *status);
/* R/sMetaC.R:17-210; n as double; tf: room for n entries */
void sharp_C_sMetaC(int *labels, double *sE1, double *n, int *p, int *hmethod, int *finalN_cluster, int *minN, int *maxN,
                    double *sil_thre, double *height_Ntimes, int *finalColor, int *tf, int *nC, int *status);
/* R/SHARP.R:44-318 (and :339-454, :478-851); want: 1 viE, 2 x0; info[5] = n_pred, x0_cols, p_used, K_used, path */
void sharp_C_SHARP(double *X, int *m, double *n, int *ensize_K, int *reduced_ndim, int *base_ncells, int *partition_ncells, int *hmethod,
                   int *N_cluster, int *enpN_cluster, int *indN_cluster, int *minN, int *maxN, double *sil_thre, double *height_Ntimes,
                   int *flashmark, int *log_flag, int *projector, double *rN_seed, int *pred, double *viE, double *x0, int *x0_cap_cols,
                   int *info, int *want, int *status);
void sharp_C_SHARP_csc(int *colptr, int *rowidx, double *val, int *m, double *n, int *ensize_K, int *reduced_ndim, int *base_ncells,
                       int *partition_ncells, int *hmethod, int *N_cluster, int *enpN_cluster, int *indN_cluster, int *minN, int *maxN,
                       double *sil_thre, double *height_Ntimes, int *flashmark, int *log_flag, int *projector, double *rN_seed, int *pred,
                       double *viE, double *x0, int *x0_cap_cols, int *info, int *want, int *status);
/* allrpinfo (R/SHARP.R:350-387,446); dims[3] = n, K, p; want: 1 enrp, 2 indE */
void sharp_C_last_rpinfo(int *dims, int *enrp, double *indE, int *want, int *status);
/* R/SHARP_unlimited.R:29-242; Xcat: the blocks one after the other (each m x ncb[b] column-major), ncb as doubles; want: 1 viE;
 * info[2] = n_pred, p_used */
void sharp_C_SHARP_unlimited(double *Xcat, int *nblocks, double *ncb, int *m, int *ensize_K, int *N_cluster, int *minN, int *maxN,
                             double *rN_seed, int *pred, double *viE, int *info, int *want, int *status);
/* sharp_SHARP_unlimited_multi: devices = integer vector of GPU indices (block b on devices[b mod *ndevices]) */
void sharp_C_unlimited_view_dim(int *kdim, int *status);      /* sharp_unlimited_view_dim: the next sharp_C_SHARP_unlimited* call's viE is ncells x *kdim */
void sharp_C_decision_log(int *enable, int *status);          /* sharp_decision_log */
void sharp_C_last_decisions(double *rows, int *cap_rows, int *n_rows, int *status);   /* sharp_last_decisions */
void sharp_C_SHARP_unlimited_multi(double *Xcat, int *nblocks, double *ncb, int *m, int *ensize_K, int *N_cluster, int *minN, int *maxN,
                                   double *rN_seed, int *devices, int *ndevices, int *pred, double *viE, int *info, int *want, int *status);
/* sharp_SHARP_unlimited_csc_multi for a list of dgCMatrix blocks: pcat / icat / xcat = the blocks' @p / @i / @x one after the other;
 * *ndevices = 0: the caller's GPU (R/SHARP_unlimited.R:125-135, R/SHARP.R:343-345,579) */
void sharp_C_SHARP_unlimited_csc(int *pcat, int *icat, double *xcat, int *nblocks, double *ncb, int *m, int *ensize_K, int *N_cluster,
                                 int *minN, int *maxN, double *rN_seed, int *devices, int *ndevices, int *pred, double *viE, int *info,
                                 int *want, int *status);
/* R/SHARP_unlimited2.R:29-292 */
void sharp_C_SHARP_unlimited2(double *Xcat, int *nblocks, double *ncb, int *m, int *ensize_K, int *reduced_ndim, int *partition_ncells,
                              int *hmethod, int *N_cluster, int *enpN, int *indN, int *minN, int *maxN, double *sil_thre,
                              double *height_Ntimes, int *flag, double *rN_seed, int *pred, double *viE, int *info, int *want, int *status);
/* R/SHARP_unlimited.R:163-183 on gathered centroid tables; counts, ncells as doubles */
void sharp_C_unlimited_merge(double *means, double *counts, int *nC, int *p, double *ncells, int *N_cluster, int *minN, int *maxN,
                             int *final_id, int *n_final, int *status);
/* R/get_marker_genes.R:120-152 */
void sharp_C_marker_genes(double *X, int *m, double *n, int *label, int *n_cluster, double *theta, int *ng, double *out, int *status);

#ifdef __cplusplus
}
#endif
#endif /* SHARP_HIP_H */
