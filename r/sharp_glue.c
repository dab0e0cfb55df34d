/* sharp_glue.c -- .Call shim between R and libsharp_hip.so (include/sharp_hip.h) for the reference package's hot path.
 *
 *     R CMD SHLIB sharp_glue.c -I../include -L../sharp_amd -lsharp_hip          (needs R's headers: not built in the build image,
 *                                                                                which has no R; the .C() route in sharp_hip.R
 *                                                                                needs no compiled glue at all)
 *
 * Unlike .C(), .Call passes R's vectors by reference: the 8 GB of a 50 000 x 20 000 numeric matrix are read in place by the
 * library's threaded upload (sharp_SHARP) instead of being duplicated first.  Errors become R errors (the reference's stop()),
 * the two warning bits R warnings.  Called from R's main thread only; the library never calls back into R.
 * Parameter vectors (built in r/sharp_hip.R::.sharp_run):
 *   ipar = ensize.K, reduced.ndim, base.ncells, partition.ncells, hmethod, N.cluster, enpN.cluster, indN.cluster, minN.cluster,
 *          maxN.cluster, flashmark, flag (log transform), rM handle          (0 = missing -> the reference default inside the library)
 *   dpar = sil.thre (< 0 = missing), height.Ntimes (0 = missing), rN.seed (0.5 = the reference's "not reproducible" sentinel) */
#include <math.h>
#include <string.h>

#include <R.h>
#include <Rinternals.h>

#include "sharp_hip.h"

static void chk(int rc) {
    if (rc == SHARP_OK) return;
    if ((rc & ~(SHARP_WARN_RANGE | SHARP_WARN_NA_VOTE)) == 0) {
        if (rc & SHARP_WARN_RANGE) warning("SHARP: the model selection left the range of candidate cluster numbers (clamped)");
        if (rc & SHARP_WARN_NA_VOTE) warning("SHARP: wMetaC's single-cluster fallback met a cell with one vote value");
        return;
    }
    error("%s", sharp_last_error());
}

/* R/get_opt_hclust.R:76-83: flashmark = TRUE means flashClust(d, "ward") = the ward.D criterion; any other hmethod is an R error there */
static int method_with_flashmark(int hmethod, int flashmark) {
    if (!flashmark) return hmethod;
    if ((hmethod > 0 ? hmethod : 1) != 1) error("invalid 'y' type in 'x || y'");
    return 1;
}

SEXP R_sharp_init(SEXP dev) { chk(sharp_init(asInteger(dev))); return R_NilValue; }
SEXP R_sharp_trim(void) { chk(sharp_trim()); return R_NilValue; }

static SEXP sharp_result(SEXP pred, SEXP viE, SEXP x0, int n, int x0c, int p, int K, int path, int forview) {
    const char *names[] = {"pred", "viE", "x0", "p", "K", "path", ""};
    SEXP out = PROTECT(mkNamed(VECSXP, names));
    SET_VECTOR_ELT(out, 0, pred);
    if (forview) {
        /* viE arrives n x p row-major = a p x n R matrix: hand back its transpose, cells x p, like enE/K (R/SHARP.R:416,776-783) */
        SEXP v = PROTECT(allocMatrix(REALSXP, n, p));
        const double *s = REAL(viE);
        double *d = REAL(v);
        for (int i = 0; i < n; ++i) for (int c = 0; c < p; ++c) d[(size_t)c * n + i] = s[(size_t)i * p + c];
        SET_VECTOR_ELT(out, 1, v);
        SEXP x = PROTECT(allocMatrix(REALSXP, n, x0c));
        memcpy(REAL(x), REAL(x0), sizeof(double) * (size_t)n * (size_t)x0c);
        SET_VECTOR_ELT(out, 2, x);
        UNPROTECT(2);
    }
    SET_VECTOR_ELT(out, 3, ScalarInteger(p));
    SET_VECTOR_ELT(out, 4, ScalarInteger(K));
    SET_VECTOR_ELT(out, 5, ScalarInteger(path));
    UNPROTECT(1);
    return out;
}

/* SHARP(): X is an R numeric matrix genes x cells (column-major doubles) -- exactly the layout the ABI takes (R/SHARP.R:251-280) */
SEXP R_sharp_SHARP(SEXP X, SEXP ipar, SEXP dpar, SEXP forview_) {
    const int m = nrows(X), n = ncols(X), *ip = INTEGER(ipar), forview = asLogical(forview_);
    const double *dp = REAL(dpar);
    const int hm = method_with_flashmark(ip[4], ip[10]);
    const int pmax = ip[1] > 0 ? ip[1] : (int)ceil(log2((double)n) / 0.04);
    int capc = (ip[9] > 40 ? ip[9] : 40);
    if ((n + 4999) / 5000 > capc) capc = (n + 4999) / 5000;
    capc += 2;
    SEXP pred = PROTECT(allocVector(INTSXP, n));
    SEXP viE = PROTECT(allocVector(REALSXP, forview ? (R_xlen_t)n * pmax : 1));
    SEXP x0 = PROTECT(allocVector(REALSXP, forview ? (R_xlen_t)n * capc : 1));
    int npred = 0, p = 0, K = 0, path = 0, x0c = 0;
    chk(sharp_SHARP(REAL(X), m, n, m, ip[0], ip[1], ip[2], ip[3], hm, ip[5], ip[6], ip[7], ip[8], ip[9], dp[0], dp[1], ip[11], ip[12], dp[2],
                    INTEGER(pred), &npred, forview ? REAL(viE) : NULL, forview ? REAL(x0) : NULL, capc, &x0c, &p, &K, &path));
    SEXP out = sharp_result(pred, viE, x0, n, x0c, p, K, path, forview);
    UNPROTECT(3);
    return out;
}

/* scExp held as a Matrix::dgCMatrix: the three slots go over as they are, nothing is densified on the host */
SEXP R_sharp_SHARP_csc(SEXP Xp, SEXP Xi, SEXP Xx, SEXP dim, SEXP ipar, SEXP dpar, SEXP forview_) {
    const int m = INTEGER(dim)[0], n = INTEGER(dim)[1], *ip = INTEGER(ipar), forview = asLogical(forview_);
    const double *dp = REAL(dpar);
    const int hm = method_with_flashmark(ip[4], ip[10]);
    const int pmax = ip[1] > 0 ? ip[1] : (int)ceil(log2((double)n) / 0.04);
    int capc = (ip[9] > 40 ? ip[9] : 40);
    if ((n + 4999) / 5000 > capc) capc = (n + 4999) / 5000;
    capc += 2;
    SEXP pred = PROTECT(allocVector(INTSXP, n));
    SEXP viE = PROTECT(allocVector(REALSXP, forview ? (R_xlen_t)n * pmax : 1));
    SEXP x0 = PROTECT(allocVector(REALSXP, forview ? (R_xlen_t)n * capc : 1));
    int npred = 0, p = 0, K = 0, path = 0, x0c = 0;
    chk(sharp_SHARP_csc(INTEGER(Xp), INTEGER(Xi), REAL(Xx), m, n, ip[0], ip[1], ip[2], ip[3], hm, ip[5], ip[6], ip[7], ip[8], ip[9], dp[0],
                        dp[1], ip[11], ip[12], dp[2], INTEGER(pred), &npred, forview ? REAL(viE) : NULL, forview ? REAL(x0) : NULL, capc,
                        &x0c, &p, &K, &path));
    SEXP out = sharp_result(pred, viE, x0, n, x0c, p, K, path, forview);
    UNPROTECT(3);
    return out;
}

/* SHARP_unlimited(): blocks = list of numeric matrices sharing the gene axis (R/SHARP_unlimited.R:96-183);
 * ipar = ensize.K, N.cluster, minN.cluster, maxN.cluster */
SEXP R_sharp_unlimited(SEXP blocks, SEXP ipar, SEXP seed, SEXP viewflag_) {
    const int nb = LENGTH(blocks), *ip = INTEGER(ipar), viewflag = asLogical(viewflag_);
    if (nb < 1) error("No expression data is provided!");
    const int m = nrows(VECTOR_ELT(blocks, 0));
    const double **ptrs = (const double **)R_alloc((size_t)nb, sizeof(double *));
    long long *ncb = (long long *)R_alloc((size_t)nb, sizeof(long long));
    long long ncells = 0;
    for (int b = 0; b < nb; ++b) {
        SEXP B = VECTOR_ELT(blocks, b);
        if (!isReal(B) || nrows(B) != m) error("The input should be a LIST of partitioned scRNA-seq expression matrices!");
        ptrs[b] = REAL(B); ncb[b] = ncols(B); ncells += ncb[b];
    }
    const int p = (int)ceil(log2((double)ncells) / 0.04);
    SEXP pred = PROTECT(allocVector(INTSXP, (R_xlen_t)ncells));
    SEXP viE = PROTECT(allocVector(REALSXP, viewflag ? (R_xlen_t)ncells * p : 1));
    int npred = 0, pu = 0;
    chk(sharp_SHARP_unlimited_view(ptrs, ncb, nb, m, ip[0], ip[1], ip[2], ip[3], asReal(seed), INTEGER(pred), &npred, &pu,
                                   viewflag ? REAL(viE) : NULL));
    const char *names[] = {"pred", "viE", "p", ""};
    SEXP out = PROTECT(mkNamed(VECSXP, names));
    SET_VECTOR_ELT(out, 0, pred);
    if (viewflag) {
        SEXP v = PROTECT(allocMatrix(REALSXP, (int)ncells, pu));
        const double *s = REAL(viE);
        double *d = REAL(v);
        for (long long i = 0; i < ncells; ++i) for (int c = 0; c < pu; ++c) d[(size_t)c * (size_t)ncells + (size_t)i] = s[(size_t)i * pu + c];
        SET_VECTOR_ELT(out, 1, v);
        UNPROTECT(1);
    }
    SET_VECTOR_ELT(out, 2, ScalarInteger(pu));
    UNPROTECT(3);
    return out;
}

/* SHARP_unlimited() over the GPUs of `devices` inside this R process, the list read in place (R/SHARP_unlimited.R:125-163: the block loop
 * dealt out, block b to devices[b mod N]).  blocks = list whose elements are numeric matrices (genes x cells) or lists
 * list(p = <@p>, i = <@i>, x = <@x>, dim = <@Dim>) made from a Matrix::dgCMatrix by r/sharp_hip.R::.sharp_block -- all dense or all
 * sparse; devices = integer vector of GPU indices (length 0: the library's own choice: its current GPU, or SHARP_DEVICES).  Nothing is
 * copied on the R side and no vector length passes through an int: a 162 500 x 27 000 block (4.39e9 doubles) goes through, which
 * .C() -- whose arguments are duplicated and may not be long vectors -- cannot carry.
 * ipar = ensize.K, N.cluster, minN.cluster, maxN.cluster */
static SEXP list_elt(SEXP lst, const char *name) {
    SEXP nm = getAttrib(lst, R_NamesSymbol);
    for (R_xlen_t q = 0; q < XLENGTH(lst); ++q)
        if (nm != R_NilValue && strcmp(CHAR(STRING_ELT(nm, q)), name) == 0) return VECTOR_ELT(lst, q);
    return R_NilValue;
}
SEXP R_sharp_unlimited_multi(SEXP blocks, SEXP ipar, SEXP seed, SEXP viewflag_, SEXP devices, SEXP view_dim_) {
    /* view_dim > 0 (r/sharp_hip.R passes 50 above 1e5 cells): viE comes back as ncells x view_dim, E1 reduced per block on its GPU by one
     * more sparse projection -- what R/SHARP_unlimited.R:216-228 computes from E1 on the host (sharp_unlimited_view_dim) */
    const int nb = LENGTH(blocks), *ip = INTEGER(ipar), viewflag = asLogical(viewflag_), ndev = LENGTH(devices), view_dim = viewflag ? asInteger(view_dim_) : 0;
    if (nb < 1) error("No expression data is provided!");
    const int sparse = isNewList(VECTOR_ELT(blocks, 0));
    long long *ncb = (long long *)R_alloc((size_t)nb, sizeof(long long));
    const double **ptrs = (const double **)R_alloc((size_t)nb, sizeof(double *));      /* dense: the matrices; sparse: the @x slots */
    const int **cp = (const int **)R_alloc((size_t)nb, sizeof(int *)), **ri = (const int **)R_alloc((size_t)nb, sizeof(int *));
    long long ncells = 0;
    int m = -1;
    for (int b = 0; b < nb; ++b) {
        SEXP B = VECTOR_ELT(blocks, b);
        if (sparse != isNewList(B)) error("The input should be a LIST of partitioned scRNA-seq expression matrices!");
        if (sparse) {
            SEXP P = list_elt(B, "p"), I = list_elt(B, "i"), X = list_elt(B, "x"), D = list_elt(B, "dim");
            if (!isInteger(P) || !isInteger(I) || !isReal(X) || !isInteger(D) || LENGTH(D) != 2 || XLENGTH(P) != (R_xlen_t)INTEGER(D)[1] + 1)
                error("The input should be a LIST of partitioned scRNA-seq expression matrices!");
            if (m < 0) m = INTEGER(D)[0];
            if (INTEGER(D)[0] != m) error("The input should be a LIST of partitioned scRNA-seq expression matrices!");
            /* the three slots go to the upload threads as raw pointers: a hand-built list(p, i, x, dim) must not make them read past
             * the R vectors (a dgCMatrix satisfies all of this by its validity method) */
            {
                const int *pp = INTEGER(P), nc = INTEGER(D)[1];
                if (pp[0] != 0) error("sparse block %d: slot p does not start at 0", b + 1);
                for (int c = 0; c < nc; ++c)
                    if (pp[c + 1] < pp[c]) error("sparse block %d: slot p is not non-decreasing", b + 1);
                if (XLENGTH(I) != XLENGTH(X) || XLENGTH(I) < (R_xlen_t)pp[nc])
                    error("sparse block %d: slots i and x must have equal length, at least p[ncol + 1]", b + 1);
                const int *ii = INTEGER(I);
                for (int q = 0; q < pp[nc]; ++q)
                    if (ii[q] < 0 || ii[q] >= m) error("sparse block %d: a row index of slot i is outside 0 .. nrow - 1", b + 1);
            }
            cp[b] = INTEGER(P); ri[b] = INTEGER(I); ptrs[b] = REAL(X); ncb[b] = INTEGER(D)[1];
        } else {
            if (!isReal(B)) error("The input should be a LIST of partitioned scRNA-seq expression matrices!");
            if (m < 0) m = nrows(B);
            if (nrows(B) != m) error("The input should be a LIST of partitioned scRNA-seq expression matrices!");
            ptrs[b] = REAL(B); ncb[b] = ncols(B);
        }
        ncells += ncb[b];
    }
    const int p = (int)ceil(log2((double)ncells) / 0.04);
    if (view_dim < 0 || view_dim > 4096) error("view.dim must lie in 0 .. 4096");
    const int vcols = view_dim > 0 ? view_dim : p;
    SEXP pred = PROTECT(allocVector(INTSXP, (R_xlen_t)ncells));
    SEXP viE = PROTECT(allocVector(REALSXP, viewflag ? (R_xlen_t)ncells * vcols : 1));
    int npred = 0, pu = 0;
    if (view_dim > 0) chk(sharp_unlimited_view_dim(view_dim));      /* one-shot: the call below takes it */
    if (sparse)
        chk(sharp_SHARP_unlimited_csc_multi(cp, ri, ptrs, ncb, nb, m, ip[0], ip[1], ip[2], ip[3], asReal(seed), ndev ? INTEGER(devices) : NULL,
                                            ndev, INTEGER(pred), &npred, &pu, viewflag ? REAL(viE) : NULL));
    else if (ndev >= 1)
        chk(sharp_SHARP_unlimited_multi(ptrs, ncb, nb, m, ip[0], ip[1], ip[2], ip[3], asReal(seed), INTEGER(devices), ndev, INTEGER(pred),
                                        &npred, &pu, viewflag ? REAL(viE) : NULL));
    else
        chk(sharp_SHARP_unlimited_view(ptrs, ncb, nb, m, ip[0], ip[1], ip[2], ip[3], asReal(seed), INTEGER(pred), &npred, &pu,
                                       viewflag ? REAL(viE) : NULL));
    const char *names[] = {"pred", "viE", "p", ""};
    SEXP out = PROTECT(mkNamed(VECSXP, names));
    SET_VECTOR_ELT(out, 0, pred);
    if (viewflag) {
        SEXP v = PROTECT(allocMatrix(REALSXP, (int)ncells, vcols));
        const double *s = REAL(viE);
        double *d = REAL(v);
        for (long long i = 0; i < ncells; ++i) for (int c = 0; c < vcols; ++c) d[(size_t)c * (size_t)ncells + (size_t)i] = s[(size_t)i * vcols + c];
        SET_VECTOR_ELT(out, 1, v);
        UNPROTECT(1);
    }
    SET_VECTOR_ELT(out, 2, ScalarInteger(pu));
    UNPROTECT(3);
    return out;
}

static const R_CallMethodDef call_methods[] = {
    {"R_sharp_init", (DL_FUNC)&R_sharp_init, 1},
    {"R_sharp_trim", (DL_FUNC)&R_sharp_trim, 0},
    {"R_sharp_SHARP", (DL_FUNC)&R_sharp_SHARP, 4},
    {"R_sharp_SHARP_csc", (DL_FUNC)&R_sharp_SHARP_csc, 7},
    {"R_sharp_unlimited", (DL_FUNC)&R_sharp_unlimited, 4},
    {"R_sharp_unlimited_multi", (DL_FUNC)&R_sharp_unlimited_multi, 6},
    {NULL, NULL, 0}};

void R_init_sharp_glue(DllInfo *dll) {
    R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
    R_useDynamicSymbols(dll, TRUE);      /* the sharp_C_* symbols of libsharp_hip.so are looked up by name through .C() */
}
