# sharp_hip.R -- R side of libsharp_hip.so for the reference package (shibiaowan/SHARP): drop-in bodies for the exported
# functions on the hot path (NAMESPACE:3-28).  Source this file after the package (or paste the bodies into R/*.R):
# the functions keep the reference's names, argument lists, defaults, messages and return lists, and hand the computation
# to the MI355X library in ONE native call each, so no foreach/%dopar% worker ever touches the GPU context.
#
# Two bindings of the same C ABI (include/sharp_hip.h):
#   .C()    -- sharp_C_* entry points: needs nothing but dyn.load("libsharp_hip.so") (no code compiled against R.h);
#              R copies every argument, so a 50 000 x 20 000 matrix costs one extra 8 GB copy.
#   .Call() -- r/sharp_glue.c (R CMD SHLIB sharp_glue.c -L. -lsharp_hip): no copies; used when sharp_glue is loaded.
# NOT run in the build image (no R there): the ctypes tests call the sharp_C_* symbols with exactly these argument lists
# (tests/test_dotc_gpu.py).

sharp_hip_load <- function(libdir = ".", device = 0L) {
    dyn.load(file.path(libdir, paste0("libsharp_hip", .Platform$dynlib.ext)))
    glue <- file.path(libdir, paste0("sharp_glue", .Platform$dynlib.ext))
    if (file.exists(glue)) dyn.load(glue)
    st <- .C("sharp_C_init", as.integer(device), status = integer(1))$status
    .sharp_check(st)
    invisible(TRUE)
}

.sharp_has_glue <- function() is.loaded("R_sharp_SHARP")

.sharp_hmethods <- c(ward.D = 1L, single = 2L, complete = 3L, average = 4L, mcquitty = 5L, median = 6L, centroid = 7L, ward.D2 = 8L)
.sharp_hmethod <- function(h) {
    if (missing(h) || is.null(h)) return(1L)
    if (!h %in% names(.sharp_hmethods)) stop("invalid clustering method '", h, "'")
    .sharp_hmethods[[h]]
}
# a numeric matrix in DOUBLE storage: an integer count matrix (common with prep = FALSE) is INTSXP, and the .Call glue takes REAL()
# of what it is handed -- the .C route coerces with as.double, so both bindings accept the same inputs
.sharp_dmat <- function(x) { x <- data.matrix(x); storage.mode(x) <- "double"; x }

.sharp_int <- function(x) if (missing(x) || is.null(x)) 0L else as.integer(x)

# status -> R condition: 0 ok; 16 / 32 warning bits (reference quirks 8 and 11, DESIGN.md 9); anything else stop()
.sharp_check <- function(status) {
    if (status == 0L) return(invisible(0L))
    if (bitwAnd(status, bitwNot(48L)) == 0L) {
        if (bitwAnd(status, 16L) != 0L) warning("SHARP: the model selection left the range of candidate cluster numbers (clamped)")
        if (bitwAnd(status, 32L) != 0L) warning("SHARP: wMetaC's single-cluster fallback met a cell with one vote value")
        return(invisible(status))
    }
    msg <- .C("sharp_C_last_error", msg = paste(rep(" ", 2048), collapse = ""), len = 2048L)$msg
    stop(sub(" +$", "", msg), call. = FALSE)
}

# the 40 colour names of R/getrowColor.R:52-58
.sharp_colorL <- c("red", "purple", "blue", "yellow", "green", "orange", "brown", "gray", "black", "coral", "beige", "cyan",
    "turquoise", "pink", "khaki", "magenta", "violet", "salmon", "goldenrod", "orchid", "seagreen", "slategray", "darkred",
    "darkblue", "darkcyan", "darkgreen", "darkgray", "darkkhaki", "darkorange", "darkmagenta", "darkviolet", "darkturquoise",
    "darksalmon", "darkgoldenrod", "darkorchid", "darkseagreen", "darkslategray", "deeppink", "lightcoral", "lightcyan")

# ---- ranM / ranM2 / RPmat (R/ranM.R:11-33, R/ranM2.R:11-35, R/RPmat.R:14-47) --------------------------------------------
.sharp_projector <- function(m, p, seeds) {
    r <- .C("sharp_C_projector_create", as.integer(m), as.integer(p), length(seeds), as.double(seeds), handle = integer(1),
            status = integer(1))
    .sharp_check(r$status)
    r$handle
}
.sharp_projector_matrix <- function(h, m, p, k = 0L) {
    nn <- .C("sharp_C_projector_triplets", h, as.integer(k), integer(1), integer(1), integer(1), nnz = 0, status = integer(1))
    .sharp_check(nn$status)
    r <- .C("sharp_C_projector_triplets", h, as.integer(k), gene = integer(nn$nnz), col = integer(nn$nnz), sign = integer(nn$nnz),
            nnz = as.double(nn$nnz), status = integer(1))
    .sharp_check(r$status)
    Matrix::sparseMatrix(i = r$gene + 1L, j = r$col + 1L, x = r$sign * sqrt(sqrt(m)), dims = c(m, p))
}
ranM2 <- function(m, p, seedn) {
    if (!is.numeric(seedn)) stop("The seed should be a numeric!")
    h <- .sharp_projector(m, p, seedn)
    on.exit(.C("sharp_C_projector_destroy", h, integer(1)))
    .sharp_projector_matrix(h, m, p)
}
ranM <- function(scdata, p, seedn) ranM2(nrow(scdata), p, seedn)
RPmat <- function(scdata, p, seedn) {
    m <- nrow(scdata); n <- ncol(scdata)
    h <- .sharp_projector(m, p, seedn)
    on.exit(.C("sharp_C_projector_destroy", h, integer(1)))
    r <- .C("sharp_C_project", h, as.double(data.matrix(scdata)), m, n, 0L, E = double(n * p), status = integer(1))
    .sharp_check(r$status)
    list(R = .sharp_projector_matrix(h, m, p), projmat = matrix(r$E, nrow = p, ncol = n))    # p x n, like 1/sqrt(p) * t(x) %*% scdata
}

# ---- get_opt_hclust (R/get_opt_hclust.R:33-244) --------------------------------------------------------------------------
get_opt_hclust <- function(mat, hmethod, N.cluster, minN.cluster, maxN.cluster, sil.thre, height.Ntimes, flashmark) {
    if (missing(hmethod) || is.null(hmethod)) hmethod <- "ward.D"
    if (missing(minN.cluster) || is.null(minN.cluster)) minN.cluster <- 2
    if (missing(maxN.cluster) || is.null(maxN.cluster)) maxN.cluster <- 40
    if (missing(sil.thre) || is.null(sil.thre)) sil.thre <- 0.35
    if (missing(height.Ntimes) || is.null(height.Ntimes)) height.Ntimes <- 2
    if (missing(flashmark) || is.null(flashmark)) flashmark <- FALSE
    if (missing(N.cluster)) N.cluster <- NULL
    if (is.numeric(N.cluster)) {
        if (N.cluster %% 1 != 0) stop("The given N.cluster is not an integer!")
        if (N.cluster < 2) stop("The given N.cluster is less than 2, which is not suitable for clustering!")
    } else if (!is.null(N.cluster)) stop("The given N.cluster is not a numeric!")
    n <- nrow(mat); p <- ncol(mat)
    nk <- if (is.numeric(N.cluster)) 1L else max(1L, min(maxN.cluster, n - 1) - minN.cluster + 1)
    r <- .C("sharp_C_get_opt_hclust", as.double(t(mat)), n, p, .sharp_hmethod(hmethod), .sharp_int(N.cluster), as.integer(minN.cluster),
            as.integer(maxN.cluster), as.double(sil.thre), as.double(height.Ntimes), as.integer(flashmark), f = integer(n),
            v = integer(n * nk), msil = double(nk), CHind = double(nk), maxsil = double(1), height = double(max(n - 1, 1)),
            optN = integer(1), nk = integer(1), branch = integer(1), 7L, status = integer(1))
    .sharp_check(r$status)
    list(f = r$f, v = matrix(r$v[seq_len(n * r$nk)], nrow = n), maxsil = r$maxsil, msil = r$msil[seq_len(r$nk)],
         CHind = r$CHind[seq_len(r$nk)], height = r$height[seq_len(n - 1)], optN.cluster = r$optN)
}

# ---- getrowColor (R/getrowColor.R:17-121) ---------------------------------------------------------------------------------
getrowColor <- function(Emat, hmethod, indN.cluster, minN.cluster, maxN.cluster, sil.thre, height.Ntimes, flashmark) {
    if (missing(height.Ntimes) || is.null(height.Ntimes) || height.Ntimes <= 0) height.Ntimes <- 1
    if (missing(flashmark)) flashmark <- FALSE
    if (missing(indN.cluster)) indN.cluster <- NULL
    n <- nrow(Emat); p <- ncol(Emat)
    r <- .C("sharp_C_getrowColor", as.double(t(Emat)), n, p, .sharp_hmethod(hmethod), .sharp_int(indN.cluster), as.integer(minN.cluster),
            as.integer(maxN.cluster), as.double(sil.thre), as.double(height.Ntimes), as.integer(flashmark), rowColor = integer(n),
            maxsil = double(1), status = integer(1))
    .sharp_check(r$status)
    list(rowColor = .sharp_colorL[r$rowColor], maxsil = r$maxsil, mat = Emat)
}

# ---- wMetaC (R/wMetaC.R:15-226) ----------------------------------------------------------------------------------------------
wMetaC <- function(nC, hmethod, enN.cluster, minN.cluster, maxN.cluster, sil.thre, height.Ntimes) {
    if (missing(sil.thre) || is.null(sil.thre)) sil.thre <- 0                 # :94-97
    if (missing(height.Ntimes) || is.null(height.Ntimes)) height.Ntimes <- 2
    if (missing(enN.cluster)) enN.cluster <- NULL
    if (missing(minN.cluster) || is.null(minN.cluster)) minN.cluster <- 2
    if (missing(maxN.cluster) || is.null(maxN.cluster)) maxN.cluster <- 40
    N <- nrow(nC); C <- ncol(nC)
    nCi <- apply(nC, 2, function(x) match(x, unique(x)))                       # labels only ever compare for equality inside a column
    r <- .C("sharp_C_wMetaC", as.integer(nCi), N, C, .sharp_hmethod(hmethod), .sharp_int(enN.cluster), as.integer(minN.cluster),
            as.integer(maxN.cluster), as.double(sil.thre), as.double(height.Ntimes), finalC = integer(N),
            x0 = double(N * (maxN.cluster + 2)), ncl = integer(1), 1L, status = integer(1))
    .sharp_check(r$status)
    list(finalC = as.character(r$finalC), x0 = matrix(r$x0[seq_len(N * r$ncl)], nrow = N))
}

# ---- sMetaC (R/sMetaC.R:17-210) ----------------------------------------------------------------------------------------------
sMetaC <- function(rerowColor, sE1, folds, hmethod, finalN.cluster, minN.cluster, maxN.cluster, sil.thre, height.Ntimes) {
    if (missing(finalN.cluster)) finalN.cluster <- NULL
    n <- length(rerowColor); p <- ncol(sE1)
    r <- .C("sharp_C_sMetaC", match(rerowColor, unique(rerowColor)), as.double(t(sE1)), as.double(n), p, .sharp_hmethod(hmethod),
            .sharp_int(finalN.cluster), as.integer(minN.cluster), as.integer(maxN.cluster), as.double(sil.thre), as.double(height.Ntimes),
            finalColor = integer(n), tf = integer(n), nC = integer(1), status = integer(1))
    .sharp_check(r$status)
    list(finalColor = as.character(r$finalColor), tf = r$tf[seq_len(r$nC)])
}

# ---- the body of SHARP() between its argument handling and its result list (replaces R/SHARP.R:251-280) ---------------------
# scExp: the prepared matrix (after :48-117); the arguments are those SHARP() holds at :251; returns what SHARP_small / SHARP_large
# return (pred_clusters, unique_pred_clusters, distr_pred_clusters, N.pred_cluster, x0, viE, allrpinfo for the small path).
.sharp_run <- function(scExp, ensize.K, reduced.ndim, base.ncells, partition.ncells, hmethod, N.cluster, enpN.cluster, indN.cluster,
                       minN.cluster, maxN.cluster, sil.thre, height.Ntimes, flashmark, flag, forview, rM, rN.seed) {
    m <- nrow(scExp); n <- ncol(scExp)
    p <- if (is.null(reduced.ndim) || reduced.ndim <= 0) ceiling(log2(n)/(0.2^2)) else reduced.ndim
    capc <- max(.sharp_int(maxN.cluster), 40L, ceiling(n/5000)) + 2L
    proj <- if (is.numeric(rM)) as.integer(rM) else 0L                       # rM: a projector handle from .sharp_projector()
    ipar <- c(.sharp_int(ensize.K), .sharp_int(reduced.ndim), .sharp_int(base.ncells), .sharp_int(partition.ncells),
              .sharp_hmethod(hmethod), .sharp_int(N.cluster), .sharp_int(enpN.cluster), .sharp_int(indN.cluster),
              .sharp_int(minN.cluster), .sharp_int(maxN.cluster), as.integer(flashmark), as.integer(flag), proj)
    dpar <- c(if (is.null(sil.thre)) -1 else sil.thre, if (is.null(height.Ntimes)) 0 else height.Ntimes, rN.seed)
    sparse <- methods::is(scExp, "dgCMatrix")
    if (.sharp_has_glue()) {
        r <- if (sparse) .Call("R_sharp_SHARP_csc", scExp@p, scExp@i, scExp@x, dim(scExp), ipar, dpar, as.logical(forview))
             else .Call("R_sharp_SHARP", .sharp_dmat(scExp), ipar, dpar, as.logical(forview))
    } else {
        want <- if (forview) 3L else 0L
        args <- list(ipar[1], ipar[2], ipar[3], ipar[4], ipar[5], ipar[6], ipar[7], ipar[8], ipar[9], ipar[10], dpar[1], dpar[2],
                     ipar[11], ipar[12], ipar[13], dpar[3], pred = integer(n), viE = double(if (forview) n * p else 1),
                     x0 = double(if (forview) n * capc else 1), as.integer(capc), info = integer(5), want, status = integer(1))
        r <- if (sparse) do.call(.C, c(list("sharp_C_SHARP_csc", scExp@p, scExp@i, as.double(scExp@x), m, as.double(n)), args))
             else do.call(.C, c(list("sharp_C_SHARP", as.double(data.matrix(scExp)), m, as.double(n)), args))
        .sharp_check(r$status)
        r <- list(pred = r$pred, viE = if (forview) t(matrix(r$viE[seq_len(n * r$info[3])], nrow = r$info[3])) else NULL,
                  x0 = if (forview) matrix(r$x0[seq_len(n * r$info[2])], nrow = n) else NULL, p = r$info[3], K = r$info[4], path = r$info[5])
    }
    y <- r$pred
    tn <- table(y)
    en <- list(pred_clusters = y, unique_pred_clusters = sort(unique(y)), distr_pred_clusters = tn[order(as.numeric(names(tn)))],
               N.pred_cluster = length(unique(y)))
    if (forview) {
        if (r$path == 0L) {                                                   # SHARP_small only (R/SHARP.R:446)
            d <- .C("sharp_C_last_rpinfo", dims = integer(3), integer(1), double(1), 0L, status = integer(1))$dims
            q <- .C("sharp_C_last_rpinfo", dims = integer(3), enrp = integer(d[1] * d[2]), indE = double(d[1] * d[2] * d[3]), 3L,
                    status = integer(1))
            .sharp_check(q$status)
            enrp <- matrix(q$enrp, nrow = d[1]); indE <- matrix(q$indE, nrow = d[2] * d[3])   # (K p) x n
            en$allrpinfo <- lapply(seq_len(d[2]), function(k) {
                rc <- .sharp_colorL[enrp[, k]]
                list(tag = paste("_RP", d[3], "_", k, sep = ""), rowColor = rc, N.cluster = length(unique(rc)),
                     indE = t(indE[(k - 1) * d[3] + seq_len(d[3]), , drop = FALSE]))
            })
        }
        en$x0 <- r$x0
        en$viE <- r$viE
    }
    en$.reduced.dim <- r$p; en$.ensize.K <- r$K
    en
}
# In R/SHARP.R the maintainer replaces :251-280 by
#     enresults = .sharp_run(scExp, ensize.K, reduced.ndim, base.ncells, partition.ncells, hmethod, N.cluster, enpN.cluster,
#                            indN.cluster, minN.cluster, maxN.cluster, sil.thre, height.Ntimes, flashmark, flag, forview, rM, rN.seed)
# (with the `missing()` arguments passed as NULL) and keeps :48-249 (checks, prep, CPM, defaults, testlog) and :282-317 (N.cells,
# N.genes, reduced.dim, ensize.K, time, paras).  SHARP_small / SHARP_large are the same call with base.ncells = ncells + 1 / 1.

# ---- SHARP_unlimited (R/SHARP_unlimited.R:29-242): replaces :96-183 -----------------------------------------------------------
# devices: integer vector of GPU indices (default: getOption("sharp.devices"), e.g. options(sharp.devices = 0:7)); the serial block loop
# of :125-163 is dealt out, block b to devices[b mod N], inside this R process (sharp_SHARP_unlimited_multi: per GPU one host thread that
# clusters and one that uploads the next block meanwhile; nothing crosses between GPUs but the per-block centroid tables).
# With the .Call glue loaded (r/sharp_glue.c) the list -- numeric matrices or Matrix::dgCMatrix blocks -- is read in place by
# R_sharp_unlimited_multi, whatever the number of devices.  The .C() fallback duplicates its arguments, needs the blocks unlist()-ed into
# ONE vector and does not take long vectors: it carries at most 2^31 - 1 values per call (a list of 1.3 M cells x 27 000 genes, 3.5e10
# values, cannot pass -- one cfg4 block, 4.39e9, already cannot), i.e. it is for small inputs and for installations without a compiler.
.sharp_block <- function(b) {                          # a block as the glue takes it: a double matrix, or the slots of a dgCMatrix
    if (inherits(b, "dgCMatrix")) list(p = b@p, i = b@i, x = b@x, dim = b@Dim)
    else if (inherits(b, "sparseMatrix")) { b <- methods::as(b, "CsparseMatrix"); list(p = b@p, i = b@i, x = as.double(b@x), dim = b@Dim) }
    else .sharp_dmat(b)
}
.sharp_unlimited_run <- function(scExp, ensize.K, N.cluster, minN.cluster, maxN.cluster, rN.seed, viewflag,
                                 devices = getOption("sharp.devices"), view.reduce = TRUE) {
    nb <- length(scExp); m <- nrow(scExp[[1]])
    ncb <- vapply(scExp, ncol, 1)
    ncells <- sum(ncb)
    p <- ceiling(log2(ncells)/(0.2^2))
    # R/SHARP_unlimited.R:216-228: above 1e5 cells enresults$viE is 1/sqrt(50) * E1 %*% ranM2(p, 50, seed), never E1.  With view.reduce the
    # library takes that product per block on the GPU (sharp_unlimited_view_dim) and r$viE comes back ncells x 50: the caller then sets
    # enresults$viE = r$viE instead of running lines 216-228 on a ncells x p matrix it no longer has.
    kdim <- if (isTRUE(view.reduce) && viewflag && ncells > 1e5) 50L else 0L
    vcols <- if (kdim > 0L) kdim else p
    arm <- function() if (kdim > 0L) .sharp_check(.C("sharp_C_unlimited_view_dim", kdim, status = integer(1))$status)
    ipar <- c(.sharp_int(ensize.K), .sharp_int(N.cluster), .sharp_int(minN.cluster), .sharp_int(maxN.cluster))
    sparse <- all(vapply(scExp, function(b) inherits(b, "sparseMatrix"), TRUE))
    if (.sharp_has_glue()) {
        blocks <- if (sparse) lapply(scExp, .sharp_block) else lapply(scExp, .sharp_dmat)
        r <- .Call("R_sharp_unlimited_multi", blocks, ipar, as.double(rN.seed), as.logical(viewflag), as.integer(devices), kdim)
    } else if (sparse) {
        cs <- lapply(scExp, .sharp_block)
        if (sum(as.numeric(vapply(cs, function(b) length(b$x), 1))) >= 2^31) stop("SHARP_unlimited: this input needs the .Call glue (r/sharp_glue.c): .C() carries at most 2^31 - 1 values")
        arm()
        r <- .C("sharp_C_SHARP_unlimited_csc", unlist(lapply(cs, `[[`, "p")), unlist(lapply(cs, `[[`, "i")), unlist(lapply(cs, `[[`, "x")), nb,
                as.double(ncb), m, ipar[1], ipar[2], ipar[3], ipar[4], as.double(rN.seed), as.integer(c(devices, 0L)), length(devices),
                pred = integer(ncells), viE = double(if (viewflag) ncells * vcols else 1), info = integer(2), as.integer(viewflag), status = integer(1))
        .sharp_check(r$status)
        r <- list(pred = r$pred, viE = if (viewflag) t(matrix(r$viE, nrow = vcols)) else NULL, p = r$info[2])
    } else {
        if (as.numeric(m) * ncells >= 2^31) stop("SHARP_unlimited: this input needs the .Call glue (r/sharp_glue.c): .C() carries at most 2^31 - 1 values")
        xcat <- unlist(lapply(scExp, function(b) as.double(data.matrix(b))))
        arm()
        if (length(devices) >= 2)
            r <- .C("sharp_C_SHARP_unlimited_multi", xcat, nb, as.double(ncb), m, ipar[1], ipar[2], ipar[3], ipar[4], as.double(rN.seed),
                    as.integer(devices), length(devices), pred = integer(ncells), viE = double(if (viewflag) ncells * vcols else 1),
                    info = integer(2), as.integer(viewflag), status = integer(1))
        else
            r <- .C("sharp_C_SHARP_unlimited", xcat, nb, as.double(ncb), m, ipar[1], ipar[2], ipar[3], ipar[4], as.double(rN.seed),
                    pred = integer(ncells), viE = double(if (viewflag) ncells * vcols else 1), info = integer(2), as.integer(viewflag),
                    status = integer(1))
        .sharp_check(r$status)
        r <- list(pred = r$pred, viE = if (viewflag) t(matrix(r$viE, nrow = vcols)) else NULL, p = r$info[2])
    }
    r$view.dim <- kdim                                                        # > 0: r$viE IS enresults$viE (ncells x 50); 0: r$viE is E1
    r                                                                         # finalrowColor = r$pred (ids by decreasing size)
}

# ---- the decision log (include/sharp_hip.h: sharp_decision_log / sharp_last_decisions; SURVEY.md 7, App. D.2) -------------------------
# sharp_decision_log(TRUE); res <- SHARP(...); d <- sharp_last_decisions(); sharp_decision_log(FALSE)
# One row per get_opt_hclust call of the run: which rule of R/get_opt_hclust.R:162-229 chose the number of clusters, how many levels tied
# exactly at the maximum (the reference takes the middle one), the maximum and the runner-up.  To attribute a label difference between this
# library and the reference to ONE decision, print the same quantities from the reference's own get_opt_hclust and compare row by row.
sharp_decision_log <- function(enable = TRUE) invisible(.sharp_check(.C("sharp_C_decision_log", as.integer(enable), status = integer(1))$status))
sharp_last_decisions <- function(cap = 65536L) {
    r <- .C("sharp_C_last_decisions", rows = double(14L * cap), as.integer(cap), n = integer(1), status = integer(1))
    .sharp_check(r$status)
    d <- as.data.frame(t(matrix(r$rows[seq_len(14L * min(r$n, cap))], nrow = 14L)))
    names(d) <- c("level", "block", "k", "fold", "n", "branch", "chosen.k", "ties", "best", "runner.up", "sil.minus.thre", "height.ratio",
                  "smetac.override.k", "levels")
    d$level <- c("direct", "base", "wMetaC", "sMetaC", "merge")[d$level + 2L]
    d$branch <- c("silhouette", "CH", "height", "N.cluster")[d$branch + 1L]
    d
}
