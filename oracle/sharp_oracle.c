/*
 * sharp_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, fp64 (long double where R itself uses LDOUBLE) restatement of the
 * SHARP hot path of the reference R package (shibiaowan/SHARP, /root/reference):
 *   ranM / RPmat -> get_opt_hclust / getrowColor -> wMetaC -> sMetaC ->
 *   SHARP_small / SHARP_large / SHARP_unlimited / SHARP_unlimited2 (SHARP_fpart).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into this file.  The product (sharp_amd/, libsharp_hip.so) never does.
 *
 * PARITY STATUS: "parity unpinned" against real R output.  The reference has no
 * tests / golden vectors for this path and R is not installed in
 * the build container.  What IS pinned (tests/test_oracle_*.py):
 *   - the R RNG restatement against the well-known R outputs
 *     set.seed(1);runif(3), set.seed(42);sample(10), ...;
 *   - hclust(ward.D) against scipy.cluster.hierarchy (App. B);
 *   - silhouette against sklearn.metrics.silhouette_samples;
 *   - ARI against sklearn.metrics.adjusted_rand_score.
 * Third-party arithmetic restated here (none of it is vendored in the reference):
 *   R core (unpinned, >= 3.6 assumed): set.seed, sample, scale, cor, hclust.f,
 *   cutree, median, table/sort/unique, round(x, 1) (nmath/fround.c of R >= 4.0);  cluster::silhouette (sildist.c);
 *   clues 0.6.2.2 get_CH / adjustedRand (formula per SURVEY.md App. A.6, unverified);
 *   clusterCrit::intCriteria("Calinski_Harabasz");  Matrix sparse %*% dense
 *   (ascending-row accumulation).
 *
 * Every function cites the reference file:line it follows.
 * Index conventions: cells/observations 0-based in C arrays, cluster ids 1-based
 * like R.  Matrices are documented per function.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef long double LD; /* R's LDOUBLE on x86-64 */

#define OR_OK 0
#define OR_ERR_ARG 1
#define OR_ERR_RANGE 2 /* R would raise "subscript out of bounds" or similar */
#define OR_WARN_NA_VOTE 4

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory (%zu)\n", n); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

/* ------------------------------------------------------------------------- */
/* R's default RNG: Mersenne-Twister + set.seed() scrambling  [R-internal,    */
/* src/main/RNG.c; SURVEY.md App. A.1].  Used by R/ranM.R:22, R/SHARP.R:497.  */
/* ------------------------------------------------------------------------- */
#define MT_N 624
#define MT_M 397
typedef struct { uint32_t mt[MT_N]; int mti; } rrng;

static void rrng_set_seed(rrng *g, uint32_t seed) {
    /* RNG_Init: 50 scrambling steps, then 625 LCG outputs fill i_seed[];
       FixupSeeds overwrites i_seed[0] (= mti) with 624. */
    for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
    for (int j = 0; j < MT_N + 1; j++) {
        seed = 69069u * seed + 1u;
        if (j > 0) g->mt[j - 1] = seed;
    }
    g->mti = MT_N;
}
static double rrng_unif(rrng *g) {
    static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
    uint32_t y;
    if (g->mti >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (g->mt[kk] & 0x80000000u) | (g->mt[kk + 1] & 0x7fffffffu);
            g->mt[kk] = g->mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (g->mt[kk] & 0x80000000u) | (g->mt[kk + 1] & 0x7fffffffu);
            g->mt[kk] = g->mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (g->mt[MT_N - 1] & 0x80000000u) | (g->mt[0] & 0x7fffffffu);
        g->mt[MT_N - 1] = g->mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
        g->mti = 0;
    }
    y = g->mt[g->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    double x = (double)y * 2.3283064365386963e-10;
    /* fixup(): keep strictly inside (0,1) */
    const double i2_32m1 = 2.328306437080797e-10;
    if (x <= 0.0) return 0.5 * i2_32m1;
    if ((1.0 - x) <= 0.0) return 1.0 - 0.5 * i2_32m1;
    return x;
}
/* R >= 3.6 sample.kind="Rejection": R_unif_index() [SURVEY.md App. A.2] */
static double rrng_rbits(rrng *g, int bits) {
    int64_t v = 0;
    for (int n = 0; n <= bits; n += 16) {
        int v1 = (int)floor(rrng_unif(g) * 65536);
        v = 65536 * v + v1;
    }
    if (bits < 64) v &= (((int64_t)1 << bits) - 1);
    return (double)v;
}
static double rrng_unif_index(rrng *g, double dn) {
    if (dn <= 0) return 0.0;
    int bits = (int)ceil(log2(dn));
    double dv;
    do { dv = rrng_rbits(g, bits); } while (dn <= dv);
    return dv;
}
/* sample(n): permutation of 1..n  (do_sample, replace = FALSE, no prob) */
static void rrng_sample_perm(rrng *g, int n, int *out /* 1-based values */) {
    int *x = (int *)xmalloc(sizeof(int) * (size_t)n);
    int nleft = n;
    for (int i = 0; i < n; i++) x[i] = i;
    for (int i = 0; i < n; i++) {
        int j = (int)rrng_unif_index(g, (double)nleft);
        out[i] = x[j] + 1;
        x[j] = x[--nleft];
    }
    free(x);
}

void oracle_runif(uint32_t seed, int n, double *out) {
    rrng g; rrng_set_seed(&g, seed);
    for (int i = 0; i < n; i++) out[i] = rrng_unif(&g);
}
void oracle_sample_perm(uint32_t seed, int n, int *out) {
    rrng g; rrng_set_seed(&g, seed);
    rrng_sample_perm(&g, n, out);
}

/* ------------------------------------------------------------------------- */
/* ranM / ranM2 / projector half of RPmat   (R/ranM.R:11-33, R/ranM2.R:11-35, */
/* R/RPmat.R:14-31).                                                          */
/*   x0 = sample(c(sqrt(s),0,-sqrt(s)), m*p, TRUE, prob=c(1/(2s),1-1/s,1/(2s)))*/
/*   Matrix(x0, nrow=m, byrow=TRUE): element i = r*p + c  ->  R[r,c].         */
/* sample() with prob and 3 candidates = ProbSampleReplace (inversion, one    */
/* unif_rand() per element) after FixupProb + revsort; revsort of (q,P,q)     */
/* yields the order (0 ; -sqrt(s) ; +sqrt(s))  [SURVEY.md App. A.3].          */
/* Output: tern[r*p+c] in {+1,0,-1}; the magnitude is sqrt(s), s = sqrt(m).   */
/* seedn integer -> set.seed(seedn); non-integer (0.5) -> unseeded in R: here */
/* the caller-supplied fallback seed is used instead (not reproducible in R). */
/* ------------------------------------------------------------------------- */
void oracle_ranM(int m, int p, double seedn, int8_t *tern) {
    double s = sqrt((double)m);
    double pr[3] = {1.0 / (2.0 * s), 1.0 - 1.0 / s, 1.0 / (2.0 * s)};
    double sum = 0.0;
    for (int i = 0; i < 3; i++) if (pr[i] > 0.0) sum += pr[i];
    for (int i = 0; i < 3; i++) pr[i] /= sum;
    /* after revsort: a = [P, q(elt 3), q(elt 1)], cumulative */
    double c0 = pr[1];
    double c1 = pr[1] + pr[2];
    rrng g;
    uint32_t seed = (fmod(seedn, 1.0) == 0.0) ? (uint32_t)(int32_t)seedn : 20261003u;
    rrng_set_seed(&g, seed);
    size_t tot = (size_t)m * (size_t)p;
    for (size_t i = 0; i < tot; i++) {
        double u = rrng_unif(&g);
        int8_t v;
        if (u <= c0) v = 0;            /* perm[0] = 2 -> value 0        */
        else if (u <= c1) v = -1;      /* perm[1] = 3 -> -sqrt(s)       */
        else v = 1;                    /* perm[2] = 1 -> +sqrt(s)       */
        tern[i] = v;
    }
}

/* ------------------------------------------------------------------------- */
/* RP matmul (R/RPmat.R:32, R/SHARP.R:343-345,363,569-571,579-585):          */
/*   E1 = t( (1/sqrt(p)) * (t(R) %*% L) ),  L = log2(X+1) if logflag.         */
/* X: genes x cells, column-major (cell contiguous), like R.                  */
/* E: cells x p, row-major (= R's p x n projmat, column-major).               */
/* Sparse %*% dense accumulates in ascending gene order, term = fl(v*x).      */
/* ------------------------------------------------------------------------- */
void oracle_project(const double *X, int m, int n, const int8_t *tern, int p,
                    int logflag, double *E) {
    double s = sqrt((double)m);
    double val = sqrt(s);
    double scale = 1.0 / sqrt((double)p);
    /* row lists of the sparse projector (what Matrix(..., sparse=TRUE) stores) */
    size_t *rp = (size_t *)xcalloc((size_t)m + 1, sizeof(size_t));
    for (int g = 0; g < m; g++) {
        size_t c = 0; const int8_t *row = tern + (size_t)g * (size_t)p;
        for (int q = 0; q < p; q++) c += (row[q] != 0);
        rp[g + 1] = rp[g] + c;
    }
    int *ci = (int *)xmalloc(sizeof(int) * rp[m]);
    { size_t w = 0;
      for (int g = 0; g < m; g++) { const int8_t *row = tern + (size_t)g * (size_t)p;
        for (int q = 0; q < p; q++) if (row[q]) ci[w++] = row[q] > 0 ? q : -q - 1; } }
    for (int cell = 0; cell < n; cell++) {
        const double *x = X + (size_t)cell * (size_t)m;
        double *e = E + (size_t)cell * (size_t)p;
        for (int c = 0; c < p; c++) e[c] = 0.0;
        for (int g = 0; g < m; g++) {          /* ascending gene order per output */
            double lv = logflag == 2 ? log10(x[g] + 1.0) : (logflag ? log2(x[g] + 1.0) : x[g]);   /* 2: R/SHARP_unlimited2.R:391 */
            if (lv == 0.0) continue;           /* adding fl(v*0) = +-0 changes nothing */
            double tp = val * lv, tm = -val * lv;
            for (size_t q = rp[g]; q < rp[g + 1]; q++) {
                int c = ci[q];
                if (c >= 0) e[c] += tp; else e[-c - 1] += tm;
            }
        }
        for (int c = 0; c < p; c++) e[c] = scale * e[c];
    }
    free(rp); free(ci);
}

/* ------------------------------------------------------------------------- */
/* stats::hclust Fortran kernel (hclust.f) [R-internal; SURVEY.md App. A.4].  */
/* Used by R/get_opt_hclust.R:77 (and :82, flashClust "ward" = same criterion)*/
/* method: 1 ward.D, 2 single, 3 complete, 4 average, 5 mcquitty, 6 median,   */
/*         7 centroid, 8 ward.D2.                                             */
/* diss: condensed, IOFFST(n,i,j) = j + (i-1)n - i(i+1)/2, 1-based, i<j.      */
/* Outputs ia/ib (1-based representatives, i2<j2) and crit, n-1 entries each. */
/* diss is modified in place.                                                 */
/* ------------------------------------------------------------------------- */
static inline size_t ioffst(size_t n, size_t i, size_t j) {
    return j + (i - 1) * n - (i * (i + 1)) / 2; /* 1-based position */
}
void oracle_hclust(int n_, int method, double *diss_, int *ia, int *ib, double *crit) {
    size_t n = (size_t)n_;
    double *diss = diss_ - 1; /* 1-based */
    const double INF = 1.0e300;
    int *nn = (int *)xcalloc(n + 1, sizeof(int));
    double *disnn = (double *)xcalloc(n + 1, sizeof(double));
    double *membr = (double *)xmalloc(sizeof(double) * (n + 1));
    char *flag = (char *)xmalloc(n + 1);
    size_t len = n * (n - 1) / 2;
    int isWard = (method == 1 || method == 8);
    for (size_t i = 1; i <= n; i++) { flag[i] = 1; membr[i] = 1.0; }
    if (method == 8) for (size_t i = 1; i <= len; i++) diss[i] = diss[i] * diss[i];
    size_t im = 0, jm = 0, jj = 0;
    for (size_t i = 1; i <= n - 1; i++) {
        double dmin = INF;
        for (size_t j = i + 1; j <= n; j++) {
            size_t ind = ioffst(n, i, j);
            if (dmin > diss[ind]) { dmin = diss[ind]; jm = j; }
        }
        nn[i] = (int)jm; disnn[i] = dmin;
    }
    size_t ncl = n;
    while (ncl > 1) {
        double dmin = INF;
        for (size_t i = 1; i <= n - 1; i++) {
            if (flag[i] && disnn[i] < dmin) { dmin = disnn[i]; im = i; jm = (size_t)nn[i]; }
        }
        ncl--;
        size_t i2 = im < jm ? im : jm, j2 = im < jm ? jm : im;
        ia[n - ncl - 1] = (int)i2; ib[n - ncl - 1] = (int)j2;
        crit[n - ncl - 1] = (method == 8) ? sqrt(dmin) : dmin;
        flag[j2] = 0;
        dmin = INF;
        double d12 = diss[ioffst(n, i2, j2)];
        for (size_t k = 1; k <= n; k++) {
            if (flag[k] && k != i2) {
                size_t ind1 = (i2 < k) ? ioffst(n, i2, k) : ioffst(n, k, i2);
                size_t ind2 = (j2 < k) ? ioffst(n, j2, k) : ioffst(n, k, j2);
                double d1 = diss[ind1], d2 = diss[ind2], dn;
                if (isWard) {
                    dn = (membr[i2] + membr[k]) * d1 + (membr[j2] + membr[k]) * d2 - membr[k] * d12;
                    dn = dn / (membr[i2] + membr[j2] + membr[k]);
                } else if (method == 2) dn = d1 < d2 ? d1 : d2;
                else if (method == 3) dn = d1 > d2 ? d1 : d2;
                else if (method == 4) dn = (membr[i2] * d1 + membr[j2] * d2) / (membr[i2] + membr[j2]);
                else if (method == 5) dn = (d1 + d2) / 2;
                else if (method == 6) dn = ((d1 + d2) - d12 / 2) / 2;
                else dn = (membr[i2] * d1 + membr[j2] * d2 - membr[i2] * membr[j2] * d12 / (membr[i2] + membr[j2])) / (membr[i2] + membr[j2]);
                diss[ind1] = dn;
                if (i2 < k) {
                    if (dn < dmin) { dmin = dn; jj = k; }
                } else { /* i2 > k: keep NN list right for non-monotone methods */
                    if (dn < disnn[k]) { disnn[k] = dn; nn[k] = (int)i2; }
                }
            }
        }
        membr[i2] = membr[i2] + membr[j2];
        disnn[i2] = dmin; nn[i2] = (int)jj;
        for (size_t i = 1; i <= n - 1; i++) {
            if (flag[i] && ((size_t)nn[i] == i2 || (size_t)nn[i] == j2)) {
                dmin = INF;
                for (size_t j = i + 1; j <= n; j++) {
                    if (flag[j]) {
                        size_t ind = ioffst(n, i, j);
                        if (diss[ind] < dmin) { dmin = diss[ind]; jj = j; }
                    }
                }
                nn[i] = (int)jj; disnn[i] = dmin;
            }
        }
    }
    free(nn); free(disnn); free(membr); free(flag);
}

/* cutree(h, k): apply the first n-k merges; ids by first appearance in         */
/* observation order (R_cutree).  lab: n ints, 1-based ids.                     */
static void cutree_k(int n, const int *ia, const int *ib, int k, int *lab) {
    int *rep = (int *)xmalloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 1; i <= n; i++) rep[i] = i;
    int nm = n - k; if (nm < 0) nm = 0; if (nm > n - 1) nm = n - 1;
    /* j2 joins i2; i2 stays the representative (i2 < j2 always) */
    for (int s = 0; s < nm; s++) rep[ib[s]] = ia[s];
    int *id = (int *)xcalloc((size_t)n + 1, sizeof(int));
    int ncl = 0;
    for (int i = 1; i <= n; i++) {
        int r = i; while (rep[r] != r) r = rep[r];
        if (!id[r]) id[r] = ++ncl;
        lab[i - 1] = id[r];
    }
    free(rep); free(id);
}
/* cutree(h, h=hc): k = n + 1 - which.max(c(height, Inf) > hc) */
static int cutree_h_k(int n, const double *height, double hc) {
    int idx = n; /* position of Inf, 1-based = n */
    for (int i = 0; i < n - 1; i++) if (height[i] > hc) { idx = i + 1; break; }
    return n + 1 - idx;
}

/* cluster::silhouette.default -> sildist() [SURVEY.md App. A.5]; d condensed.  */
/* Returns median(sil[,3]) like R/get_opt_hclust.R:134-137.                     */
static int cmp_dbl(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}
static double r_median(double *x, int n) { /* stats::median.default */
    qsort(x, (size_t)n, sizeof(double), cmp_dbl);
    if (n % 2 == 1) return x[n / 2];
    /* mean(c(a,b)): long double sum/2 then one refinement pass (summary.c) */
    LD a = x[n / 2 - 1], b = x[n / 2];
    LD s = (a + b) / 2;
    LD t = (a - s) + (b - s);
    s += t / 2;
    return (double)s;
}
static void silhouette_widths(int n, int k, const int *cl, const double *dcond, double *si) {
    double *diC = (double *)xcalloc((size_t)n * (size_t)k, sizeof(double));
    int *counts = (int *)xcalloc((size_t)k, sizeof(int));
    size_t l = 0;
    for (int i = 0; i < n; i++) {
        int ci = cl[i] - 1;
        counts[ci]++;
        for (int j = i + 1; j < n; j++, l++) {
            int cj = cl[j] - 1;
            diC[(size_t)k * i + cj] += dcond[l];
            diC[(size_t)k * j + ci] += dcond[l];
        }
    }
    for (int i = 0; i < n; i++) {
        size_t ki = (size_t)k * i; int ci = cl[i] - 1; int computeSi = 1;
        for (int j = 0; j < k; j++) {
            if (j == ci) { if (counts[j] == 1) computeSi = 0; else diC[ki + j] /= (counts[j] - 1); }
            else diC[ki + j] /= counts[j];
        }
        double a_i = diC[ki + ci], b_i;
        if (ci == 0) b_i = diC[ki + 1]; else b_i = diC[ki];
        for (int j = 1; j < k; j++) if (j != ci) { if (b_i > diC[ki + j]) b_i = diC[ki + j]; }
        si[i] = (computeSi && (b_i != a_i)) ? (b_i - a_i) / fmax(a_i, b_i) : 0.0;
    }
    free(diC); free(counts);
}
double oracle_median_silhouette(int n, int k, const int *cl, const double *dcond) {
    double *si = (double *)xmalloc(sizeof(double) * (size_t)n);
    silhouette_widths(n, k, cl, dcond, si);
    double m = r_median(si, n);
    free(si);
    return m;
}
void oracle_silhouette_widths(int n, int k, const int *cl, const double *dcond, double *si) {
    silhouette_widths(n, k, cl, dcond, si);
}

/* Pearson correlation of two p-vectors, as stats::cor(x, y) does it (cov.c,    */
/* two-pass means + LDOUBLE accumulation, clamp to [-1,1]).                     */
static double r_cor_vec(const double *x, const double *y, int p) {
    LD sum = 0, tmp;
    for (int k = 0; k < p; k++) sum += x[k];
    tmp = sum / p; sum = 0;
    for (int k = 0; k < p; k++) sum += (x[k] - tmp);
    LD xm = (double)(tmp + sum / p);
    sum = 0; for (int k = 0; k < p; k++) sum += y[k];
    tmp = sum / p; sum = 0;
    for (int k = 0; k < p; k++) sum += (y[k] - tmp);
    LD ym = (double)(tmp + sum / p);
    LD sxx = 0, syy = 0, sxy = 0;
    for (int k = 0; k < p; k++) {
        LD a = x[k] - xm, b = y[k] - ym;
        sxx += a * a; syy += b * b; sxy += a * b;
    }
    int n1 = p - 1;
    double vxx = (double)(sxx / n1), vyy = (double)(syy / n1), vxy = (double)(sxy / n1);
    double sx = sqrt(vxx), sy = sqrt(vyy);
    if (sx == 0 || sy == 0) return NAN;
    LD r = (LD)vxy / ((LD)sx * (LD)sy);
    if (r > 1) r = 1; if (r < -1) r = -1;
    return (double)r;
}

/* clues::get_CH(y, mem, disMethod="1-corr") -- formula per SURVEY.md App. A.6  */
/* (clues 0.6.2.2 source unavailable: UNVERIFIED, principal parity risk).       */
/*   CH = [B/(g-1)] / [W/(n-g)],  B = sum_k n_k d(ybar_k, ybar)^2,              */
/*   W = sum_k sum_{i in k} d(y_i, ybar_k)^2,  d = 1 - Pearson correlation.     */
/* y: n x p row-major.  Called at R/get_opt_hclust.R:144.                       */
double oracle_get_CH_1corr(const double *y, int n, int p, const int *cl, int g) {
    double *cen = (double *)xcalloc((size_t)g * (size_t)p, sizeof(double));
    double *all = (double *)xcalloc((size_t)p, sizeof(double));
    int *cnt = (int *)xcalloc((size_t)g, sizeof(int));
    for (int i = 0; i < n; i++) {
        int c = cl[i] - 1; cnt[c]++;
        for (int k = 0; k < p; k++) { cen[(size_t)c * p + k] += y[(size_t)i * p + k]; all[k] += y[(size_t)i * p + k]; }
    }
    for (int c = 0; c < g; c++) for (int k = 0; k < p; k++) cen[(size_t)c * p + k] /= cnt[c];
    for (int k = 0; k < p; k++) all[k] /= n;
    double B = 0, W = 0;
    for (int c = 0; c < g; c++) { double d = 1.0 - r_cor_vec(cen + (size_t)c * p, all, p); B += cnt[c] * d * d; }
    for (int i = 0; i < n; i++) { double d = 1.0 - r_cor_vec(y + (size_t)i * p, cen + (size_t)(cl[i] - 1) * p, p); W += d * d; }
    free(cen); free(all); free(cnt);
    return (B / (g - 1)) / (W / (n - g));
}
/* clusterCrit::intCriteria(.., "Calinski_Harabasz"): Euclidean BGSS/WGSS form  */
/* (R/get_opt_hclust.R:105, N.cluster-given branch; value is returned only).    */
static double ch_euclid(const double *y, int n, int p, const int *cl, int g) {
    double *cen = (double *)xcalloc((size_t)g * (size_t)p, sizeof(double));
    double *all = (double *)xcalloc((size_t)p, sizeof(double));
    int *cnt = (int *)xcalloc((size_t)g, sizeof(int));
    for (int i = 0; i < n; i++) {
        int c = cl[i] - 1; cnt[c]++;
        for (int k = 0; k < p; k++) { cen[(size_t)c * p + k] += y[(size_t)i * p + k]; all[k] += y[(size_t)i * p + k]; }
    }
    for (int c = 0; c < g; c++) for (int k = 0; k < p; k++) cen[(size_t)c * p + k] /= cnt[c];
    for (int k = 0; k < p; k++) all[k] /= n;
    double B = 0, W = 0;
    for (int c = 0; c < g; c++) for (int k = 0; k < p; k++) { double d = cen[(size_t)c * p + k] - all[k]; B += cnt[c] * d * d; }
    for (int i = 0; i < n; i++) for (int k = 0; k < p; k++) { double d = y[(size_t)i * p + k] - cen[(size_t)(cl[i] - 1) * p + k]; W += d * d; }
    free(cen); free(all); free(cnt);
    return (B / (g - 1)) / (W / (n - g));
}

/* isSymmetric(mat): square and all.equal(mat, t(mat), tol = 100*eps)           */
static int r_is_symmetric(const double *mat, int n, int p) {
    if (n != p) return 0;
    LD num = 0, den = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
        num += fabsl((LD)mat[(size_t)i * n + j] - (LD)mat[(size_t)j * n + i]);
        den += fabsl((LD)mat[(size_t)i * n + j]);
    }
    double tol = 100 * 2.220446049250313e-16;
    LD xy = num;
    if (den > 0 && (den / ((LD)n * n)) > tol) xy = num / den;
    return xy < tol;
}

/* Row standardisation t(scale(t(mat))) (R/get_opt_hclust.R:71): colMeans with  */
/* LDOUBLE sum, sd = sqrt(sum(v^2)/(p-1)) with LDOUBLE sum.  In place.          */
void oracle_scale_rows(double *mat, int n, int p) {
    for (int i = 0; i < n; i++) {
        double *r = mat + (size_t)i * p;
        LD s = 0; for (int k = 0; k < p; k++) s += r[k];
        double mean = (double)(s / p);
        for (int k = 0; k < p; k++) r[k] = r[k] - mean;
        LD ss = 0; for (int k = 0; k < p; k++) ss += (LD)(r[k] * r[k]);
        double sd = sqrt((double)ss / (double)((p - 1) > 1 ? (p - 1) : 1));
        for (int k = 0; k < p; k++) r[k] = r[k] / sd;
    }
}
/* d = as.dist(1 - cor(t(mat))) (R/get_opt_hclust.R:72), cov.c cov_complete1    */
/* semantics: two-pass LDOUBLE means, LDOUBLE cross products / (p-1), divide by */
/* sqrt(diag) products, clamp.  mat: n x p row-major.  dcond: n(n-1)/2.         */
void oracle_cor_dist(const double *mat, int n, int p, double *dcond) {
    double *xm = (double *)xmalloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) {
        const double *x = mat + (size_t)i * p;
        LD sum = 0; for (int k = 0; k < p; k++) sum += x[k];
        LD tmp = sum / p; sum = 0;
        for (int k = 0; k < p; k++) sum += (x[k] - tmp);
        xm[i] = (double)(tmp + sum / p);
    }
    LD *cen = (LD *)xmalloc(sizeof(LD) * (size_t)n * (size_t)p);
    for (int i = 0; i < n; i++) for (int k = 0; k < p; k++) cen[(size_t)i * p + k] = (LD)mat[(size_t)i * p + k] - (LD)xm[i];
    double *sdv = (double *)xmalloc(sizeof(double) * (size_t)n);
    int n1 = p - 1;
    for (int i = 0; i < n; i++) {
        LD s = 0; const LD *a = cen + (size_t)i * p;
        for (int k = 0; k < p; k++) s += a[k] * a[k];
        sdv[i] = sqrt((double)(s / n1));
    }
    size_t l = 0;
    for (int i = 0; i < n; i++) {
        const LD *a = cen + (size_t)i * p;
        for (int j = i + 1; j < n; j++, l++) {
            const LD *b = cen + (size_t)j * p;
            LD s = 0;
            for (int k = 0; k < p; k++) s += a[k] * b[k];
            double cov = (double)(s / n1);
            LD r = (LD)cov / ((LD)sdv[i] * (LD)sdv[j]);
            if (r > 1) r = 1; if (r < -1) r = -1;
            dcond[l] = 1.0 - (double)r;
        }
    }
    free(xm); free(cen); free(sdv);
}

/* ------------------------------------------------------------------------- */
/* The decision log ("tests must log the best-vs-     */
/* second-best margin of every model-selection decision so sub-1e-9 flips are   */
/* attributable").  While on, every get_opt_hclust call leaves one row of       */
/* OR_DEC_COLS doubles -- the row include/sharp_hip.h documents for             */
/* sharp_last_decisions, computed by the same arithmetic: [0] level (0 base     */
/* clustering, 1 a fold's wMetaC, 2 sMetaC across a block's folds, 3 sMetaC      */
/* across blocks, -1 a direct call) [1] block [2] k [3] fold [4] n [5] branch    */
/* (0 msil, 1 CH, 2 height gap, 3 N.cluster given) [6] chosen number of clusters */
/* [7] exact ties at the deciding maximum (R/get_opt_hclust.R:162-168,194-195)   */
/* [8] that maximum [9] the largest value strictly below it [10] max(msil) -     */
/* sil.thre [11] the height rule's ratio (:196-210) [12] sMetaC's two-cluster    */
/* override (R/sMetaC.R:139-148): clusters of the column taken, 0 = none         */
/* [13] candidate levels.  Where the call sits in the run is thread-local state  */
/* set by the drivers below (the task loops are OpenMP loops).                   */
/* ------------------------------------------------------------------------- */
#define OR_DEC_COLS 14
static _Thread_local int dec_level = -1, dec_k = 0, dec_fold = 0;
static int dec_block = 0;                     /* (the block loop of SHARP_unlimited is serial, R/SHARP_unlimited.R:125) */
static int dec_on = 0;
static double *dec_rows = NULL; static size_t dec_n = 0, dec_cap = 0;
static void dec_set(int level, int k, int fold) { dec_level = level; dec_k = k; dec_fold = fold; }
void oracle_decision_log(int enable) {
    #pragma omp critical(or_declog)
    { dec_on = enable != 0; dec_n = 0; }
}
static void dec_add(const double *row) {
    #pragma omp critical(or_declog)
    {
        if (dec_n == dec_cap) { dec_cap = dec_cap ? 2 * dec_cap : 1024; dec_rows = (double *)realloc(dec_rows, sizeof(double) * OR_DEC_COLS * dec_cap); }
        memcpy(dec_rows + dec_n * OR_DEC_COLS, row, sizeof(double) * OR_DEC_COLS);
        dec_n++;
    }
}
static void dec_override(int level, int block, int k_taken) {
    #pragma omp critical(or_declog)
    {
        for (size_t r = dec_n; r-- > 0;) {
            double *row = dec_rows + r * OR_DEC_COLS;
            if ((int)row[0] == level && (int)row[1] == block) { row[12] = k_taken; break; }
        }
    }
}
static int dec_cmp(const void *a, const void *b) {
    const double *x = (const double *)a, *y = (const double *)b;
    for (int c = 0; c < 4; c++) if (x[c] != y[c]) return x[c] < y[c] ? -1 : 1;
    return x[OR_DEC_COLS] < y[OR_DEC_COLS] ? -1 : 1;        /* (insertion order: the sort is stable through the extra column) */
}
int oracle_last_decisions(double *rows, int cap_rows) {
    int nr;
    #pragma omp critical(or_declog)
    {
        nr = (int)dec_n;
        double *tmp = (double *)xmalloc(sizeof(double) * (OR_DEC_COLS + 1) * (size_t)(nr > 0 ? nr : 1));
        for (int i = 0; i < nr; i++) { memcpy(tmp + (size_t)i * (OR_DEC_COLS + 1), dec_rows + (size_t)i * OR_DEC_COLS, sizeof(double) * OR_DEC_COLS); tmp[(size_t)i * (OR_DEC_COLS + 1) + OR_DEC_COLS] = i; }
        qsort(tmp, (size_t)nr, sizeof(double) * (OR_DEC_COLS + 1), dec_cmp);
        for (int i = 0; i < nr && i < cap_rows; i++) memcpy(rows + (size_t)i * OR_DEC_COLS, tmp + (size_t)i * (OR_DEC_COLS + 1), sizeof(double) * OR_DEC_COLS);
        free(tmp);
    }
    return nr;
}
static void decision_row(int N_cluster, double sil_thre, double height_Ntimes, int n, int kmin, int nk, const double *msil, const double *CH,
                         const double *height, int oind, int branch, double *row) {
    for (int c = 0; c < OR_DEC_COLS; c++) row[c] = NAN;
    row[0] = dec_level; row[1] = dec_block; row[2] = dec_k; row[3] = dec_fold; row[4] = n;
    row[5] = branch; row[6] = kmin + oind - 1; row[12] = 0; row[13] = nk;
    if (N_cluster > 0) { row[5] = 3; row[6] = N_cluster; row[7] = 1; row[8] = msil[0]; row[13] = 1; return; }
    double mx = msil[0];
    for (int c = 1; c < nk; c++) if (msil[c] > mx) mx = msil[c];
    row[10] = mx - sil_thre;
    const double *val = branch == 0 ? msil : CH;
    double best = branch == 0 ? mx : val[0];
    if (branch != 0) for (int c = 1; c < nk; c++) if (val[c] > best) best = val[c];
    int ties = 0;
    double second = NAN;
    for (int c = 0; c < nk; c++) {
        if (val[c] == best) ties++;
        else if (val[c] < best && (!(second == second) || val[c] > second)) second = val[c];
    }
    row[7] = ties; row[8] = best; row[9] = second;
    if (branch >= 1 && (branch == 2 || CH[0] == best)) {
        int first = 1;
        for (int c = 1; c < nk; c++) if (CH[c] > CH[0]) first = 0;
        if (first) {
            int nh = n - 1, t0 = nh > 10 ? nh - 10 : 0, tl = nh - t0;
            const double *tmp = height + t0;
            double rmax = NAN;
            for (int i = 0; i + 1 < tl; i++) {
                double dif = tmp[i + 1] - tmp[i], den = (height_Ntimes - 1) * tmp[i];
                double r = den > 0 ? dif / den : (dif > 0 ? INFINITY : 0.0);
                if (branch == 2) { if (dif > den) { rmax = r; break; } }
                else if (!(rmax == rmax) || r > rmax) rmax = r;
            }
            row[11] = rmax;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* get_opt_hclust  (R/get_opt_hclust.R:33-244).                               */
/* mat: n x p row-major (features) or n x n symmetric similarity.             */
/* N_cluster: 0 = NULL.  Outputs: f[n]; v[n*nk] column-major (nk columns for  */
/* k = minN..min(maxN,n-1), or 1 column if N_cluster given); msil, CHind [nk];*/
/* height[n-1]; *maxsil; *optN; *nk_out; *branch: 0 silhouette, 1 CH, 2 height*/
/* Returns OR_OK or an error code where R would stop().                       */
/* ------------------------------------------------------------------------- */
int oracle_get_opt_hclust(const double *mat_in, int n, int p, int hmethod, int N_cluster,
                          int minN, int maxN, double sil_thre, double height_Ntimes,
                          int *f, int *v, double *msil, double *CHind, double *maxsil,
                          double *height, int *optN, int *nk_out, int *branch) {
    if (n < 2) return OR_ERR_ARG;
    size_t len = (size_t)n * (size_t)(n - 1) / 2;
    double *d = (double *)xmalloc(sizeof(double) * len);
    double *mat = (double *)xmalloc(sizeof(double) * (size_t)n * (size_t)p);
    memcpy(mat, mat_in, sizeof(double) * (size_t)n * (size_t)p);
    if (r_is_symmetric(mat, n, p)) {                       /* :66-69 */
        size_t l = 0;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++, l++) d[l] = 1.0 - mat[(size_t)j * n + i];
    } else {                                               /* :70-74 */
        oracle_scale_rows(mat, n, p);
        oracle_cor_dist(mat, n, p, d);
    }
    double *dwork = (double *)xmalloc(sizeof(double) * len);
    memcpy(dwork, d, sizeof(double) * len);
    int *ia = (int *)xmalloc(sizeof(int) * (size_t)n), *ib = (int *)xmalloc(sizeof(int) * (size_t)n);
    oracle_hclust(n, hmethod, dwork, ia, ib, height);      /* :76-83 */
    free(dwork);
    int rc = OR_OK;
    if (branch) *branch = 0;
    if (N_cluster > 0) {                                   /* :90-107 */
        if (N_cluster < 2) { rc = OR_ERR_ARG; goto done; }
        cutree_k(n, ia, ib, N_cluster, v);
        memcpy(f, v, sizeof(int) * (size_t)n);
        int kk = 0; for (int i = 0; i < n; i++) if (v[i] > kk) kk = v[i];
        msil[0] = (kk >= 2 && kk <= n - 1) ? oracle_median_silhouette(n, kk, v, d) : NAN;
        CHind[0] = ch_euclid(mat, n, p, v, kk);
        *maxsil = msil[0]; *optN = N_cluster; *nk_out = 1;
        if (dec_on) { double row[OR_DEC_COLS]; decision_row(N_cluster, sil_thre, height_Ntimes, n, minN, 1, msil, CHind, height, 1, 0, row); dec_add(row); }
        goto done;
    }
    {
        int kmax = maxN < n - 1 ? maxN : n - 1;            /* :113 */
        int nk = kmax - minN + 1;
        if (nk < 1) { rc = OR_ERR_ARG; goto done; }
        *nk_out = nk;
        /* the levels are independent: OpenMP over them when called from serial code (a cross-block sMetaC at 1e7 cells tries
           1801 levels); inside the task-parallel loops of SHARP_large nested parallelism is off and this runs serially */
        #pragma omp parallel for schedule(dynamic, 1)
        for (int c = 0; c < nk; c++) {                     /* :129-154 */
            int k = minN + c;
            int *vc = v + (size_t)c * n;
            cutree_k(n, ia, ib, k, vc);
            msil[c] = oracle_median_silhouette(n, k, vc, d);
            CHind[c] = oracle_get_CH_1corr(mat, n, p, vc, k);
        }
        double mx = msil[0];
        for (int c = 1; c < nk; c++) if (msil[c] > mx) mx = msil[c];
        int ntie = 0; for (int c = 0; c < nk; c++) if (msil[c] == mx) ntie++;   /* :162-168 */
        int want = (ntie > 1) ? (ntie + 1) / 2 : 1, seen = 0, oind = 1;
        for (int c = 0; c < nk; c++) if (msil[c] == mx) { if (++seen == want) { oind = c + 1; break; } }
        if (mx <= sil_thre) {                              /* :194-210 */
            if (branch) *branch = 1;
            int wm = 0; for (int c = 1; c < nk; c++) if (CHind[c] > CHind[wm]) wm = c;
            oind = wm + 1;
            if (oind == 1) {
                int nh = n - 1, t0 = nh > 10 ? nh - 10 : 0, tl = nh - t0;
                const double *tmp = height + t0;
                int pind = -1;
                for (int i = 0; i + 1 < tl; i++) {
                    double dif = tmp[i + 1] - tmp[i];
                    if (dif > (height_Ntimes - 1) * tmp[i]) { pind = i; break; }
                }
                if (pind >= 0) {
                    if (branch) *branch = 2;
                    double opth = (tmp[pind] + tmp[pind + 1]) / 2;
                    int kk = cutree_h_k(n, height, opth);
                    oind = kk - 1;                          /* length(unique(optv)) - 1 */
                }
            }
        }
        if (oind < 1 || oind > nk) { rc = OR_ERR_RANGE; oind = oind < 1 ? 1 : nk; }
        if (dec_on) { double row[OR_DEC_COLS]; decision_row(0, sil_thre, height_Ntimes, n, minN, nk, msil, CHind, height, oind, branch ? *branch : 0, row); dec_add(row); }
        memcpy(f, v + (size_t)(oind - 1) * n, sizeof(int) * (size_t)n);
        int kk = 0; for (int i = 0; i < n; i++) if (f[i] > kk) kk = f[i];
        *optN = kk; *maxsil = mx;
    }
done:
    free(d); free(mat); free(ia); free(ib);
    return rc;
}

/* getrowColor (R/getrowColor.R:17-121): colour index 1..40, wrapping j>40      */
/* (collisions reproduce the reference's silent merge, :59-68).                 */
int oracle_getrowColor(const double *E, int n, int p, int hmethod, int indN, int minN, int maxN,
                       double sil_thre, double height_Ntimes, int *rowColor, double *maxsil) {
    int kmax = maxN < n - 1 ? maxN : n - 1, nk = kmax - minN + 1; if (nk < 1) nk = 1;
    int *f = (int *)xmalloc(sizeof(int) * (size_t)n);
    int *v = (int *)xmalloc(sizeof(int) * (size_t)n * (size_t)nk);
    double *msil = (double *)xmalloc(sizeof(double) * (size_t)nk), *ch = (double *)xmalloc(sizeof(double) * (size_t)nk);
    double *height = (double *)xmalloc(sizeof(double) * (size_t)n);
    int optN, nko, br;
    int rc = oracle_get_opt_hclust(E, n, p, hmethod, indN, minN, maxN, sil_thre, height_Ntimes,
                                   f, v, msil, ch, maxsil, height, &optN, &nko, &br);
    /* unf = unique(as.character(f)): f is numbered by first appearance already */
    for (int i = 0; i < n; i++) { int j = f[i]; j = j % 40; if (j == 0) j = 40; rowColor[i] = (f[i] > 40) ? j : f[i]; }
    free(f); free(v); free(msil); free(ch); free(height);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* helpers for R's unique()/table()/string-ordered votes                      */
/* ------------------------------------------------------------------------- */
static int lexcmp_int(int a, int b) { /* compare decimal strings as R's table() sorts them */
    char sa[16], sb[16];
    snprintf(sa, sizeof sa, "%d", a); snprintf(sb, sizeof sb, "%d", b);
    return strcmp(sa, sb);
}

/* ------------------------------------------------------------------------- */
/* wMetaC (R/wMetaC.R:15-226; getA :242-283, getss :299-311, getnewk :313-320) */
/* nC: N x C column-major ints; labels compare by equality within a column.    */
/* sil_thre: callers in SHARP pass 0.35 explicitly (R/SHARP.R:401,697).        */
/* Outputs: finalC[N] (meta id, the number R stores as a string), x0 (N x ncl, */
/* column-major, caller buffer >= N*maxN cols or NULL), *ncl.  Optional debug  */
/* outs (may be NULL): w1_out[N], S_out[allC*allC] (row-major), *allC_out,     */
/* tf_out[allC].                                                               */
/* ------------------------------------------------------------------------- */
int oracle_wMetaC(const int *nC, int N, int C, int hmethod, int enN, int minN, int maxN,
                  double sil_thre, double height_Ntimes, int *finalC, double *x0, int *ncl_out,
                  double *w1_out, double *S_out, int *allC_out, int *tf_out) {
    int rc = OR_OK;
    /* AA = (1/C) sum_c [lab_ic == lab_jc]; w0 = 4/N * rowSums(AA*(1-AA))  (:24-41) */
    double *w1 = (double *)xmalloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; i++) {
        double rs = 0.0;
        for (int j = 0; j < N; j++) {
            int cnt = 0;
            for (int c = 0; c < C; c++) cnt += (nC[(size_t)c * N + i] == nC[(size_t)c * N + j]);
            if (cnt) { double x = (double)cnt / (double)C; rs += x * (1 - x); }
        }
        double w0 = 4.0 / N * rs;
        w1[i] = (w0 + 0.01) / (1 + 0.01);                  /* :43-44 */
    }
    if (w1_out) memcpy(w1_out, w1, sizeof(double) * (size_t)N);
    /* R = unique(x), x = column-major concatenation of "<label>_<c>" (:60-67) */
    int *cid = (int *)xmalloc(sizeof(int) * (size_t)N * (size_t)C); /* global cluster id 0..allC-1 */
    int allC = 0;
    {
        int *ul = (int *)xmalloc(sizeof(int) * (size_t)N);
        for (int c = 0; c < C; c++) {
            int base = allC, nu = 0;
            for (int i = 0; i < N; i++) {
                int lab = nC[(size_t)c * N + i], q;
                for (q = 0; q < nu; q++) if (ul[q] == lab) break;
                if (q == nu) { ul[nu++] = lab; allC++; }
                cid[(size_t)c * N + i] = base + q;
            }
        }
        free(ul);
    }
    if (allC_out) *allC_out = allC;
    if (allC < 2) { free(w1); free(cid); return OR_ERR_ARG; }
    /* S[a,b] = sum(w1[intersect]) / sum(w1[union]) (:70-77, getss) with R's set  */
    /* orders: intersect = k1 order; union = k1 then (k2 \ k1); LDOUBLE sum().    */
    double *S = (double *)xcalloc((size_t)allC * (size_t)allC, sizeof(double));
    int *col_of = (int *)xmalloc(sizeof(int) * (size_t)allC);
    int *mstart = (int *)xcalloc((size_t)allC + 1, sizeof(int));
    for (int c = 0; c < C; c++) for (int i = 0; i < N; i++) { col_of[cid[(size_t)c * N + i]] = c; mstart[cid[(size_t)c * N + i] + 1]++; }
    for (int a = 0; a < allC; a++) mstart[a + 1] += mstart[a];
    int *memb = (int *)xmalloc(sizeof(int) * (size_t)N * (size_t)C);
    { int *fill = (int *)xcalloc((size_t)allC, sizeof(int));
      for (int c = 0; c < C; c++) for (int i = 0; i < N; i++) { int a = cid[(size_t)c * N + i]; memb[mstart[a] + fill[a]++] = i; }
      free(fill); }
    for (int a = 0; a < allC; a++) {
        S[(size_t)a * allC + a] = 1.0;
        int ca = col_of[a];
        for (int b = a + 1; b < allC; b++) {
            int cb = col_of[b];
            LD si = 0, su = 0; int ni = 0;
            for (int q = mstart[a]; q < mstart[a + 1]; q++) { int i = memb[q]; if (cid[(size_t)cb * N + i] == b) { si += w1[i]; ni++; } }
            double ss = 0.0;
            if (ni) {
                for (int q = mstart[a]; q < mstart[a + 1]; q++) su += w1[memb[q]];
                for (int q = mstart[b]; q < mstart[b + 1]; q++) { int i = memb[q]; if (cid[(size_t)ca * N + i] != a) su += w1[i]; }
                ss = (double)si / (double)su;
            }
            S[(size_t)a * allC + b] = S[(size_t)b * allC + a] = ss;
        }
    }
    free(mstart); free(memb);
    if (S_out) memcpy(S_out, S, sizeof(double) * (size_t)allC * (size_t)allC);
    /* hres = get_opt_hclust(S, ...) (:98-99) */
    int kmax = maxN < allC - 1 ? maxN : allC - 1, nk = kmax - minN + 1; if (nk < 1) nk = 1;
    int *tf = (int *)xmalloc(sizeof(int) * (size_t)allC);
    int *v = (int *)xmalloc(sizeof(int) * (size_t)allC * (size_t)nk);
    double *msil = (double *)xmalloc(sizeof(double) * (size_t)nk), *ch = (double *)xmalloc(sizeof(double) * (size_t)nk);
    double *height = (double *)xmalloc(sizeof(double) * (size_t)allC);
    double maxsil; int optN, nko, br;
    rc |= oracle_get_opt_hclust(S, allC, allC, hmethod, enN, minN, maxN, sil_thre, height_Ntimes,
                                tf, v, msil, ch, &maxsil, height, &optN, &nko, &br);
    if (tf_out) memcpy(tf_out, tf, sizeof(int) * (size_t)allC);
    /* newnC[] <- tf[match(q, R)] ; finalC = names(sort(table(d), decreasing=TRUE)[1]) (:141-143) */
    int *vote = (int *)xmalloc(sizeof(int) * (size_t)N * (size_t)C); /* row-major N x C */
    for (int i = 0; i < N; i++) for (int c = 0; c < C; c++) vote[(size_t)i * C + c] = tf[cid[(size_t)c * N + i]];
    int *uv = (int *)xmalloc(sizeof(int) * (size_t)C), *uc = (int *)xmalloc(sizeof(int) * (size_t)C);
    int *second = (int *)xmalloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < N; i++) {
        int nu = 0;
        for (int c = 0; c < C; c++) {
            int val = vote[(size_t)i * C + c], q;
            for (q = 0; q < nu; q++) if (uv[q] == val) break;
            if (q == nu) { uv[nu] = val; uc[nu] = 0; nu++; }
            uc[q]++;
        }
        /* stable sort by decreasing count, ties in lexicographic level order */
        int best = -1, sec = -1;
        for (int q = 0; q < nu; q++) {
            if (best < 0 || uc[q] > uc[best] || (uc[q] == uc[best] && lexcmp_int(uv[q], uv[best]) < 0)) best = q;
        }
        for (int q = 0; q < nu; q++) {
            if (q == best) continue;
            if (sec < 0 || uc[q] > uc[sec] || (uc[q] == uc[sec] && lexcmp_int(uv[q], uv[sec]) < 0)) sec = q;
        }
        finalC[i] = uv[best];
        second[i] = sec >= 0 ? uv[sec] : -1;
    }
    int ncl = 0;
    int *uC = (int *)xmalloc(sizeof(int) * (size_t)N);
    #define COUNT_UNIQUE() do { ncl = 0; for (int i = 0; i < N; i++) { int q; for (q = 0; q < ncl; q++) if (uC[q] == finalC[i]) break; if (q == ncl) uC[ncl++] = finalC[i]; } } while (0)
    COUNT_UNIQUE();
    if (ncl == 1) {                                        /* :148-161 */
        /* x = sort(table(d), decreasing=TRUE)[1:2]; x[2] >= 0.5 whenever a second
           value exists; with a single value x[2] is NA and R stops -- keep x[1]. */
        for (int i = 0; i < N; i++) { if (second[i] >= 0) finalC[i] = second[i]; else rc |= OR_WARN_NA_VOTE; }
        COUNT_UNIQUE();
    }
    *ncl_out = ncl;
    if (x0) {                                              /* :180-208 */
        for (size_t q = 0; q < (size_t)N * (size_t)ncl; q++) x0[q] = 0.0;
        for (int i = 0; i < N; i++) {
            int xind = 0; for (int q = 0; q < ncl; q++) if (uC[q] == finalC[i]) xind = q;
            int own = 0; for (int c = 0; c < C; c++) own += (vote[(size_t)i * C + c] == uC[xind]);
            x0[(size_t)xind * N + i] = 1.0;
            for (int q = 0; q < ncl; q++) {
                if (q == xind) continue;
                int y = 0; for (int c = 0; c < C; c++) y += (vote[(size_t)i * C + c] == uC[q]);
                if (y) x0[(size_t)q * N + i] = 0.5 * (double)y / (double)own;
            }
        }
    }
    free(w1); free(cid); free(S); free(col_of); free(tf); free(v); free(msil); free(ch); free(height);
    free(vote); free(uv); free(uc); free(second); free(uC);
    return rc;
}

/* The part of sMetaC after the centroids (R/sMetaC.R:67-151): S = cor(centroids), the k-range rule by the number of   */
/* cells, get_opt_hclust(S), the second-best override.  aG: nC x p row-major centroids; tf_out[nC].                     */
static int smetac_core(const double *aG, int nC, int p, long long ncells_ll, int hmethod, int finalN, int minN, int maxN,
                       double sil_thre, double height_Ntimes, int *tf_out) {
    int rc = OR_OK;
    /* S = cor between centroids, diag 1 (:67-85) */
    double *S = (double *)xmalloc(sizeof(double) * (size_t)nC * (size_t)nC);
    for (int a = 0; a < nC; a++) {
        S[(size_t)a * nC + a] = 1.0;
        for (int b = a + 1; b < nC; b++) S[(size_t)a * nC + b] = S[(size_t)b * nC + a] = r_cor_vec(aG + (size_t)a * p, aG + (size_t)b * p, p);
    }
    /* k-range adjustment (:103-119) */
    long long ncells = ncells_ll; int mm = (int)(ncells / 10000);
    if (ncells < 1000000) {
        int baseN = mm > 2 ? mm : 2; if (baseN > 10) baseN = 10;
        int mx = maxN < nC ? maxN : nC;
        if (minN == 2 && mx - baseN >= 3) minN = baseN;
    } else {
        int mm3 = (int)(ncells / 50000), mm2 = (int)(ncells / 5000);
        if (mm2 > maxN) maxN = mm2;
        if (mm3 > minN) minN = mm3;
    }
    int kmax = maxN < nC - 1 ? maxN : nC - 1, nk = kmax - minN + 1; if (nk < 1) nk = 1;
    int *f = (int *)xmalloc(sizeof(int) * (size_t)nC);
    int *v = (int *)xmalloc(sizeof(int) * (size_t)nC * (size_t)nk);
    double *msil = (double *)xmalloc(sizeof(double) * (size_t)nk), *ch = (double *)xmalloc(sizeof(double) * (size_t)nk);
    double *height = (double *)xmalloc(sizeof(double) * (size_t)nC);
    double maxsil; int optN, nko, br;
    rc |= oracle_get_opt_hclust(S, nC, nC, hmethod, finalN, minN, maxN, sil_thre, height_Ntimes,
                                f, v, msil, ch, &maxsil, height, &optN, &nko, &br);   /* :128-129 */
    int *tf = f;
    int nuf = 0; { int mxf = 0; for (int t = 0; t < nC; t++) if (f[t] > mxf) mxf = f[t]; nuf = mxf; }
    if (nko > 1 && nuf == 2 && maxsil > sil_thre) {        /* :139-148 */
        double *s0 = (double *)xmalloc(sizeof(double) * (size_t)nko);
        memcpy(s0, msil, sizeof(double) * (size_t)nko);
        qsort(s0, (size_t)nko, sizeof(double), cmp_dbl);
        double s1 = s0[nko - 2];
        free(s0);
        /* s2 = which(s0 == s1): if several columns tie R would pick a matrix (quirk 9);
           take the first. */
        int s2 = 0; for (int c = 0; c < nko; c++) if (msil[c] == s1) { s2 = c; break; }
        tf = v + (size_t)s2 * nC;
        if (dec_on) dec_override(dec_level, dec_block, minN + s2);
    }
    for (int t = 0; t < nC; t++) tf_out[t] = tf[t];
    free(S); free(f); free(v); free(msil); free(ch); free(height);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* sMetaC (R/sMetaC.R:17-210).  labels[n]: ints, equal <=> same string;       */
/* sE1: n x p row-major.  finalN: 0 = NULL.  Outputs finalColor[n] (tf value), */
/* tf[nC], *nC_out.  `folds` is unused in the reference.                       */
/* ------------------------------------------------------------------------- */
int oracle_sMetaC(const int *labels, const double *sE1, int n, int p, int hmethod, int finalN,
                  int minN, int maxN, double sil_thre, double height_Ntimes,
                  int *finalColor, int *tf_out, int *nC_out) {
    int rc = OR_OK;
    /* R = unique(rerowColor) (:21) -- first appearance */
    int *uid = (int *)xmalloc(sizeof(int) * (size_t)n);
    int *ulab = (int *)xmalloc(sizeof(int) * (size_t)n);
    int nC = 0;
    {   /* hash-free but O(n * nC); fine for oracle sizes; speed up with a direct map if labels are small */
        int maxlab = 0; for (int i = 0; i < n; i++) if (labels[i] > maxlab) maxlab = labels[i];
        int minlab = 0; for (int i = 0; i < n; i++) if (labels[i] < minlab) minlab = labels[i];
        if (minlab >= 0 && maxlab < (1 << 26)) {
            int *map = (int *)xmalloc(sizeof(int) * (size_t)(maxlab + 1));
            for (int q = 0; q <= maxlab; q++) map[q] = -1;
            for (int i = 0; i < n; i++) { if (map[labels[i]] < 0) { map[labels[i]] = nC; ulab[nC++] = labels[i]; } uid[i] = map[labels[i]]; }
            free(map);
        } else {
            for (int i = 0; i < n; i++) { int q; for (q = 0; q < nC; q++) if (ulab[q] == labels[i]) break; if (q == nC) ulab[nC++] = labels[i]; uid[i] = q; }
        }
    }
    *nC_out = nC;
    if (nC < 2) { free(uid); free(ulab); return OR_ERR_ARG; }
    /* aG[t,] = colMeans(sE1[cluster t, ]) (:58-63): LDOUBLE column sums / count */
    LD *acc = (LD *)xcalloc((size_t)nC * (size_t)p, sizeof(LD));
    int *cnt = (int *)xcalloc((size_t)nC, sizeof(int));
    for (int i = 0; i < n; i++) { int t = uid[i]; cnt[t]++; for (int k = 0; k < p; k++) acc[(size_t)t * p + k] += sE1[(size_t)i * p + k]; }
    double *aG = (double *)xmalloc(sizeof(double) * (size_t)nC * (size_t)p);
    for (int t = 0; t < nC; t++) for (int k = 0; k < p; k++) aG[(size_t)t * p + k] = (double)(acc[(size_t)t * p + k] / cnt[t]);
    free(acc);
    rc |= smetac_core(aG, nC, p, (long long)n, hmethod, finalN, minN, maxN, sil_thre, height_Ntimes, tf_out);
    for (int i = 0; i < n; i++) finalColor[i] = tf_out[uid[i]];  /* :182 */
    free(uid); free(ulab); free(cnt); free(aG);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* shared tail: merge clusters with < 10 cells (R/SHARP.R:418-427,816-825;     */
/* R/SHARP_unlimited.R:168-177) and relabel by first appearance                */
/* (R/SHARP.R:429-443,828-843).                                                */
/* ------------------------------------------------------------------------- */
static void merge_small_clusters(int *lab, int n) {
    int mx = 0; for (int i = 0; i < n; i++) if (lab[i] > mx) mx = lab[i];
    int *cnt = (int *)xcalloc((size_t)mx + 1, sizeof(int));
    for (int i = 0; i < n; i++) cnt[lab[i]]++;
    int mn = -1;
    for (int q = 1; q <= mx; q++) if (cnt[q] > 0 && cnt[q] < 10) { if (mn < 0) mn = q; }
    if (mn >= 0) for (int i = 0; i < n; i++) if (cnt[lab[i]] < 10) lab[i] = mn;
    free(cnt);
}
static int relabel_first_appearance(int *lab, int n) {
    int mx = 0; for (int i = 0; i < n; i++) if (lab[i] > mx) mx = lab[i];
    int *map = (int *)xcalloc((size_t)mx + 1, sizeof(int));
    int k = 0;
    for (int i = 0; i < n; i++) { if (!map[lab[i]]) map[lab[i]] = ++k; lab[i] = map[lab[i]]; }
    free(map);
    return k;
}

/* ------------------------------------------------------------------------- */
/* SHARP_small (R/SHARP.R:339-454).  X: m x n column-major (after prep).       */
/* flag: log-transform.  rN_seed: integer seed or 0.5 (unseeded sentinel).     */
/* Outputs: pred[n] (1..G by first appearance), viE (n x p row-major, or NULL),*/
/* enrp (n x K column-major colour ids, or NULL), x0 (N x ncl col-major, NULL),*/
/* ------------------------------------------------------------------------- */
int oracle_SHARP_small(const double *X, int m, int n, int K, int p, int hmethod, int N_cluster,
                       int indN, int minN, int maxN, double sil_thre, double height_Ntimes,
                       int flag, double rN_seed, int *pred, double *viE, int *enrp_out,
                       double *x0, int *ncl_x0) {
    int rc = OR_OK;
    int8_t *tern = (int8_t *)xmalloc((size_t)m * (size_t)p);
    double *E = (double *)xmalloc(sizeof(double) * (size_t)n * (size_t)p);
    double *enE = (double *)xcalloc((size_t)n * (size_t)p, sizeof(double));
    int *enrp = (int *)xmalloc(sizeof(int) * (size_t)n * (size_t)K);
    for (int k = 1; k <= K; k++) {                         /* :350-387 */
        double seedn = (rN_seed == 0.5) ? 0.5 : 50 + rN_seed + k;
        oracle_ranM(m, p, seedn, tern);
        oracle_project(X, m, n, tern, p, flag, E);
        double maxsil;
        dec_set(0, k - 1, 0);
        rc |= oracle_getrowColor(E, n, p, hmethod, indN, minN, maxN, sil_thre, height_Ntimes,
                                 enrp + (size_t)(k - 1) * n, &maxsil);
        dec_set(-1, 0, 0);
        for (size_t q = 0; q < (size_t)n * (size_t)p; q++) enE[q] += E[q];   /* :398 */
    }
    int ncl;
    dec_set(1, 0, 0);
    rc |= oracle_wMetaC(enrp, n, K, hmethod, N_cluster, minN, maxN, sil_thre, height_Ntimes,
                        pred, x0, &ncl, NULL, NULL, NULL, NULL);             /* :401 */
    dec_set(-1, 0, 0);
    if (ncl_x0) *ncl_x0 = ncl;
    if (viE) for (size_t q = 0; q < (size_t)n * (size_t)p; q++) viE[q] = enE[q] / K;  /* :416 */
    if (N_cluster <= 0 && n > 10000) merge_small_clusters(pred, n);          /* :418-427 */
    relabel_first_appearance(pred, n);                                       /* :429-443 */
    if (enrp_out) memcpy(enrp_out, enrp, sizeof(int) * (size_t)n * (size_t)K);
    free(tern); free(E); free(enE); free(enrp);
    return rc;
}

/* fold assignment (R/SHARP.R:513-536). folds[i] in 1..T   */
int oracle_make_folds(int ncells, int ng, int *folds) {
    int T = (ncells + ng - 1) / ng;
    if (T > 1) {
        for (int i = 0; i < ncells; i++) folds[i] = i / ng + 1;
        int nt = ncells - (T - 2) * ng;
        int start = (T - 2) * ng;                   /* first cell (0-based) of fold T-1 */
        for (int q = nt / 2; q < ng; q++) { int idx = start + q; if (idx < ncells) folds[idx] = T; }
    } else {
        for (int i = 0; i < ncells; i++) folds[i] = 1;
    }
    return T;
}

/* ------------------------------------------------------------------------- */
/* SHARP_large (R/SHARP.R:478-851).  tern_in: optional K projectors, each      */
/* m x p row-major, concatenated (the rM list, :539-549); NULL -> ranM here.   */
/* Outputs: pred[n]; viE (n x p, original cell order) or NULL.                 */
/* nthreads: OpenMP threads over the K*T task grid (the %dopar% at :554).      */
/* ------------------------------------------------------------------------- */
/* R's round(x, 1): nmath/fround.c of R >= 4.0.0 restated (R core: unpinned third-party dependency of the reference):  */
/* the closer of floor(10x)/10 and ceil(10x)/10, the even multiple on a tie.  Used by R/SHARP_unlimited2.R:410.        */
/* Stage clocks for bench.py's cpu_baseline leg (per-stage seconds beside the GPU's): seconds spent in each stage of  */
/* SHARP_large / SHARP_unlimited since the last reset.  [0] projector build (ranM), [1] RP matmul (log2 + projection; THREAD-seconds:  */
/* summed over the OpenMP threads of the task grid), [2] base clustering (getrowColor; thread-seconds), [3] per-fold wMetaC (wall),   */
/* [4] sMetaC within blocks (wall), [5] cross-block sMetaC + relabel of SHARP_unlimited (wall), [6] wall of the K*T task loop.        */
#include <omp.h>
static double oracle_stage_s[7];
void oracle_stage_seconds(double *out, int reset) {
    if (out) for (int q = 0; q < 7; q++) out[q] = oracle_stage_s[q];
    if (reset) for (int q = 0; q < 7; q++) oracle_stage_s[q] = 0.0;
}
static void stage_add(int q, double dt) {
    #pragma omp atomic
    oracle_stage_s[q] += dt;
}

double oracle_round1(double x) {
    if (x != x || x == 0.0 || x - x != 0.0) return x;
    double sgn = x < 0.0 ? -1.0 : 1.0;
    x = fabs(x);
    if (x >= 1e14) return sgn * x;
    double x10 = 10.0 * x, i10 = floor(x10), xd = i10 / 10.0, xu = ceil(x10) / 10.0;
    double du = xu - x, dd = x - xd;
    return sgn * ((dd < du || (dd == du && fmod(i10, 2.0) == 0.0)) ? xd : xu);
}

/* fpart != 0: SHARP_fpart (R/SHARP_unlimited2.R:297-544) = the same path with flag 2 = log10 (:391), newE1 rounded to  */
/* one decimal (:410), maxN.cluster = 40 for the base tasks only (:421), and no sMetaC: pred receives the per-fold        */
/* ensemble labels "<id>en<t>" packed as t*65536+id in the ORIGINAL cell order (:522-526), viE_out = E1 = enE/K.          */
static int sharp_large_core(const double *X, int m, int n, int K, int p, int ng, int hmethod,
                            int N_cluster, int enpN, int indN, int minN, int maxN, double sil_thre,
                            double height_Ntimes, int flag, const int8_t *tern_in, double rN_seed,
                            int nthreads, int *pred, double *viE_out, int fpart, double *x0_out, int x0_cap, int *x0_ncol) {
    /* x0_out (n x x0_cap doubles, column-major; NULL: not wanted): the soft cluster matrix of :717-731 (block-diagonal over the folds'   */
    /* wMetaC x0), its columns summed per final cluster (:763-773) and its rows put back in the original cell order (:779).                 */
    int rc = OR_OK;
    (void)nthreads;
    int *reind = (int *)xmalloc(sizeof(int) * (size_t)n);
    {   rrng g; rrng_set_seed(&g, (rN_seed == 0.5) ? 20261003u : 50u);      /* :493-499 */
        rrng_sample_perm(&g, n, reind); }
    int shuffle = (n < 100000);                                              /* :504-507 */
    int *folds = (int *)xmalloc(sizeof(int) * (size_t)n);
    int T = oracle_make_folds(n, ng, folds);
    int8_t *tern = NULL; const int8_t *tn = tern_in;
    if (!tn) {                                                               /* :539-549 */
        double ts = omp_get_wtime();
        tern = (int8_t *)xmalloc((size_t)K * (size_t)m * (size_t)p);
        for (int k = 1; k <= K; k++) oracle_ranM(m, p, (rN_seed == 0.5) ? 0.5 : 50 + rN_seed + k, tern + (size_t)(k - 1) * m * p);
        tn = tern;
        stage_add(0, omp_get_wtime() - ts);
    }
    /* fold start offsets (folds are contiguous) */
    int *fstart = (int *)xcalloc((size_t)T + 2, sizeof(int));
    for (int i = 0; i < n; i++) fstart[folds[i] + 1]++;
    for (int t = 1; t <= T + 1; t++) fstart[t] += fstart[t - 1];  /* fstart[t]..fstart[t+1]-1 for fold t (1-based t) */
    int *enrp = (int *)xmalloc(sizeof(int) * (size_t)n * (size_t)K);
    double *enE = (double *)xcalloc((size_t)n * (size_t)p, sizeof(double));
    double *Eall = (double *)xmalloc(sizeof(double) * (size_t)K * (size_t)n * (size_t)p);
    int ntask = K * T;
    double t_loop = omp_get_wtime();
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int task = 0; task < ntask; task++) {                               /* :554-618 */
        double ts0 = omp_get_wtime();
        int k = task / T, t = task % T + 1;
        int c0 = fstart[t], nt = fstart[t + 1] - fstart[t];
        double *Xt = (double *)xmalloc(sizeof(double) * (size_t)m * (size_t)nt);
        for (int j = 0; j < nt; j++) {
            int src = shuffle ? reind[c0 + j] - 1 : c0 + j;
            memcpy(Xt + (size_t)j * m, X + (size_t)src * m, sizeof(double) * (size_t)m);
        }
        double *Et = Eall + ((size_t)k * n + c0) * p;
        oracle_project(Xt, m, nt, tn + (size_t)k * m * p, p, flag, Et);
        if (fpart) for (size_t q = 0; q < (size_t)nt * (size_t)p; q++) Et[q] = oracle_round1(Et[q]);
        double ts1 = omp_get_wtime();
        stage_add(1, ts1 - ts0);
        double maxsil;
        dec_set(0, k, t - 1);
        int r = oracle_getrowColor(Et, nt, p, hmethod, indN, minN, fpart ? 40 : maxN, sil_thre, height_Ntimes,
                                   enrp + (size_t)k * n + c0, &maxsil);
        dec_set(-1, 0, 0);
        if (r) {
            #pragma omp atomic
            rc |= r;
        }
        free(Xt);
        stage_add(2, omp_get_wtime() - ts1);
    }
    stage_add(6, omp_get_wtime() - t_loop);
    for (int k = 0; k < K; k++)                                              /* :629-635, k ascending */
        for (size_t q = 0; q < (size_t)n * (size_t)p; q++) enE[q] += Eall[(size_t)k * n * p + q];
    free(Eall);
    /* per-fold wMetaC (:692-709); label "<id>en<t>" -> (t, id) packed */
    int *fColor = (int *)xmalloc(sizeof(int) * (size_t)n);
    double **fx0 = (double **)xcalloc((size_t)T + 1, sizeof(double *));      /* per fold: wres$x0 (nt x nwC, column-major), wres$nwC (:703-707) */
    int *fncl = (int *)xcalloc((size_t)T + 1, sizeof(int));
    double t_wm = omp_get_wtime();
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int t = 1; t <= T; t++) {
        int c0 = fstart[t], nt = fstart[t + 1] - fstart[t];
        int *sub = (int *)xmalloc(sizeof(int) * (size_t)nt * (size_t)K);
        for (int k = 0; k < K; k++) memcpy(sub + (size_t)k * nt, enrp + (size_t)k * n + c0, sizeof(int) * (size_t)nt);
        int *fc = (int *)xmalloc(sizeof(int) * (size_t)nt); int ncl;
        int capc = K * (maxN > 40 ? maxN : 40) + 2;        /* (the meta-clusters of a fold cannot outnumber its K * maxN base clusters) */
        if (x0_out && !fpart) fx0[t] = (double *)xmalloc(sizeof(double) * (size_t)nt * (size_t)capc);
        dec_set(1, 0, t - 1);
        int r = oracle_wMetaC(sub, nt, K, hmethod, enpN, minN, maxN, sil_thre, height_Ntimes, fc, fx0[t], &ncl, NULL, NULL, NULL, NULL);
        dec_set(-1, 0, 0);
        fncl[t] = ncl;
        if (r) {
            #pragma omp atomic
            rc |= r;
        }
        for (int j = 0; j < nt; j++) fColor[c0 + j] = t * 65536 + fc[j];
        free(sub); free(fc);
    }
    stage_add(3, omp_get_wtime() - t_wm);
    int *S = (int *)xmalloc(sizeof(int) * (size_t)n);
    double *E1 = (double *)xmalloc(sizeof(double) * (size_t)n * (size_t)p);
    for (size_t q = 0; q < (size_t)n * (size_t)p; q++) E1[q] = enE[q] / K;   /* :750 */
    if (fpart) {
        for (int i = 0; i < n; i++) {
            int dst = shuffle ? reind[i] - 1 : i;
            pred[dst] = fColor[i];
            if (viE_out) memcpy(viE_out + (size_t)dst * p, E1 + (size_t)i * p, sizeof(double) * (size_t)p);
        }
        free(reind); free(folds); free(tern); free(fstart); free(enrp); free(enE); free(fColor); free(S); free(E1); free(fx0); free(fncl);
        return rc;
    }
    /* column of sx0 (:717-731) of fold t's j-th cluster (uC = unique(fColor), :714: fold after fold, first appearance inside a fold -- */
    /* the column order of the fold's own x0, R/wMetaC.R:180-208)                                                                        */
    int *coff = (int *)xcalloc((size_t)T + 2, sizeof(int));
    for (int t = 1; t <= T; t++) coff[t + 1] = coff[t] + fncl[t];
    int lenuC = coff[T + 1], ncol = 0;
    int *colmap = (int *)xmalloc(sizeof(int) * (size_t)(lenuC > 0 ? lenuC : 1));   /* sx0 column -> x0 column */
    if (T == 1) {
        /* :738-746 then :828: as.numeric("<id>en1") is NA for every cell -> one cluster */
        for (int i = 0; i < n; i++) S[i] = 1;
        for (int q = 0; q < lenuC; q++) colmap[q] = q;     /* x0 = sx0 (:746) */
        ncol = lenuC;
    } else {
        int *tf = (int *)xmalloc(sizeof(int) * (size_t)n); int nCu;
        double t_sm = omp_get_wtime();
        dec_set(2, 0, 0);
        rc |= oracle_sMetaC(fColor, E1, n, p, hmethod, N_cluster, minN, maxN, sil_thre, height_Ntimes, S, tf, &nCu);  /* :754 */
        dec_set(-1, 0, 0);
        stage_add(4, omp_get_wtime() - t_sm);
        /* :761-773: sn = length(unique(stf)); x0[, i] = rowSums(sx0[, which(stf == i)]).  (nCu == lenuC: both count unique(fColor).) */
        for (int q = 0; q < lenuC && q < nCu; q++) { colmap[q] = tf[q] - 1; if (tf[q] > ncol) ncol = tf[q]; }
        free(tf);
    }
    if (x0_out) {
        if (ncol > x0_cap) rc |= OR_ERR_ARG;
        else {
            for (size_t q = 0; q < (size_t)n * (size_t)ncol; q++) x0_out[q] = 0.0;
            for (int t = 1; t <= T; t++) {
                int c0 = fstart[t], nt = fstart[t + 1] - fstart[t];
                for (int j = 0; j < fncl[t]; j++) {
                    double *dstc = x0_out + (size_t)colmap[coff[t] + j] * n;
                    for (int i = 0; i < nt; i++) {
                        int dst = shuffle ? reind[c0 + i] - 1 : c0 + i;          /* x0[reind, ] = x0 (:779) */
                        dstc[dst] += fx0[t][(size_t)j * nt + i];
                    }
                }
            }
        }
        if (x0_ncol) *x0_ncol = ncol;
    }
    for (int t = 1; t <= T; t++) free(fx0[t]);
    free(fx0); free(fncl); free(coff); free(colmap);
    for (int i = 0; i < n; i++) {                                            /* :775-783 */
        int dst = shuffle ? reind[i] - 1 : i;
        pred[dst] = S[i];
        if (viE_out) memcpy(viE_out + (size_t)dst * p, E1 + (size_t)i * p, sizeof(double) * (size_t)p);
    }
    if (N_cluster <= 0 && n > 10000) merge_small_clusters(pred, n);          /* :816-825 */
    relabel_first_appearance(pred, n);                                       /* :828-843 */
    free(reind); free(folds); free(tern); free(fstart); free(enrp); free(enE); free(fColor); free(S); free(E1);
    return rc;
}

int oracle_SHARP_large(const double *X, int m, int n, int K, int p, int ng, int hmethod,
                       int N_cluster, int enpN, int indN, int minN, int maxN, double sil_thre,
                       double height_Ntimes, int flag, const int8_t *tern_in, double rN_seed,
                       int nthreads, int *pred, double *viE_out) {
    return sharp_large_core(X, m, n, K, p, ng, hmethod, N_cluster, enpN, indN, minN, maxN, sil_thre, height_Ntimes, flag, tern_in,
                            rN_seed, nthreads, pred, viE_out, 0, NULL, 0, NULL);
}
/* the same with enresults$x0 (R/SHARP.R:717-731,761-779): x0 is n x x0_cap doubles, column-major; *x0_ncol columns are filled */
int oracle_SHARP_large_x0(const double *X, int m, int n, int K, int p, int ng, int hmethod,
                          int N_cluster, int enpN, int indN, int minN, int maxN, double sil_thre,
                          double height_Ntimes, int flag, const int8_t *tern_in, double rN_seed,
                          int nthreads, int *pred, double *viE_out, double *x0, int x0_cap, int *x0_ncol) {
    return sharp_large_core(X, m, n, K, p, ng, hmethod, N_cluster, enpN, indN, minN, maxN, sil_thre, height_Ntimes, flag, tern_in,
                            rN_seed, nthreads, pred, viE_out, 0, x0, x0_cap, x0_ncol);
}

/* ------------------------------------------------------------------------- */
/* SHARP front door, the parts that change numbers (R/SHARP.R:44-318) for an   */
/* already-prepared matrix: defaults, dispatch.  prep/CPM/testlog are host-side*/
/* and restated in oracle_testlog / Python helpers.                            */
/* reduced_ndim <= 0 -> ceiling(log2(ncells)/0.04) (:119-122).  K <= 0 ->      */
/* default 15 / 5 (:254-257,268-271).  maxN <= 0 -> max(40, ceil(n/5000)).     */
/* ------------------------------------------------------------------------- */
int oracle_SHARP(const double *X, int m, int n, int K, int reduced_ndim, int base_ncells,
                 int partition_ncells, int hmethod, int N_cluster, int enpN, int indN, int minN,
                 int maxN, double sil_thre, double height_Ntimes, int flag, const int8_t *tern_in,
                 double rN_seed, int nthreads, int *pred, double *viE, int *p_out, int *K_out) {
    int p = reduced_ndim > 0 ? reduced_ndim : (int)ceil(log2((double)n) / (0.2 * 0.2));
    if (base_ncells <= 0) base_ncells = 5000;
    if (partition_ncells <= 0) partition_ncells = 2000;
    if (hmethod <= 0) hmethod = 1;
    if (minN <= 0) minN = 2;
    if (maxN <= 0) { int c = (n + 4999) / 5000; maxN = c > 40 ? c : 40; }
    if (sil_thre < 0) sil_thre = 0.35;
    if (height_Ntimes <= 0) height_Ntimes = 2;
    if (N_cluster > 0 && n < base_ncells) {                                  /* :181-191 */
        indN = N_cluster; base_ncells = (n + 1) / 2; partition_ncells = (n + 1) / 2;
        if (K <= 0) K = 15;
    }
    int rc;
    if (n < base_ncells) {
        if (K <= 0) K = 15;
        /* SHARP_small ignores rM (:263-264): projectors always from the seed */
        rc = oracle_SHARP_small(X, m, n, K, p, hmethod, N_cluster, indN, minN, maxN, sil_thre, height_Ntimes,
                                flag, rN_seed, pred, viE, NULL, NULL, NULL);
    } else {
        if (K <= 0) K = 5;
        rc = oracle_SHARP_large(X, m, n, K, p, partition_ncells, hmethod, N_cluster, enpN, indN, minN, maxN,
                                sil_thre, height_Ntimes, flag, tern_in, rN_seed, nthreads, pred, viE);
    }
    if (p_out) *p_out = p;
    if (K_out) *K_out = K;
    return rc;
}

/* ------------------------------------------------------------------------- */
/* SHARP_unlimited (R/SHARP_unlimited.R:29-242).  Blocks: nb matrices, each    */
/* m x ncb[b] column-major, concatenated in Xcat.  Output pred[ncells] numbered*/
/* by decreasing cluster size (:180-183).                                      */
/* ------------------------------------------------------------------------- */
static int cmp_size_then_lex(const void *a, const void *b) {
    const int *x = (const int *)a, *y = (const int *)b; /* {count, id} */
    if (x[0] != y[0]) return (y[0] > x[0]) - (y[0] < x[0]);
    return lexcmp_int(x[1], y[1]);
}
int oracle_SHARP_unlimited(const double *Xcat, int m, int nb, const int *ncb, int K, int N_cluster,
                           int minN, int maxN, double rN_seed, int nthreads, int *pred, double *viE_out,
                           int *p_out) {
    int rc = OR_OK;
    int ncells = 0; for (int b = 0; b < nb; b++) ncells += ncb[b];
    int p = (int)ceil(log2((double)ncells) / (0.2 * 0.2));                   /* :65-66 */
    if (minN <= 0) minN = 2;
    if (maxN <= 0) { int c = (ncells + 4999) / 5000; maxN = c > 40 ? c : 40; }
    if (K <= 0) K = 5;
    double t_pr = omp_get_wtime();
    int8_t *tern = (int8_t *)xmalloc((size_t)K * (size_t)m * (size_t)p);     /* :97-104 */
    for (int k = 1; k <= K; k++) oracle_ranM(m, p, (rN_seed == 0.5) ? 0.5 : 50 + rN_seed + k, tern + (size_t)(k - 1) * m * p);
    stage_add(0, omp_get_wtime() - t_pr);
    int *fColor = (int *)xmalloc(sizeof(int) * (size_t)ncells);
    double *E1 = (double *)xmalloc(sizeof(double) * (size_t)ncells * (size_t)p);
    size_t off = 0;
    for (int b = 0; b < nb; b++) {                                           /* :125-149 */
        int nbk = ncb[b];
        int *pb = (int *)xmalloc(sizeof(int) * (size_t)nbk);
        dec_block = b;
        /* SHARP(mat, reduced.ndim=p, prep=FALSE, logflag=FALSE, rM=rM, ensize.K, rN.seed) (:135):
           logflag FALSE -> flag TRUE (log always, R/SHARP.R:225-228); block-level defaults */
        rc |= oracle_SHARP(Xcat + off * (size_t)m, m, nbk, K, p, 0, 0, 0, 0, 0, 0, 0, 0, -1.0, 0, 1, tern,
                           rN_seed, nthreads, pb, E1 + off * (size_t)p, NULL, NULL);
        for (int j = 0; j < nbk; j++) fColor[off + j] = (b + 1) * 65536 + pb[j];   /* "<pred>s<i>" */
        free(pb);
        off += (size_t)nbk;
    }
    int *tf = (int *)xmalloc(sizeof(int) * (size_t)ncells); int nCu;
    double t_mg = omp_get_wtime();
    /* sMetaC(fColor, E1, folds, hmethod, N.cluster, minN, maxN, sil.thre, height.Ntimes) (:163);
       hmethod/sil.thre/height.Ntimes come from block 1's paras = the defaults */
    dec_block = 0;
    dec_set(3, 0, 0);
    rc |= oracle_sMetaC(fColor, E1, ncells, p, 1, N_cluster, minN, maxN, 0.35, 2.0, pred, tf, &nCu);
    dec_set(-1, 0, 0);
    free(tf);
    if (N_cluster <= 0 && ncells > 10000) merge_small_clusters(pred, ncells);       /* :168-177 */
    {   /* x = sort(table(finalrowColor), decreasing=TRUE); map (:180-183) */
        int mx = 0; for (int i = 0; i < ncells; i++) if (pred[i] > mx) mx = pred[i];
        int *cnt = (int *)xcalloc((size_t)mx + 1, sizeof(int));
        for (int i = 0; i < ncells; i++) cnt[pred[i]]++;
        int (*pairs)[2] = (int (*)[2])xmalloc(sizeof(int) * 2 * (size_t)(mx + 1)); int np = 0;
        for (int q = 1; q <= mx; q++) if (cnt[q]) { pairs[np][0] = cnt[q]; pairs[np][1] = q; np++; }
        qsort(pairs, (size_t)np, sizeof(int) * 2, cmp_size_then_lex);
        int *map = (int *)xcalloc((size_t)mx + 1, sizeof(int));
        for (int q = 0; q < np; q++) map[pairs[q][1]] = q + 1;
        for (int i = 0; i < ncells; i++) pred[i] = map[pred[i]];
        free(cnt); free(pairs); free(map);
    }
    stage_add(5, omp_get_wtime() - t_mg);
    if (viE_out) memcpy(viE_out, E1, sizeof(double) * (size_t)ncells * (size_t)p);
    if (p_out) *p_out = p;
    free(tern); free(fColor); free(E1);
    return rc;
}

/* The tail of SHARP_unlimited from the per-(block, cluster) centroids on (R/SHARP_unlimited.R:163-183), for checking a    */
/* sharded run's merge step on ITS OWN centroid tables: sMetaC only ever uses colMeans per label (R/sMetaC.R:58-63), so     */
/* means (nC x p row-major, block order then first appearance inside the block) and counts are all it needs.  ncells = the  */
/* TOTAL number of cells (decides the k range, R/sMetaC.R:103-119, and the default maxN, R/SHARP_unlimited.R:75-77).        */
/* final_id[nC]: 1-based id of every row, numbered by decreasing cluster size.                                             */
int oracle_unlimited_merge(const double *means, const long long *counts, int nC, int p, long long ncells, int N_cluster,
                           int minN, int maxN, int *final_id, int *n_final) {
    if (minN <= 0) minN = 2;                                                 /* :70-72 */
    if (maxN <= 0) { long long c = (ncells + 4999) / 5000; maxN = c > 40 ? (int)c : 40; }   /* :75-77 */
    dec_set(3, 0, 0);
    int rc = smetac_core(means, nC, p, ncells, 1, N_cluster, minN, maxN, 0.35, 2.0, final_id);   /* :163 */
    dec_set(-1, 0, 0);
    int mx = 0; for (int q = 0; q < nC; q++) if (final_id[q] > mx) mx = final_id[q];
    long long *cnt = (long long *)xcalloc((size_t)mx + 1, sizeof(long long));
    for (int q = 0; q < nC; q++) cnt[final_id[q]] += counts[q];
    if (N_cluster <= 0 && ncells > 10000) {                                  /* :168-177 */
        int mn = -1;
        for (int q = 1; q <= mx; q++) if (cnt[q] > 0 && cnt[q] < 10) { mn = q; break; }
        if (mn >= 0) {
            for (int q = 0; q < nC; q++) if (cnt[final_id[q]] < 10) final_id[q] = mn;
            for (int q = 0; q <= mx; q++) cnt[q] = 0;
            for (int q = 0; q < nC; q++) cnt[final_id[q]] += counts[q];
        }
    }
    /* x = sort(table(finalrowColor), decreasing = TRUE): by size, ties in the string order of the ids (:180-183); sizes can
       exceed an int only beyond 2^31 cells, which the int pairs of cmp_size_then_lex do not cover */
    int (*pairs)[2] = (int (*)[2])xmalloc(sizeof(int) * 2 * (size_t)(mx + 1)); int np = 0;
    for (int q = 1; q <= mx; q++) if (cnt[q]) { pairs[np][0] = (int)cnt[q]; pairs[np][1] = q; np++; }
    qsort(pairs, (size_t)np, sizeof(int) * 2, cmp_size_then_lex);
    int *map = (int *)xcalloc((size_t)mx + 1, sizeof(int));
    for (int q = 0; q < np; q++) map[pairs[q][1]] = q + 1;
    for (int q = 0; q < nC; q++) final_id[q] = map[final_id[q]];
    *n_final = np;
    free(cnt); free(pairs); free(map);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* SHARP_unlimited2 (R/SHARP_unlimited2.R:29-292): SHARP_fpart per block, one  */
/* sMetaC over the per-fold ensemble clusters of all blocks, small-cluster      */
/* merge, size-ordered ids.  flag: the testlog decision (1 = log10 transform).  */
/* 0 / negative parameters take the defaults of :39-69.                         */
/* ------------------------------------------------------------------------- */
static int cmp_int3(const void *a, const void *b) {
    const int *x = (const int *)a, *y = (const int *)b;   /* {block, packed label, first cell} */
    if (x[0] != y[0]) return (x[0] > y[0]) - (x[0] < y[0]);
    return (x[1] > y[1]) - (x[1] < y[1]);
}
int oracle_SHARP_unlimited2(const double *Xcat, int m, int nb, const int *ncb, int K, int reduced_ndim, int ng,
                            int hmethod, int N_cluster, int enpN, int indN, int minN, int maxN, double sil_thre,
                            double height_Ntimes, int flag, double rN_seed, int nthreads, int *pred, double *viE_out,
                            int *p_out) {
    int rc = OR_OK;
    int ncells = 0; for (int b = 0; b < nb; b++) ncells += ncb[b];
    if (K <= 0) K = 5;                                                       /* :39-41 */
    int p = reduced_ndim > 0 ? reduced_ndim : (int)ceil(log2((double)ncells) / (0.2 * 0.2));   /* :42-44 */
    if (ng <= 0) ng = 2000;                                                  /* :45-47 */
    if (hmethod <= 0) hmethod = 1;                                           /* :48-50 */
    if (minN <= 0) minN = 2;                                                 /* :51-53 */
    if (maxN <= 0) { int c = (ncells + 4999) / 5000; maxN = c > 40 ? c : 40; }   /* :54-56 */
    if (sil_thre < 0) sil_thre = 0.35;                                       /* :57-59 */
    if (height_Ntimes <= 0) height_Ntimes = 2.0;                             /* :60-62 */
    int8_t *tern = (int8_t *)xmalloc((size_t)K * (size_t)m * (size_t)p);     /* :130-137 */
    for (int k = 1; k <= K; k++) oracle_ranM(m, p, (rN_seed == 0.5) ? 0.5 : 50 + rN_seed + k, tern + (size_t)(k - 1) * m * p);
    int *lab = (int *)xmalloc(sizeof(int) * (size_t)ncells);
    int (*keys)[3] = (int (*)[3])xmalloc(sizeof(int) * 3 * (size_t)ncells);
    double *E1 = (double *)xmalloc(sizeof(double) * (size_t)ncells * (size_t)p);
    size_t off = 0;
    for (int b = 0; b < nb; b++) {                                           /* :146-163 */
        int nbk = ncb[b];
        rc |= sharp_large_core(Xcat + off * (size_t)m, m, nbk, K, p, ng, hmethod, 0, enpN, indN, minN, maxN, sil_thre,
                               height_Ntimes, flag ? 2 : 0, tern, rN_seed, nthreads, lab + off, E1 + off * (size_t)p, 1, NULL, 0, NULL);
        for (int j = 0; j < nbk; j++) { keys[off + j][0] = b; keys[off + j][1] = lab[off + j]; keys[off + j][2] = (int)(off + j); }
        off += (size_t)nbk;
    }
    /* "<fColor>s<i>" (:159): one integer id per distinct (block, label); any injective map does (sMetaC takes unique()) */
    qsort(keys, (size_t)ncells, sizeof(int) * 3, cmp_int3);
    {   int id = 0;
        for (int q = 0; q < ncells; q++) {
            if (q == 0 || keys[q][0] != keys[q - 1][0] || keys[q][1] != keys[q - 1][1]) id++;
            lab[keys[q][2]] = id;
        }
    }
    int *tf = (int *)xmalloc(sizeof(int) * (size_t)ncells); int nCu;
    rc |= oracle_sMetaC(lab, E1, ncells, p, hmethod, N_cluster, minN, maxN, sil_thre, height_Ntimes, pred, tf, &nCu);   /* :184-186 */
    free(tf);
    if (N_cluster <= 0 && ncells > 10000) merge_small_clusters(pred, ncells);       /* :189-198 */
    {   /* size-ordered ids (:200-203) */
        int mx = 0; for (int i = 0; i < ncells; i++) if (pred[i] > mx) mx = pred[i];
        int *cnt = (int *)xcalloc((size_t)mx + 1, sizeof(int));
        for (int i = 0; i < ncells; i++) cnt[pred[i]]++;
        int (*pairs)[2] = (int (*)[2])xmalloc(sizeof(int) * 2 * (size_t)(mx + 1)); int np = 0;
        for (int q = 1; q <= mx; q++) if (cnt[q]) { pairs[np][0] = cnt[q]; pairs[np][1] = q; np++; }
        qsort(pairs, (size_t)np, sizeof(int) * 2, cmp_size_then_lex);
        int *map = (int *)xcalloc((size_t)mx + 1, sizeof(int));
        for (int q = 0; q < np; q++) map[pairs[q][1]] = q + 1;
        for (int i = 0; i < ncells; i++) pred[i] = map[pred[i]];
        free(cnt); free(pairs); free(map);
    }
    if (viE_out) memcpy(viE_out, E1, sizeof(double) * (size_t)ncells * (size_t)p);
    if (p_out) *p_out = p;
    free(tern); free(lab); free(keys); free(E1);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* get_marker_genes, the per-gene statistics (R/get_marker_genes.R:120-152).   */
/* X: genes x cells column-major (a cell is contiguous, stride m); label 1..G.  */
/* out: m x 5 row-major: auc, icluster, pvalue (unadjusted), sparsity, FC.      */
/* Third-party pieces restated: base::rank (average ties), ROCR's "auc"         */
/* (trapezoidal area under the ROC of the rank score, ties as one threshold),   */
/* stats::wilcox.test (normal approximation, correct = TRUE, tie correction;    */
/* the exact small-sample branch n1 < 50 && n2 < 50 without ties is not built). */
/* ------------------------------------------------------------------------- */
typedef struct { double v; int i; } mg_pair;
static int mg_cmp(const void *a, const void *b) {
    const mg_pair *x = (const mg_pair *)a, *y = (const mg_pair *)b;
    return (x->v > y->v) - (x->v < y->v);
}
int oracle_marker_genes(const double *X, int m, int n, const int *label, int G, double theta, int ng, double *out) {
    mg_pair *pr = (mg_pair *)xmalloc(sizeof(mg_pair) * (size_t)n);
    double *rank = (double *)xmalloc(sizeof(double) * (size_t)n);
    long double *sr = (long double *)xmalloc(sizeof(long double) * (size_t)G), *sx = (long double *)xmalloc(sizeof(long double) * (size_t)G);
    int *cs = (int *)xcalloc((size_t)G, sizeof(int));
    char *taken = (char *)xmalloc((size_t)G);
    for (int i = 0; i < n; i++) cs[label[i] - 1]++;
    int rr = ng < G ? ng : G; if (rr < 1) rr = 1;
    for (int g = 0; g < m; g++) {
        double *o = out + (size_t)g * 5;
        int nz = 0;
        for (int i = 0; i < n; i++) { pr[i].v = X[(size_t)i * m + g]; pr[i].i = i; nz += pr[i].v != 0.0; }
        double dp = (double)nz / (double)n;                                     /* :122 */
        if (!(dp > theta)) { o[0] = 0; o[1] = 0; o[2] = 1; o[3] = dp; o[4] = 0; continue; }
        qsort(pr, (size_t)n, sizeof(mg_pair), mg_cmp);
        long double ties = 0;
        for (int a = 0; a < n;) {                                               /* rank(): average ranks */
            int b = a; while (b < n && pr[b].v == pr[a].v) b++;
            double r = 0.5 * ((double)(a + 1) + (double)b);
            for (int q = a; q < b; q++) rank[pr[q].i] = r;
            long double t = (long double)(b - a); ties += t * t * t - t;
            a = b;
        }
        for (int c = 0; c < G; c++) { sr[c] = 0; sx[c] = 0; taken[c] = 0; }
        for (int i = 0; i < n; i++) { sr[label[i] - 1] += rank[i]; sx[label[i] - 1] += X[(size_t)i * m + g]; }   /* aggregate(..., mean) */
        double best_auc = -1; int best_c = -1;
        for (int pick = 0; pick < rr; pick++) {                                 /* order(-s$r)[1:rr] */
            int arg = -1; double top = -1;
            for (int c = 0; c < G; c++) { double mr = (double)(sr[c] / cs[c]); if (!taken[c] && mr > top) { top = mr; arg = c; } }
            if (arg < 0) break;
            taken[arg] = 1;
            /* ROCR prediction/performance("auc"): thresholds = distinct scores, descending; trapezoids */
            long double area = 0; long tp = 0, fp = 0;
            for (int b = n; b > 0;) {
                int a = b - 1; while (a > 0 && pr[a - 1].v == pr[b - 1].v) a--;
                long dtp = 0, dfp = 0;
                for (int q = a; q < b; q++) { if (label[pr[q].i] - 1 == arg) dtp++; else dfp++; }
                area += (long double)dfp * ((long double)tp + 0.5L * (long double)dtp);
                tp += dtp; fp += dfp;
                b = a;
            }
            double auc = (double)(area / ((long double)tp * (long double)fp));
            if (auc > best_auc) { best_auc = auc; best_c = arg; }               /* which.max */
        }
        double n1 = cs[best_c], n2 = n - cs[best_c], nn = n;
        double W = (double)sr[best_c] - n1 * (n1 + 1) / 2;                      /* wilcox.test STATISTIC */
        double z = W - n1 * n2 / 2;
        double sigma = sqrt((n1 * n2 / 12) * ((nn + 1) - (double)ties / (nn * (nn - 1))));
        double corr = z > 0 ? 0.5 : (z < 0 ? -0.5 : 0.0);
        z = (z - corr) / sigma;
        double y1 = (double)(sx[best_c] / cs[best_c]), y2 = -1e300;
        for (int c = 0; c < G; c++) if (c != best_c) { double v = (double)(sx[c] / cs[c]); if (v > y2) y2 = v; }
        o[0] = best_auc; o[1] = best_c + 1; o[2] = erfc(fabs(z) * 0.70710678118654752440); o[3] = dp; o[4] = y1 / y2;
    }
    free(pr); free(rank); free(sr); free(sx); free(cs); free(taken);
    return OR_OK;
}

/* ------------------------------------------------------------------------- */
/* testlog (R/SHARP.R:877-924).  The reference samples cells with the UNSEEDED */
/* global RNG (:884); here the caller passes the sampled (0-based) cell ids.   */
/* Returns flag (1 = log-transform).                                           */
/* ------------------------------------------------------------------------- */
int oracle_testlog(const double *X, int m, int ncells, int p, const int *cells, int sncells, double *msil_out) {
    (void)ncells;
    int8_t *tern = (int8_t *)xmalloc((size_t)m * (size_t)p);
    oracle_ranM(m, p, 5.0, tern);                                            /* :889 */
    double *sE = (double *)xmalloc(sizeof(double) * (size_t)m * (size_t)sncells);
    for (int j = 0; j < sncells; j++) memcpy(sE + (size_t)j * m, X + (size_t)cells[j] * m, sizeof(double) * (size_t)m);
    double *E = (double *)xmalloc(sizeof(double) * (size_t)sncells * (size_t)p);
    int *rc = (int *)xmalloc(sizeof(int) * (size_t)sncells);
    double msil[2];
    for (int k = 0; k < 2; k++) {                                            /* :895-913 */
        oracle_project(sE, m, sncells, tern, p, k == 1, E);
        oracle_getrowColor(E, sncells, p, 1, 0, 2, 40, 0.0, 2.0, rc, &msil[k]);
    }
    if (msil_out) { msil_out[0] = msil[0]; msil_out[1] = msil[1]; }
    free(tern); free(sE); free(E); free(rc);
    return (msil[0] < 0.75 && msil[0] >= 0.95 * msil[1]) ? 1 : 0;           /* :918-922 */
}

/* ------------------------------------------------------------------------- */
/* clues::adjustedRand (R/ARI.R:38): Rand, HA, MA, FM, Jaccard.               */
/* ------------------------------------------------------------------------- */
void oracle_adjusted_rand(const int *a, const int *b, int n, double *out5) {
    int ma = 0, mb = 0;
    for (int i = 0; i < n; i++) { if (a[i] > ma) ma = a[i]; if (b[i] > mb) mb = b[i]; }
    ma++; mb++;
    double *tab = (double *)xcalloc((size_t)ma * (size_t)mb, sizeof(double));
    double *ra = (double *)xcalloc((size_t)ma, sizeof(double)), *rb = (double *)xcalloc((size_t)mb, sizeof(double));
    for (int i = 0; i < n; i++) { tab[(size_t)a[i] * mb + b[i]] += 1; ra[a[i]] += 1; rb[b[i]] += 1; }
    double sij = 0, si = 0, sj = 0;
    for (size_t q = 0; q < (size_t)ma * mb; q++) sij += tab[q] * (tab[q] - 1) / 2;
    for (int q = 0; q < ma; q++) si += ra[q] * (ra[q] - 1) / 2;
    for (int q = 0; q < mb; q++) sj += rb[q] * (rb[q] - 1) / 2;
    double tot = (double)n * (n - 1) / 2;
    double A = sij, B = si - sij, Cc = sj - sij, D = tot - A - B - Cc;
    out5[0] = (A + D) / tot;                                   /* Rand */
    double e = si * sj / tot;
    out5[1] = (sij - e) / (0.5 * (si + sj) - e);               /* Hubert-Arabie */
    {   /* Morey-Agresti: expected sum n_ij^2 = sum n_i.^2 * sum n_.j^2 / n^2 */
        double sa2 = 0, sb2 = 0, nn = (double)n;
        for (int q = 0; q < ma; q++) sa2 += ra[q] * ra[q];
        for (int q = 0; q < mb; q++) sb2 += rb[q] * rb[q];
        double erand = (tot + sa2 * sb2 / (nn * nn) - 0.5 * (sa2 + sb2)) / tot;
        out5[2] = (out5[0] - erand) / (1 - erand);
    }
    out5[3] = A / sqrt((A + B) * (A + Cc));                    /* Fowlkes-Mallows */
    out5[4] = A / (A + B + Cc);                                /* Jaccard */
    free(tab); free(ra); free(rb);
}

/* ------------------------------------------------------------------------- */
/* Synthetic generator used by bench/tests (NOT from the reference; see        */
/* DESIGN.md "Synthetic inputs").  Counter-based: value = f(seed, gene, cell). */
/* Mirrors sharp_amd/csrc/synth.hpp bit-for-bit (integer-only value path).     */
/* ------------------------------------------------------------------------- */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
/* cluster of a cell: geometric-ish weights over G clusters */
int oracle_synth_cluster(uint32_t seed, uint32_t cell, int G) {
    uint32_t h = mix32(cell * 0x9e3779b9u + seed * 0x85ebca6bu + 0x1234567u);
    /* weights w_c ~ 0.82^c, cumulative in 16.16 fixed point, computed with integers */
    uint64_t w[64]; uint64_t tot = 0, cur = 1u << 20;
    for (int c = 0; c < G; c++) { w[c] = cur; tot += cur; cur = cur * 82 / 100; }
    uint64_t u = ((uint64_t)h * tot) >> 32, acc = 0;
    for (int c = 0; c < G; c++) { acc += w[c]; if (u < acc) return c; }
    return G - 1;
}
float oracle_synth_value(uint32_t seed, uint32_t gene, uint32_t cell, int G, int nmark) {
    int cl = oracle_synth_cluster(seed, cell, G);
    uint32_t hg = mix32(gene * 0x27d4eb2fu + seed);
    uint32_t h = mix32(mix32(gene * 0x9e3779b1u + seed) ^ (cell * 0x85ebca77u + 0xc2b2ae3du));
    uint32_t u = h >> 8;                               /* 24-bit uniform */
    int marker = ((int)(gene / (uint32_t)nmark) == cl) && (gene < (uint32_t)(G * nmark));
    uint32_t lvl = hg & 3u;                            /* per-gene base level */
    /* zero-probability thresholds (out of 2^24) */
    static const uint32_t z_base[4] = {16106127u, 15770583u, 15435038u, 14763950u}; /* .96 .94 .92 .88 */
    static const uint32_t z_mark[4] = {3355443u, 2516582u, 2013266u, 1677722u};     /* .20 .15 .12 .10 */
    uint32_t z = marker ? z_mark[lvl] : z_base[lvl];
    if (u < z) return 0.0f;
    /* non-zero: geometric-like count from the remaining bits */
    uint32_t r = mix32(h ^ 0x68bc21ebu);
    int cnt = 1;
    uint32_t thr = marker ? 0xD0000000u : 0x50000000u; /* continue-probability .8125 / .3125 */
    while (r < thr && cnt < 64) { cnt++; r = mix32(r + 0x9e3779b9u); }
    return (float)cnt;
}
void oracle_synth_fill(uint32_t seed, int m, int cell0, int ncell, int G, int nmark, double *X) {
    for (int j = 0; j < ncell; j++)
        for (int g = 0; g < m; g++)
            X[(size_t)j * m + g] = (double)oracle_synth_value(seed, (uint32_t)g, (uint32_t)(cell0 + j), G, nmark);
}
