/* tests/rmock/rmock.c -- the dozen R C-API functions r/sharp_glue.c uses, over malloc'ed vectors.  TEST INFRASTRUCTURE (see Rinternals.h). */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "Rinternals.h"

struct rmock_sexp {
    unsigned type;
    R_xlen_t n;
    void *data;          /* double[] / int[] / SEXP[] / char[] */
    SEXP names, dim;     /* attributes */
    struct rmock_sexp *next;
};

static struct rmock_sexp nil_obj = {NILSXP, 0, NULL, NULL, NULL, NULL}, names_sym = {NILSXP, 0, NULL, NULL, NULL, NULL},
                         dim_sym = {NILSXP, 0, NULL, NULL, NULL, NULL};
SEXP R_NilValue = &nil_obj, R_NamesSymbol = &names_sym, R_DimSymbol = &dim_sym;

static struct rmock_sexp *heap = NULL;
static void **scratch = NULL;
static int n_scratch = 0, protect_depth = 0, in_call = 0;
static jmp_buf jb;
static char err_msg[2048], warn_msg[4096];
static const R_CallMethodDef *registered = NULL;

static SEXP new_sexp(unsigned type, R_xlen_t n, size_t elt) {
    SEXP s = (SEXP)calloc(1, sizeof(*s));
    s->type = type; s->n = n; s->names = s->dim = R_NilValue;
    s->data = calloc((size_t)(n > 0 ? n : 1), elt);
    if (!s->data) { fprintf(stderr, "rmock: out of memory\n"); abort(); }
    if (type == VECSXP || type == STRSXP) for (R_xlen_t i = 0; i < n; ++i) ((SEXP *)s->data)[i] = R_NilValue;
    s->next = heap; heap = s;
    return s;
}

SEXP Rf_allocVector(unsigned type, R_xlen_t n) {
    switch (type) {
    case REALSXP: return new_sexp(type, n, sizeof(double));
    case INTSXP: case LGLSXP: return new_sexp(type, n, sizeof(int));
    case VECSXP: case STRSXP: return new_sexp(type, n, sizeof(SEXP));
    default: Rf_error("rmock: allocVector of type %u", type);
    }
}
SEXP Rf_allocMatrix(unsigned type, int nrow, int ncol) {
    SEXP s = Rf_allocVector(type, (R_xlen_t)nrow * ncol);
    s->dim = Rf_allocVector(INTSXP, 2);
    INTEGER(s->dim)[0] = nrow; INTEGER(s->dim)[1] = ncol;
    return s;
}
static SEXP mkchar(const char *c) {
    SEXP s = new_sexp(CHARSXP, (R_xlen_t)strlen(c), 1);
    free(s->data); s->data = strdup(c);
    return s;
}
SEXP Rf_mkNamed(unsigned type, const char **names) {
    int n = 0;
    while (names[n][0]) ++n;
    SEXP s = Rf_allocVector(type, n);
    s->names = Rf_allocVector(STRSXP, n);
    for (int i = 0; i < n; ++i) ((SEXP *)s->names->data)[i] = mkchar(names[i]);
    return s;
}
SEXP Rf_ScalarInteger(int v) { SEXP s = Rf_allocVector(INTSXP, 1); INTEGER(s)[0] = v; return s; }
SEXP Rf_getAttrib(SEXP x, SEXP name) { return name == R_NamesSymbol ? x->names : name == R_DimSymbol ? x->dim : R_NilValue; }
int Rf_asInteger(SEXP x) { return x->type == REALSXP ? (int)((double *)x->data)[0] : ((int *)x->data)[0]; }
int Rf_asLogical(SEXP x) { return Rf_asInteger(x) != 0; }
double Rf_asReal(SEXP x) { return x->type == REALSXP ? ((double *)x->data)[0] : (double)((int *)x->data)[0]; }
int Rf_nrows(SEXP x) { return x->dim != R_NilValue ? INTEGER(x->dim)[0] : (int)x->n; }
int Rf_ncols(SEXP x) { return x->dim != R_NilValue ? INTEGER(x->dim)[1] : 1; }
int Rf_isReal(SEXP x) { return x->type == REALSXP; }
int Rf_isInteger(SEXP x) { return x->type == INTSXP; }
int Rf_isNewList(SEXP x) { return x->type == VECSXP || x == R_NilValue; }
double *REAL(SEXP x) { if (x->type != REALSXP) Rf_error("REAL() can only be applied to a 'numeric', not a type-%u object", x->type); return (double *)x->data; }
int *INTEGER(SEXP x) { if (x->type != INTSXP && x->type != LGLSXP) Rf_error("INTEGER() can only be applied to a 'integer', not a type-%u object", x->type); return (int *)x->data; }
int LENGTH(SEXP x) { return (int)x->n; }
R_xlen_t XLENGTH(SEXP x) { return x->n; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { if (x->type != VECSXP || i < 0 || i >= x->n) Rf_error("rmock: VECTOR_ELT out of range or not a list"); return ((SEXP *)x->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { if (x->type != VECSXP || i < 0 || i >= x->n) Rf_error("rmock: SET_VECTOR_ELT out of range or not a list"); ((SEXP *)x->data)[i] = v; return v; }
SEXP STRING_ELT(SEXP x, R_xlen_t i) { if (x->type != STRSXP || i < 0 || i >= x->n) Rf_error("rmock: STRING_ELT out of range"); return ((SEXP *)x->data)[i]; }
const char *CHAR(SEXP x) { return (const char *)x->data; }
SEXP Rf_protect(SEXP x) { ++protect_depth; return x; }
void Rf_unprotect(int n) { protect_depth -= n; if (protect_depth < 0) { fprintf(stderr, "rmock: unprotect(): stack imbalance\n"); abort(); } }
char *R_alloc(size_t n, int size) {
    scratch = (void **)realloc(scratch, sizeof(void *) * (size_t)(n_scratch + 1));
    return (char *)(scratch[n_scratch++] = calloc(n ? n : 1, (size_t)size));
}
void Rf_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(err_msg, sizeof err_msg, fmt, ap); va_end(ap);
    if (!in_call) { fprintf(stderr, "rmock: error outside rmock_call: %s\n", err_msg); abort(); }
    longjmp(jb, 1);
}
void Rf_warning(const char *fmt, ...) {
    size_t l = strlen(warn_msg);
    va_list ap; va_start(ap, fmt); vsnprintf(warn_msg + l, sizeof warn_msg - l - 2, fmt, ap); va_end(ap);
    strcat(warn_msg, "\n");
}
int R_registerRoutines(DllInfo *d, const void *c, const R_CallMethodDef *call, const void *f, const void *e) { (void)d; (void)c; (void)f; (void)e; registered = call; return 1; }
Rboolean R_useDynamicSymbols(DllInfo *d, Rboolean v) { (void)d; return v; }

/* ---- test-driver helpers ---- */
SEXP rmock_real_matrix(const double *data, int nrow, int ncol) { SEXP s = Rf_allocMatrix(REALSXP, nrow, ncol); memcpy(s->data, data, sizeof(double) * (size_t)nrow * (size_t)ncol); return s; }
SEXP rmock_real_vector(const double *data, R_xlen_t n) { SEXP s = Rf_allocVector(REALSXP, n); memcpy(s->data, data, sizeof(double) * (size_t)n); return s; }
SEXP rmock_int_vector(const int *data, R_xlen_t n) { SEXP s = Rf_allocVector(INTSXP, n); memcpy(s->data, data, sizeof(int) * (size_t)n); return s; }
SEXP rmock_logical(int v) { SEXP s = Rf_allocVector(LGLSXP, 1); ((int *)s->data)[0] = v; return s; }
SEXP rmock_list(int n) { return Rf_allocVector(VECSXP, n); }
void rmock_list_set(SEXP lst, int i, SEXP v, const char *name) {
    ((SEXP *)lst->data)[i] = v;
    if (name) {
        if (lst->names == R_NilValue) { lst->names = Rf_allocVector(STRSXP, lst->n); for (R_xlen_t q = 0; q < lst->n; ++q) ((SEXP *)lst->names->data)[q] = mkchar(""); }
        ((SEXP *)lst->names->data)[i] = mkchar(name);
    }
}
SEXP rmock_list_get(SEXP lst, const char *name) {
    if (lst->names == R_NilValue) return R_NilValue;
    for (R_xlen_t q = 0; q < lst->n; ++q) if (strcmp(CHAR(((SEXP *)lst->names->data)[q]), name) == 0) return ((SEXP *)lst->data)[q];
    return R_NilValue;
}
typedef SEXP (*f0)(void); typedef SEXP (*f1)(SEXP); typedef SEXP (*f4)(SEXP, SEXP, SEXP, SEXP); typedef SEXP (*f5)(SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*f6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*f7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP rmock_call(DL_FUNC f, int nargs, SEXP *a) {
    SEXP r = NULL;
    const int depth0 = protect_depth;
    err_msg[0] = 0;
    in_call = 1;
    if (setjmp(jb) == 0) {
        switch (nargs) {
        case 0: r = ((f0)f)(); break;
        case 1: r = ((f1)f)(a[0]); break;
        case 4: r = ((f4)f)(a[0], a[1], a[2], a[3]); break;
        case 5: r = ((f5)f)(a[0], a[1], a[2], a[3], a[4]); break;
        case 6: r = ((f6)f)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
        case 7: r = ((f7)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
        default: snprintf(err_msg, sizeof err_msg, "rmock_call: %d arguments not supported", nargs);
        }
    } else {
        protect_depth = depth0;      /* R unwinds the protection stack on error */
        r = NULL;
    }
    in_call = 0;
    for (int i = 0; i < n_scratch; ++i) free(scratch[i]);      /* R_alloc memory lives until the .Call returns */
    n_scratch = 0;
    return r;
}
const char *rmock_last_error(void) { return err_msg; }
const char *rmock_warnings(void) { return warn_msg; }
int rmock_protect_depth(void) { return protect_depth; }
int rmock_registered(const char *name) {
    if (!registered) return -1;
    for (const R_CallMethodDef *d = registered; d->name; ++d) if (strcmp(d->name, name) == 0) return d->numArgs;
    return -1;
}
void rmock_reset(void) {
    while (heap) { struct rmock_sexp *nx = heap->next; free(heap->data); free(heap); heap = nx; }
    warn_msg[0] = 0; err_msg[0] = 0; protect_depth = 0;
}
/* accessors for ctypes */
int rmock_type(SEXP x) { return (int)x->type; }
void *rmock_data(SEXP x) { return x->data; }
