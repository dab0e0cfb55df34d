/* tests/rmock/Rinternals.h -- TEST INFRASTRUCTURE, not R.
 *
 * A stand-in for the handful of R C-API entry points r/sharp_glue.c (this repository's own .Call shim) uses, so that the shim can be
 * compiled and EXECUTED in an image that has no R: tests build r/sharp_glue.c + tests/rmock/rmock.c into one shared object, make SEXPs
 * with the rmock_* helpers below through ctypes and call the R_sharp_* entry points on them.  The names and argument lists are those of
 * R's public API ("Writing R Extensions", section 5); everything behind them is a few dozen lines of malloc'ed vectors (rmock.c).  What
 * this proves: the shim parses, links against libsharp_hip.so, unpacks its arguments and packs its results the way its comments say.
 * What it does not prove: anything about real R (garbage collection, ALTREP, long-vector rules). */
#ifndef RMOCK_RINTERNALS_H
#define RMOCK_RINTERNALS_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef ptrdiff_t R_xlen_t;
typedef struct rmock_sexp *SEXP;
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct rmock_dll DllInfo;
typedef enum { FALSE = 0, TRUE } Rboolean;

enum { NILSXP = 0, LGLSXP = 10, INTSXP = 13, REALSXP = 14, STRSXP = 16, VECSXP = 19, CHARSXP = 9 };

extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol;

SEXP Rf_allocVector(unsigned type, R_xlen_t n);
SEXP Rf_allocMatrix(unsigned type, int nrow, int ncol);
SEXP Rf_mkNamed(unsigned type, const char **names);
SEXP Rf_ScalarInteger(int v);
SEXP Rf_getAttrib(SEXP x, SEXP name);
int Rf_asInteger(SEXP x);
int Rf_asLogical(SEXP x);
double Rf_asReal(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);
int Rf_isReal(SEXP x);
int Rf_isInteger(SEXP x);
int Rf_isNewList(SEXP x);
double *REAL(SEXP x);
int *INTEGER(SEXP x);
int LENGTH(SEXP x);
R_xlen_t XLENGTH(SEXP x);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
const char *CHAR(SEXP x);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
char *R_alloc(size_t n, int size);
void Rf_error(const char *fmt, ...) __attribute__((noreturn));
void Rf_warning(const char *fmt, ...);
int R_registerRoutines(DllInfo *, const void *, const R_CallMethodDef *, const void *, const void *);
Rboolean R_useDynamicSymbols(DllInfo *, Rboolean);

#define allocVector Rf_allocVector
#define allocMatrix Rf_allocMatrix
#define mkNamed Rf_mkNamed
#define ScalarInteger Rf_ScalarInteger
#define getAttrib Rf_getAttrib
#define asInteger Rf_asInteger
#define asLogical Rf_asLogical
#define asReal Rf_asReal
#define nrows Rf_nrows
#define ncols Rf_ncols
#define isReal Rf_isReal
#define isInteger Rf_isInteger
#define isNewList Rf_isNewList
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
#define error Rf_error
#define warning Rf_warning

/* ---- helpers for the test driver (not part of R's API) ---- */
SEXP rmock_real_matrix(const double *data, int nrow, int ncol);      /* copies */
SEXP rmock_real_vector(const double *data, R_xlen_t n);
SEXP rmock_int_vector(const int *data, R_xlen_t n);
SEXP rmock_logical(int v);
SEXP rmock_list(int n);
void rmock_list_set(SEXP lst, int i, SEXP v, const char *name);      /* name may be NULL */
SEXP rmock_list_get(SEXP lst, const char *name);
/* call f(a0 .. a[nargs-1]) with error() caught: NULL + rmock_last_error() if the callee raised */
SEXP rmock_call(DL_FUNC f, int nargs, SEXP *args);
const char *rmock_last_error(void);
const char *rmock_warnings(void);
int rmock_protect_depth(void);
int rmock_registered(const char *name);                               /* number of arguments the shim registered for `name`, -1 if none */
void rmock_reset(void);                                               /* frees every SEXP made so far */

#ifdef __cplusplus
}
#endif
#endif
