/*
 * tests/rmock/Rinternals.h -- NOT R.  Declarations of the part of R's C API that shim/tpg_rshim.c uses, written from
 * R's documentation ("Writing R Extensions", section 5 and 6) so that the shim can be compiled with -Wall -Werror and
 * driven by tests in an image without R.  The signatures are R's; the implementation behind them (rmock.c) is a
 * minimal stand-in: vectors, attributes, environments with plain bindings, Rf_error as a longjmp.  A compile against
 * these headers is a syntax and signature guard, not a test against R.
 */
#ifndef TPG_RMOCK_RINTERNALS_H
#define TPG_RMOCK_RINTERNALS_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rmock_sexprec* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int R_len_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE } Rboolean;

#define NILSXP 0
#define SYMSXP 1
#define ENVSXP 4
#define CHARSXP 9
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19

extern SEXP R_NilValue, R_UnboundValue, R_NamesSymbol, R_DimSymbol, R_DimNamesSymbol;

int TYPEOF(SEXP x);
R_xlen_t XLENGTH(SEXP x);
R_len_t Rf_length(SEXP x);
double* REAL(SEXP x);
int* INTEGER(SEXP x);
int* LOGICAL(SEXP x);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
const char* CHAR(SEXP x);
const char* R_ExpandFileName(const char* s);
char* R_alloc(size_t n, int size); /* R_ext/Memory.h */

SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_mkChar(const char* s);
SEXP Rf_install(const char* name);
SEXP Rf_eval(SEXP expr, SEXP env);
SEXP Rf_coerceVector(SEXP x, SEXPTYPE type);
int Rf_asInteger(SEXP x);
int Rf_asLogical(SEXP x);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val);
SEXP Rf_getAttrib(SEXP x, SEXP name);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)

#if defined(__GNUC__)
void Rf_error(const char* fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
#else
void Rf_error(const char* fmt, ...);
#endif

#ifdef __cplusplus
}
#endif
#endif
