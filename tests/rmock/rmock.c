/*
 * tests/rmock/rmock.c -- NOT R.  A minimal runtime behind tests/rmock/Rinternals.h, enough to DRIVE shim/tpg_rshim.c
 * from a test the way R drives it: vectors with attributes, environments whose bindings Rf_eval(symbol, env) looks up
 * (how the shim reads the fields of a bigstatsr FBM reference-class object), Rf_error as a longjmp back into
 * rmock_call().  No garbage collector (objects live until rmock_reset), no promises, no active bindings.
 * Test helpers (rmock_*) are what tests/test_gpu_rshim.py calls through ctypes.
 */
#define _POSIX_C_SOURCE 200809L
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "R.h"
#include "R_ext/Rdynload.h"

struct rmock_sexprec {
  SEXPTYPE type;
  R_xlen_t len;
  void* data;                 /* double / int / SEXP array, char* for CHARSXP and SYMSXP */
  struct rmock_sexprec* attr_name[8];
  struct rmock_sexprec* attr_val[8];
  int nattr;
  struct rmock_sexprec* next; /* allocation list */
};

static struct rmock_sexprec nil_rec = {NILSXP, 0, NULL, {0}, {0}, 0, NULL}, unbound_rec = {SYMSXP, 0, (void*)"<unbound>", {0}, {0}, 0, NULL};
static struct rmock_sexprec names_rec = {SYMSXP, 0, (void*)"names", {0}, {0}, 0, NULL}, dim_rec = {SYMSXP, 0, (void*)"dim", {0}, {0}, 0, NULL},
                            dimnames_rec = {SYMSXP, 0, (void*)"dimnames", {0}, {0}, 0, NULL};
SEXP R_NilValue = &nil_rec, R_UnboundValue = &unbound_rec, R_NamesSymbol = &names_rec, R_DimSymbol = &dim_rec,
     R_DimNamesSymbol = &dimnames_rec;

static SEXP g_all = NULL;
static jmp_buf g_jmp;
static int g_jmp_set = 0;
static char g_err[1024];

static SEXP new_rec(SEXPTYPE type, R_xlen_t len, size_t elt) {
  SEXP s = (SEXP)calloc(1, sizeof(struct rmock_sexprec));
  if (!s) abort();
  s->type = type;
  s->len = len;
  s->data = len > 0 && elt ? calloc((size_t)len, elt) : NULL;
  s->next = g_all;
  g_all = s;
  return s;
}

void rmock_reset(void) {
  while (g_all) {
    SEXP n = g_all->next;
    free(g_all->data);
    free(g_all);
    g_all = n;
  }
}

void Rf_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  if (g_jmp_set) longjmp(g_jmp, 1);
  fprintf(stderr, "Rf_error outside rmock_call: %s\n", g_err);
  abort();
}

/* R_alloc: transient storage R reclaims when .Call returns; here it lives until rmock_reset like every object */
char* R_alloc(size_t n, int size) { return (char*)new_rec(CHARSXP, (R_xlen_t)(n ? n : 1), (size_t)size)->data; }

int TYPEOF(SEXP x) { return (int)x->type; }
R_xlen_t XLENGTH(SEXP x) { return x->len; }
R_len_t Rf_length(SEXP x) { return (R_len_t)x->len; }
double* REAL(SEXP x) { if (x->type != REALSXP) Rf_error("REAL() on a non-double"); return (double*)x->data; }
int* INTEGER(SEXP x) { if (x->type != INTSXP && x->type != LGLSXP) Rf_error("INTEGER() on a non-integer"); return (int*)x->data; }
int* LOGICAL(SEXP x) { if (x->type != LGLSXP) Rf_error("LOGICAL() on a non-logical"); return (int*)x->data; }
SEXP STRING_ELT(SEXP x, R_xlen_t i) { if (x->type != STRSXP || i >= x->len) Rf_error("STRING_ELT"); return ((SEXP*)x->data)[i]; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { if (x->type != VECSXP || i >= x->len) Rf_error("VECTOR_ELT"); return ((SEXP*)x->data)[i]; }
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) { if (x->type != STRSXP || i >= x->len) Rf_error("SET_STRING_ELT"); ((SEXP*)x->data)[i] = v; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { if (x->type != VECSXP || i >= x->len) Rf_error("SET_VECTOR_ELT"); ((SEXP*)x->data)[i] = v; return v; }
const char* CHAR(SEXP x) { return (const char*)x->data; }
const char* R_ExpandFileName(const char* s) { return s; }

SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n) {
  switch (type) {
    case REALSXP: return new_rec(type, n, sizeof(double));
    case INTSXP: case LGLSXP: return new_rec(type, n, sizeof(int));
    case STRSXP: case VECSXP: { SEXP s = new_rec(type, n, sizeof(SEXP)); for (R_xlen_t i = 0; i < n; i++) ((SEXP*)s->data)[i] = R_NilValue; return s; }
    default: Rf_error("rmock: allocVector of type %u", type);
  }
}

SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol) {
  SEXP s = Rf_allocVector(type, (R_xlen_t)nrow * ncol);
  SEXP d = Rf_allocVector(INTSXP, 2);
  INTEGER(d)[0] = nrow;
  INTEGER(d)[1] = ncol;
  Rf_setAttrib(s, R_DimSymbol, d);
  return s;
}

SEXP Rf_mkChar(const char* str) {
  SEXP s = new_rec(CHARSXP, (R_xlen_t)strlen(str), 0);
  s->data = strdup(str);
  return s;
}

SEXP Rf_install(const char* name) {
  if (!strcmp(name, "names")) return R_NamesSymbol;
  if (!strcmp(name, "dim")) return R_DimSymbol;
  if (!strcmp(name, "dimnames")) return R_DimNamesSymbol;
  SEXP s = new_rec(SYMSXP, 0, 0);
  s->data = strdup(name);
  return s;
}

static int same_sym(SEXP a, SEXP b) { return a == b || !strcmp((const char*)a->data, (const char*)b->data); }

SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val) {
  for (int k = 0; k < x->nattr; k++)
    if (same_sym(x->attr_name[k], name)) { x->attr_val[k] = val; return val; }
  if (x->nattr >= 8) Rf_error("rmock: too many attributes");
  x->attr_name[x->nattr] = name;
  x->attr_val[x->nattr++] = val;
  return val;
}

SEXP Rf_getAttrib(SEXP x, SEXP name) {
  for (int k = 0; k < x->nattr; k++)
    if (same_sym(x->attr_name[k], name)) return x->attr_val[k];
  return R_NilValue;
}

/* environments keep their bindings as attributes (name symbol -> value) */
SEXP Rf_eval(SEXP expr, SEXP env) {
  if (expr->type != SYMSXP) return expr;
  if (env->type != ENVSXP) Rf_error("rmock: eval in a non-environment");
  for (int k = 0; k < env->nattr; k++)
    if (same_sym(env->attr_name[k], expr)) return env->attr_val[k];
  Rf_error("object '%s' not found", (const char*)expr->data);
}

SEXP Rf_coerceVector(SEXP x, SEXPTYPE type) {
  if (x->type == type) return x;
  SEXP out = Rf_allocVector(type, x->len);
  for (R_xlen_t i = 0; i < x->len; i++) {
    double v;
    if (x->type == REALSXP) v = ((double*)x->data)[i];
    else if (x->type == INTSXP || x->type == LGLSXP) v = (double)((int*)x->data)[i];
    else Rf_error("rmock: coerceVector from type %u", x->type);
    if (type == REALSXP) ((double*)out->data)[i] = v;
    else if (type == INTSXP || type == LGLSXP) ((int*)out->data)[i] = (int)v;
    else Rf_error("rmock: coerceVector to type %u", type);
  }
  for (int k = 0; k < x->nattr; k++) Rf_setAttrib(out, x->attr_name[k], x->attr_val[k]);
  return out;
}

int Rf_asInteger(SEXP x) {
  if (x->len < 1) Rf_error("asInteger of an empty vector");
  return x->type == REALSXP ? (int)((double*)x->data)[0] : ((int*)x->data)[0];
}
int Rf_asLogical(SEXP x) { return Rf_asInteger(x) != 0; }
SEXP Rf_protect(SEXP x) { return x; }
void Rf_unprotect(int n) { (void)n; }

int R_registerRoutines(DllInfo* info, const R_CMethodDef* const c, const R_CallMethodDef* const call, const R_FortranMethodDef* const f,
                       const R_ExternalMethodDef* const e) {
  (void)info; (void)c; (void)call; (void)f; (void)e;
  return 1;
}
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value) { (void)info; return value; }

/* ---- helpers for the tests ------------------------------------------------------------------------------------ */
SEXP rmock_new_env(void) { return new_rec(ENVSXP, 0, 0); }
void rmock_env_set(SEXP env, const char* name, SEXP val) { Rf_setAttrib(env, Rf_install(name), val); }
SEXP rmock_real(const double* v, R_xlen_t n) { SEXP s = Rf_allocVector(REALSXP, n); if (n) memcpy(s->data, v, sizeof(double) * (size_t)n); return s; }
SEXP rmock_int(const int* v, R_xlen_t n) { SEXP s = Rf_allocVector(INTSXP, n); if (n) memcpy(s->data, v, sizeof(int) * (size_t)n); return s; }
SEXP rmock_lgl(int v) { SEXP s = Rf_allocVector(LGLSXP, 1); ((int*)s->data)[0] = v; return s; }
SEXP rmock_str(const char* v) { SEXP s = Rf_allocVector(STRSXP, 1); SET_STRING_ELT(s, 0, Rf_mkChar(v)); return s; }
SEXP rmock_real_matrix(const double* v, int nrow, int ncol) { SEXP s = Rf_allocMatrix(REALSXP, nrow, ncol); if (v) memcpy(s->data, v, sizeof(double) * (size_t)nrow * (size_t)ncol); return s; }
SEXP rmock_nil(void) { return R_NilValue; }
void* rmock_data(SEXP x) { return x->data; }
const char* rmock_last_error(void) { return g_err; }

/* call a .Call entry point with up to 10 arguments; NULL (and rmock_last_error) when it raised an R error */
SEXP rmock_call(DL_FUNC fn, int nargs, SEXP* a) {
  typedef SEXP (*F0)(void);
  typedef SEXP (*F1)(SEXP);
  typedef SEXP (*F4)(SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*F5)(SEXP, SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*F6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*F7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*F8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*F9)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*F10)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
  g_err[0] = 0;
  if (setjmp(g_jmp)) { g_jmp_set = 0; return NULL; }
  g_jmp_set = 1;
  SEXP out = NULL;
  switch (nargs) {
    case 0: out = ((F0)fn)(); break;
    case 1: out = ((F1)fn)(a[0]); break;
    case 4: out = ((F4)fn)(a[0], a[1], a[2], a[3]); break;
    case 5: out = ((F5)fn)(a[0], a[1], a[2], a[3], a[4]); break;
    case 6: out = ((F6)fn)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
    case 7: out = ((F7)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
    case 8: out = ((F8)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
    case 9: out = ((F9)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
    case 10: out = ((F10)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); break;
    default: snprintf(g_err, sizeof(g_err), "rmock_call: %d arguments not supported", nargs); break;
  }
  g_jmp_set = 0;
  return out;
}
