/* tests/rmock/R_ext/Rdynload.h -- NOT R: the registration types of R_ext/Rdynload.h as documented in "Writing R
 * Extensions" 5.4, for the compile guard of shim/tpg_rshim.c. */
#ifndef TPG_RMOCK_RDYNLOAD_H
#define TPG_RMOCK_RDYNLOAD_H
#include "../Rinternals.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void* (*DL_FUNC)(void);
typedef struct {
  const char* name;
  DL_FUNC fun;
  int numArgs;
} R_CallMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct {
  const char* name;
  DL_FUNC fun;
  int numArgs;
  void* types;
} R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef struct rmock_dllinfo DllInfo;

int R_registerRoutines(DllInfo* info, const R_CMethodDef* const croutines, const R_CallMethodDef* const callRoutines,
                       const R_FortranMethodDef* const fortranRoutines, const R_ExternalMethodDef* const externalRoutines);
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value);

#ifdef __cplusplus
}
#endif
#endif
