/* tests/rmock/R.h -- NOT R: see Rinternals.h in this directory. */
#ifndef TPG_RMOCK_R_H
#define TPG_RMOCK_R_H
#include <stdlib.h>
#include "Rinternals.h"
#endif
