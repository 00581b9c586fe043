/* The package's only translation unit: the shim itself (shim/tpg_rshim.c, found through -I$(TPG_HOME)/shim), compiled
 * with TPG_RSHIM_STANDALONE so that R_init_tpgshim registers its .Call table. */
#include "tpg_rshim.c"
