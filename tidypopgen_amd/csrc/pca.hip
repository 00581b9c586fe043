// pca.hip -- PCA Gram / partial SVD / projections (placeholder until the kernels land).
#include "common.h"

#define TPG_NOT_YET(name) do { tpg_set_error(name ": not implemented yet"); return TPG_EUNSUPPORTED; } while (0)

extern "C" int tpg_pca_center_scale(tpg_ctx*, const tpg_view*, double*, double*) { TPG_NOT_YET("tpg_pca_center_scale"); }
extern "C" int tpg_pca_gram(tpg_ctx*, const tpg_view*, const double*, const double*, double*) { TPG_NOT_YET("tpg_pca_gram"); }
extern "C" int tpg_pca_partial_svd(tpg_ctx*, const tpg_view*, int, double*, double*, double*, double*, double*, double*) { TPG_NOT_YET("tpg_pca_partial_svd"); }
extern "C" int tpg_fbm256_prod_and_rowSumsSq(tpg_ctx*, const tpg_view*, const double*, const double*, const double*, int, double*, double*) { TPG_NOT_YET("tpg_fbm256_prod_and_rowSumsSq"); }
extern "C" int tpg_square_frobenius(tpg_ctx*, const tpg_view*, const double*, const double*, double*) { TPG_NOT_YET("tpg_square_frobenius"); }
