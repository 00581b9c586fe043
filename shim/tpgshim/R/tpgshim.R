# tpgshim: R side.  The native wrappers of tidypopgen (R/RcppExports.R:4-91 there) are one-line functions of the form
#   increment_ibs_counts <- function(k, k2, genotype0, genotype1, genotype2, BM, rowInd, colInd)
#     invisible(.Call(`_tidypopgen_increment_ibs_counts`, k, k2, genotype0, genotype1, genotype2, BM, rowInd, colInd))
# tpg_enable() replaces them, inside tidypopgen's namespace, by the same one-liners bound to THIS package's symbols of the
# same names; tpg_disable() puts the originals back.  Nothing else of tidypopgen changes.

.tpg_routines <- c(
  alt_freq_dip_pseudo_cpp = 6L, fbm256_prod_and_rowSumsSq = 6L, grouped_alt_freq_dip_pseudo_cpp = 8L,
  grouped_missingness_cpp = 6L, grouped_summaries_dip_pseudo_cpp = 7L, gt_grouped_pi_diploid = 6L, gt_ind_hetero = 4L,
  gt_pi_diploid = 4L, pairwise_fst_hudson_loop = 6L, pairwise_fst_nei87_loop = 7L, pairwise_fst_wc84_loop = 6L,
  increment_as_counts = 7L, increment_ibs_counts = 8L, increment_king_numerator = 9L
)
.tpg_saved <- new.env()

tpg_enable <- function() {
  ns <- asNamespace("tidypopgen")
  for (name in names(.tpg_routines)) {
    original <- get(name, envir = ns)
    if (is.null(.tpg_saved[[name]])) assign(name, original, envir = .tpg_saved)
    sym <- getNativeSymbolInfo(paste0("_tidypopgen_", name), PACKAGE = "tpgshim")
    stopifnot(length(formals(original)) == .tpg_routines[[name]])
    replacement <- original
    body(replacement) <- bquote(.Call(.(sym), ..(lapply(names(formals(original)), as.name))), splice = TRUE)
    unlockBinding(name, ns)
    assign(name, replacement, envir = ns)
    lockBinding(name, ns)
  }
  invisible(TRUE)
}

tpg_disable <- function() {
  ns <- asNamespace("tidypopgen")
  for (name in ls(.tpg_saved)) {
    unlockBinding(name, ns)
    assign(name, get(name, envir = .tpg_saved), envir = ns)
    lockBinding(name, ns)
  }
  invisible(TRUE)
}

# only needed under TPG_RSHIM_DEFERRED=1 (then: after the block loop of snp_ibs / snp_king / snp_allele_sharing)
tpg_flush <- function() invisible(.Call(`_tidypopgen_tpg_flush`))
tpg_release <- function() invisible(.Call(`_tidypopgen_tpg_release`))
# only needed under TPG_RSHIM_CACHE=1 (an HBM copy of the whole FBM kept between calls): after anything that writes to X
tpg_invalidate <- function(X) invisible(.Call(`_tidypopgen_tpg_invalidate`, X))

# whole analyses on every GPU of the node (TPG_DEVICES); X is the FBM.code256 of a gen_tibble (attr(x$genotypes, "fbm"))
# which: the matrices wanted; only the cross-products they need are computed (GRM alone: 2 of 5, KING + GRM: 4 of 5)
tpg_snp_pairwise <- function(X, ind.row = bigstatsr::rows_along(X), ind.col = bigstatsr::cols_along(X),
                             adjusted_counts = FALSE, which = c("ibs", "king", "allele_sharing", "grm")) {
  which <- match.arg(which, several.ok = TRUE)
  mask <- sum(c(ibs = 1L, king = 2L, allele_sharing = 4L, grm = 8L)[unique(which)])
  .Call(`_tidypopgen_tpg_snp_pairwise`, X, as.integer(ind.row), as.integer(ind.col), adjusted_counts, as.integer(mask))
}
tpg_grouped_alt_freq <- function(X, ind.row, ind.col, group_ids0 = NULL, n_groups = 0L, ploidy, as_counts = FALSE) {
  .Call(`_tidypopgen_tpg_grouped_alt_freq`, X, as.integer(ind.row), as.integer(ind.col), group_ids0, as.integer(n_groups),
        as.numeric(ploidy), as_counts)
}
tpg_pairwise_pop_fst <- function(X, ind.row, ind.col, group_ids0, n_groups, ploidy,
                                 method = c("Hudson", "Nei87", "WC84"), pairwise_combn = utils::combn(n_groups, 2),
                                 by_locus = FALSE, return_num_dem = FALSE) {
  method <- match(match.arg(method), c("Hudson", "Nei87", "WC84")) - 1L
  .Call(`_tidypopgen_tpg_pairwise_pop_fst`, X, as.integer(ind.row), as.integer(ind.col), as.integer(group_ids0),
        as.integer(n_groups), as.numeric(ploidy), method, pairwise_combn, by_locus, return_num_dem)
}
tpg_pca_partial_svd <- function(X, ind.row = bigstatsr::rows_along(X), ind.col = bigstatsr::cols_along(X), k = 10L) {
  .Call(`_tidypopgen_tpg_pca_partial_svd`, X, as.integer(ind.row), as.integer(ind.col), as.integer(k))
}
