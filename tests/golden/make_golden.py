#!/usr/bin/env python3
"""Collects the golden DATA fixtures of the reference's own tests for the hot
path into tests/golden/ (run once in the build container, where
/root/reference exists; the GPU box only sees the committed copies).

These are data files (inputs and expected outputs), not source:
  related/families.{bed,fam}      12 x 961 PLINK genotypes, 3.9 % missing
  related/test_plinkIBS.mibs      plink --distance square flat-missing ibs  (tests/testthat/test_snp_ibs.R:69-105)
  related/test_king.kin0          king -b ... --kinship                      (tests/testthat/test_snp_king.R:189-236)
  lobster/lobster.{bed,fam}       176 x 79
  fst_scikit-allel/*.txt          scikit-allel 1.3.13 Hudson / WC84 outputs   (tests/testthat/test_pairwise_pop_fst.R:55-343)
The reference (and therefore these data files) is licensed GPL (>= 3).
"""
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

FILES = {
    "inst/extdata/related/families.bed": "related/families.bed",
    "inst/extdata/related/families.fam": "related/families.fam",
    "inst/extdata/related/test_plinkIBS.mibs": "related/test_plinkIBS.mibs",
    "inst/extdata/related/test_king.kin0": "related/test_king.kin0",
    "inst/extdata/lobster/lobster.bed": "lobster/lobster.bed",
    "inst/extdata/lobster/lobster.fam": "lobster/lobster.fam",
}
for f in ("fst_hudson", "fst_hudson_monomorphic", "fst_hudson_per_loc", "fst_wc", "fst_wc_monomorphic",
          "fst_wc_per_loc"):
    FILES[f"tests/testthat/testdata/fst_scikit-allel/{f}.txt"] = f"fst_scikit-allel/{f}.txt"

if __name__ == "__main__":
    for src, dst in FILES.items():
        d = os.path.join(HERE, dst)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(os.path.join(REF, src), d)
        os.chmod(d, 0o644)
        print("copied", src, "->", dst)
