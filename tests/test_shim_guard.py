"""CPU: shim/tpg_rshim.c cannot be compiled against R here (R is not installed), so it is guarded instead:
compiled with -Wall -Wextra -Werror against include/tpg.h and tests/rmock/ (stand-in declarations of the R API it uses,
written from R's documentation -- a syntax and signature guard, NOT R), in both build modes; its registration table is
compared, name by name and arity by arity, with the reference's (src/RcppExports.cpp:348-371); the package skeleton
shim/tpgshim is checked for the pieces R CMD INSTALL needs."""
import os
import re

from tests import rmock

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the hot-path rows of the reference's CallEntries (src/RcppExports.cpp:348-371): name -> arity
REFERENCE_ARITY = {
    "_tidypopgen_alt_freq_dip_pseudo_cpp": 6, "_tidypopgen_fbm256_prod_and_rowSumsSq": 6,
    "_tidypopgen_grouped_alt_freq_dip_pseudo_cpp": 8, "_tidypopgen_grouped_missingness_cpp": 6,
    "_tidypopgen_grouped_summaries_dip_pseudo_cpp": 7, "_tidypopgen_gt_grouped_pi_diploid": 6,
    "_tidypopgen_gt_ind_hetero": 4, "_tidypopgen_gt_pi_diploid": 4, "_tidypopgen_pairwise_fst_hudson_loop": 6,
    "_tidypopgen_pairwise_fst_nei87_loop": 7, "_tidypopgen_pairwise_fst_wc84_loop": 6,
    "_tidypopgen_increment_as_counts": 7, "_tidypopgen_increment_ibs_counts": 8,
    "_tidypopgen_increment_king_numerator": 9,
}


def test_shim_compiles_against_the_c_abi_and_the_r_api_guard():
    for extra in ((), ("-DTPG_RSHIM_STANDALONE",)):
        r = rmock.compile_only(extra)
        assert r.returncode == 0, r.stderr[-4000:]


def test_shim_package_unit_compiles():
    import subprocess

    r = subprocess.run(["gcc", *rmock.CFLAGS, "-fsyntax-only", "-I" + rmock.HERE, "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "shim"), "-DTPG_RSHIM_STANDALONE",
                        os.path.join(ROOT, "shim", "tpgshim", "src", "tpgshim.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_registration_table_matches_the_reference(tmp_path):
    lib = rmock.build(tmp_path)  # links against libtpg_hip.so; loading it needs no GPU
    ent = rmock.entries(lib)
    for name, arity in REFERENCE_ARITY.items():
        assert name in ent, name
        assert ent[name][1] == arity, (name, ent[name][1], arity)
        assert ent[name][0], name
    extras = set(ent) - set(REFERENCE_ARITY)
    assert all(e.startswith("_tidypopgen_tpg_") for e in extras), extras  # additions carry their own prefix


def test_safe_by_default():
    src = open(os.path.join(ROOT, "shim", "tpg_rshim.c")).read()
    assert "TPG_RSHIM_EAGER" not in src  # the old opt-in to correctness is gone
    assert "TPG_RSHIM_DEFERRED" in src   # deferral is the opt-in now
    hdr = open(os.path.join(ROOT, "include", "tpg.h")).read()
    assert "tpg_increment_defer" in hdr


def test_package_skeleton():
    pkg = os.path.join(ROOT, "shim", "tpgshim")
    desc = open(os.path.join(pkg, "DESCRIPTION")).read()
    for field in ("Package: tpgshim", "Version:", "License:", "NeedsCompilation: yes"):
        assert field in desc
    ns = open(os.path.join(pkg, "NAMESPACE")).read()
    assert "useDynLib(tpgshim, .registration = TRUE)" in ns
    mk = open(os.path.join(pkg, "src", "Makevars")).read()
    assert "-ltpg_hip" in mk and "-DTPG_RSHIM_STANDALONE" in mk
    rsrc = open(os.path.join(pkg, "R", "tpgshim.R")).read()
    exported = set(re.findall(r"export\((\w+)\)", ns))
    for fn in exported:
        assert re.search(rf"^{fn} <- function", rsrc, re.M), fn
    # every routine tpg_enable() rebinds is a row of the shim's table with the arity the R side asserts
    table = rsrc[rsrc.index(".tpg_routines <- c("):rsrc.index(".tpg_saved")]
    for name, arity in re.findall(r"(\w+) = (\d+)L", table):
        assert REFERENCE_ARITY["_tidypopgen_" + name] == int(arity), name
