"""Run by tests/test_host_sanitizers.py in a child process whose LD_PRELOAD is the ASan runtime: drives the sanitizer build
of shim/tpg_rshim.c + tests/rmock/rmock.c + tests/host/tpg_stub.c (a host stand-in for libtpg_hip.so) the way the R drivers
drive the shim.  usage: drive_shim_san.py <lib.so> <tmpdir>"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import rmock  # noqa: E402

so, tmp = sys.argv[1], sys.argv[2]
lib = rmock.bind(C.CDLL(so))
r = rmock.Session(lib)
counter = lambda name: C.c_int.in_dll(lib, name).value  # noqa: E731

rng = np.random.default_rng(7)
n_all, m_all = 37, 900
fbm = rng.integers(0, 4, size=(n_all, m_all)).astype(np.uint8)
bk = os.path.join(tmp, "geno.bk")
fbm.T.tofile(bk)
code = np.full(256, np.nan)
code[:3] = [0, 1, 2]
rows = (rng.permutation(n_all)[:29] + 1).astype(np.int32)
ploidy = np.full(len(rows), 2.0)


def expect(cols):
    sub = fbm[np.ix_(rows - 1, cols - 1)].astype(float)
    sub[sub == 3] = np.nan
    return np.stack([np.nansum(sub, axis=0), 2.0 * np.sum(~np.isnan(sub), axis=0)], axis=1)


def alt_freq(BM, cols):
    out = r.call("alt_freq_dip_pseudo_cpp", BM, r.int(rows), r.int(cols), r.real(ploidy), r.int([1]), lib.rmock_lgl(1))
    return r.as_numpy(out, (len(cols), 2))


for cache in (False, True):
    os.environ.pop("TPG_RSHIM_CACHE", None)
    if cache:
        os.environ["TPG_RSHIM_CACHE"] = "1"
    BM = r.fbm(bk, n_all, m_all, code)
    up0 = counter("g_stub_uploads")
    # the big_apply blocks of the R driver (contiguous), a reversed block, a scattered colInd (gathered upload), one column
    for cols in (np.arange(1, 301), np.arange(301, 901), np.arange(600, 100, -1), np.arange(1, 901, 13), np.array([900])):
        cols = cols.astype(np.int32)
        assert np.array_equal(alt_freq(BM, cols), expect(cols)), (cache, cols[:4])
    if cache:
        assert counter("g_stub_uploads") == up0 + 1, "the cached FBM was uploaded more than once"
        r.call("tpg_invalidate", BM)
        assert counter("g_stub_fbm_alive") == 0
        assert np.array_equal(alt_freq(BM, np.arange(1, 11, dtype=np.int32)), expect(np.arange(1, 11)))
        assert counter("g_stub_uploads") == up0 + 2
    else:
        assert counter("g_stub_uploads") == up0 + 5 and counter("g_stub_fbm_alive") == 0, "a per-call upload outlived its call"
    assert counter("g_stub_view_alive") == 0
    # an out-of-range colInd is an R error, and nothing leaks on the way out
    try:
        alt_freq(BM, np.array([m_all + 1], dtype=np.int32))
        raise SystemExit("out-of-range colInd accepted")
    except RuntimeError:
        pass
    assert counter("g_stub_view_alive") == 0
    r.call("tpg_release")
    assert counter("g_stub_fbm_alive") == 0

# the block loops of three analyses one after the other on file-backed accumulators: the table of mapped files grows,
# entries are forgotten (swap-compaction) while others are in use, pointers handed out stay valid
os.environ.pop("TPG_RSHIM_CACHE", None)
BM = r.fbm(bk, n_all, m_all, code)
n = len(rows)
cols = np.arange(1, m_all + 1, dtype=np.int32)
for which in ("ibs", "king", "as", "ibs"):
    files = []
    for nm in ("k", "k2"):
        f = os.path.join(tmp, f"{which}_{nm}_{len(files)}_{np.random.randint(1 << 30)}.bk")
        np.zeros(n * n).tofile(f)
        files.append(f)
    K, K2 = r.fbm(files[0], n, n), r.fbm(files[1], n, n)
    lo, up = np.array([1, 241, 481, 722]), np.array([240, 480, 721, 900])
    rmock.driver_loop(r, which, BM, K, K2, rows, cols, lo, up, scratch_width=1)
    k = np.fromfile(files[0]).reshape(n, n, order="F")
    k2 = np.fromfile(files[1]).reshape(n, n, order="F")
    assert k[0, 0] == m_all and k2[n - 1, n - 1] == 4 * n and k.sum() == m_all, which
maps = open("/proc/self/maps").read()
assert maps.count("_k_") + maps.count("_k2_") <= 2, "accumulators of earlier analyses are still mapped"
r.call("tpg_release")
maps = open("/proc/self/maps").read()
assert "_k_" not in maps and "_k2_" not in maps and "geno.bk" not in maps

# the whole-analysis entry point with a `which` mask: NULL for what was not asked, bad masks refused
BM = r.fbm(bk, n_all, m_all, code)
out = r.call("tpg_snp_pairwise", BM, r.int(rows), r.int(cols), lib.rmock_lgl(0), r.int([2 | 8]))
assert lib.TYPEOF(lib.VECTOR_ELT(out, 0)) == 0 and lib.TYPEOF(lib.VECTOR_ELT(out, 2)) == 0
assert np.all(r.list_elt(out, 1, (n, n)) == 2.0 * m_all) and np.all(r.list_elt(out, 3, (n, n)) == 4.0 * m_all)
for bad in (0, 16):
    try:
        r.call("tpg_snp_pairwise", BM, r.int(rows), r.int(cols), lib.rmock_lgl(0), r.int([bad]))
        raise SystemExit("bad which mask accepted")
    except RuntimeError:
        pass
lib.R_unload_tpgshim(None)
lib.rmock_reset()
print("ok shim under sanitizers")
