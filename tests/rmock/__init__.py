"""tests/rmock -- NOT R.  Declarations of the R C API functions shim/tpg_rshim.c uses plus a minimal runtime behind
them, so that the shim can be compiled (-Wall -Wextra -Werror) and driven by tests in an image without R."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CFLAGS = ["-std=c11", "-O1", "-Wall", "-Wextra", "-Werror",
          "-Wno-cast-function-type"]  # the (DL_FUNC) casts of every R registration table


def compile_only(extra=()):
    """the compile guard: shim/tpg_rshim.c against include/tpg.h and the stand-in R headers"""
    cmd = ["gcc", *CFLAGS, "-fsyntax-only", "-I" + HERE, "-I" + os.path.join(ROOT, "include"), *extra,
           os.path.join(ROOT, "shim", "tpg_rshim.c")]
    return subprocess.run(cmd, capture_output=True, text=True)


def build(out_dir):
    """libtpgshim_mock.so = the shim + the mock runtime, linked against the HIP library"""
    out = os.path.join(str(out_dir), "libtpgshim_mock.so")
    libdir = os.path.join(ROOT, "tidypopgen_amd")
    cmd = ["gcc", *CFLAGS, "-shared", "-fPIC", "-I" + HERE, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "shim", "tpg_rshim.c"), os.path.join(HERE, "rmock.c"), "-o", out, "-L" + libdir, "-ltpg_hip",
           "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return bind(C.CDLL(out))


def bind(lib):
    """ctypes signatures of the mock runtime's helpers and of the few R API functions the tests call themselves"""
    vp = C.c_void_p
    for name, res, args in (("rmock_new_env", vp, []), ("rmock_env_set", None, [vp, C.c_char_p, vp]),
                            ("rmock_real", vp, [vp, C.c_ssize_t]), ("rmock_int", vp, [vp, C.c_ssize_t]),
                            ("rmock_lgl", vp, [C.c_int]), ("rmock_str", vp, [C.c_char_p]),
                            ("rmock_real_matrix", vp, [vp, C.c_int, C.c_int]), ("rmock_nil", vp, []),
                            ("rmock_data", vp, [vp]), ("rmock_last_error", C.c_char_p, []),
                            ("rmock_call", vp, [vp, C.c_int, C.POINTER(vp)]), ("rmock_reset", None, []),
                            ("XLENGTH", C.c_ssize_t, [vp]), ("TYPEOF", C.c_int, [vp]), ("VECTOR_ELT", vp, [vp, C.c_ssize_t]),
                            ("Rf_getAttrib", vp, [vp, vp]), ("R_unload_tpgshim", None, [vp])):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


class Entry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("fun", C.c_void_p), ("numArgs", C.c_int)]


def entries(lib):
    """the registration table of the shim: {name: (function pointer, arity)}"""
    tab = (Entry * 64).in_dll(lib, "tpg_rshim_entries")
    out = {}
    for e in tab:
        if not e.name:
            break
        out[e.name.decode()] = (e.fun, e.numArgs)
    return out


import numpy as np  # noqa: E402


class Session:
    """the few R objects the drivers handle, built in the mock runtime (tests/test_gpu_rshim.py; bench.py's drop-in leg)"""

    def __init__(self, lib):
        self.lib = lib
        self.ent = entries(lib)

    def fbm(self, path, nrow, ncol, code256=None):
        env = self.lib.rmock_new_env()
        self.lib.rmock_env_set(env, b"backingfile", self.lib.rmock_str(str(path).encode()))
        self.lib.rmock_env_set(env, b"nrow", self.real([float(nrow)]))
        self.lib.rmock_env_set(env, b"ncol", self.real([float(ncol)]))
        if code256 is not None:
            self.lib.rmock_env_set(env, b"code256", self.real(code256))
        return env

    def real(self, v):
        a = np.ascontiguousarray(v, dtype=np.float64)
        return self.lib.rmock_real(a.ctypes.data, a.size)

    def int(self, v):
        a = np.ascontiguousarray(v, dtype=np.int32)
        return self.lib.rmock_int(a.ctypes.data, a.size)

    def matrix(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        return self.lib.rmock_real_matrix(a.ctypes.data, a.shape[0], a.shape[1])

    def call(self, name, *args):
        fn, arity = self.ent["_tidypopgen_" + name]
        assert arity == len(args), (name, arity, len(args))
        arr = (C.c_void_p * max(1, len(args)))(*args)
        out = self.lib.rmock_call(fn, len(args), arr)
        if out is None:
            raise RuntimeError(self.lib.rmock_last_error().decode())
        return out

    def as_numpy(self, sexp, shape=None):
        n = self.lib.XLENGTH(sexp)
        t = self.lib.TYPEOF(sexp)
        ct = C.c_double if t == 14 else C.c_int
        a = np.ctypeslib.as_array(C.cast(self.lib.rmock_data(sexp), C.POINTER(ct)), shape=(n,)).copy()
        return a.reshape(shape, order="F") if shape else a

    def list_elt(self, sexp, k, shape=None):
        return self.as_numpy(self.lib.VECTOR_ELT(sexp, k), shape)


def driver_loop(r, which, BM, K, K2, rows, cols, lo, up, scratch_width=None):
    """the block loop of snp_ibs / snp_king / snp_allele_sharing (R/snp_ibs.R:59-82, R/snp_king.R:51-77,
    R/snp_allele_sharing.R:49-70), scratch matrices included (the shim ignores them): blocks cols[lo[b]-1 : up[b]] (1-based,
    inclusive bounds, as CutBySize returns them)"""
    n = len(rows)
    # the R drivers allocate n x (widest block) scratch matrices (1 GiB each at 5 000 x 26 843); a timing harness may pass
    # narrower ones -- the shim reads only ncol(dos_mat), and only under TPG_EMULATE_AS_PAD_QUIRK=1
    width = int((np.asarray(up) - np.asarray(lo) + 1).max()) if scratch_width is None else scratch_width
    scratch = [r.matrix(np.zeros((n, width))) for _ in range(4)]
    ri = r.int(rows)
    for a, b in zip(lo, up):
        cb = r.int(cols[a - 1:b])
        if which == "ibs":
            r.call("increment_ibs_counts", K, K2, scratch[0], scratch[1], scratch[2], BM, ri, cb)
        elif which == "king":
            r.call("increment_king_numerator", K, K2, scratch[0], scratch[1], scratch[2], scratch[3], BM, ri, cb)
        else:
            r.call("increment_as_counts", K, K2, scratch[0], scratch[1], BM, ri, cb)
