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
    lib = C.CDLL(out)
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
