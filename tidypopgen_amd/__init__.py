"""tidypopgen's genotype-matrix hot path, MI355X-native (HIP kernels behind a C ABI).

`import tidypopgen_amd` needs the built shared library (tidypopgen_amd/libtpg_hip.so);
there is no CPU fallback.  See DESIGN.md and INTEGRATION.md.
"""
from . import _lib  # noqa: F401  (raises ImportError if the HIP library is missing)
from .api import *  # noqa: F401,F403
from .api import (CODE_012, CODE_IMPUTE_PRED, FBM, Comm, Context, Multi, Pairwise, ShardedPairwise, View, default_context)  # noqa: F401
