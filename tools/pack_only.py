#!/usr/bin/env python3
"""Time the pack kernel alone (view creation) on the bench panel."""
import sys
sys.path.insert(0, ".")
import numpy as np
import tidypopgen_amd as tpg
ctx = tpg.default_context(); ctx.prof_enable(True)
X = tpg.FBM.synth(3, 5000, 1000000, npop=51, imputed_bytes=True)
for rep in range(4):
    ctx.prof_reset()
    v = tpg.View(X, None, None, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED)); ctx.sync()
    ms = ctx.prof_dump()["pack"][1]
    v.free()
print(f"pack {ms:.3f} ms = {7.5e9/ms/1e9:.2f} TB/s")
