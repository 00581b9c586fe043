#!/usr/bin/env python3
"""Time the pack kernels alone (view creation) on the bench panel: one view (L + T), and the pair of views the bench
step packs from one read (L + T4, L)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import tidypopgen_amd as tpg
from tidypopgen_amd import api
ctx = tpg.default_context(); ctx.prof_enable(True)
X = tpg.FBM.synth(3, 5000, 1000000, npop=51, imputed_bytes=True)
imp = np.ascontiguousarray(tpg.CODE_IMPUTE_PRED)
c012 = np.ascontiguousarray(tpg.CODE_012)
for rep in range(4):
    ctx.prof_reset()
    v = tpg.View(X, None, None, code256=imp); ctx.sync()
    ms = ctx.prof_dump()["pack"][1]
    v.free()
print(f"pack  {ms:.3f} ms = {7.5e9/ms/1e9:.2f} TB/s")
for rep in range(4):
    ctx.prof_reset()
    a, b = api.View.pair(X, None, None, c012, imp); ctx.sync()
    ms = ctx.prof_dump()["pack2"][1]
    a.free(); b.free()
print(f"pack2 {ms:.3f} ms = {10e9/ms/1e9:.2f} TB/s")
