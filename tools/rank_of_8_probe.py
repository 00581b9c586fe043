#!/usr/bin/env python3
"""What ONE rank of an 8-GPU strong-scaling run of the bench step executes, kernel by kernel, measured on one GPU:
  * the shard-local kernels on a 125 000-locus shard of the 5 000 x 1 000 000 panel (pack, counts, Fst sums, the pairwise
    cross-products of the shard, loadings, eigen step): bench.py on that shard;
  * the PCA Gram after the class exchange: rank r owns the loci of ALL shards whose allele-count key falls into its key
    range, so its class Gram is run here on exactly those columns of the whole panel (one run per rank 0 .. 7), next to the
    Gram a rank would run on its own shard without the exchange (class path and digit path).
The exchanges themselves (all-to-all of 140 MB per rank, reduce-scatter of the pair counts, the Gram all-reduce) need the
other GPUs and are not measured.  Writes profiles/<tag>_rank_of_8_kernel_times.json."""
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, ".")
import tidypopgen_amd as tpg  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
n, m, G, R = 5000, 1_000_000, 51, 8
out = {"workload": f"{n} x {m}, {G} populations, k = 20, {R} ranks (strong scaling: 125 000 loci per rank)"}

# shard-local kernels
r = subprocess.run([sys.executable, "bench.py", "--snps", str(m // R), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                    "--no-end-to-end"], capture_output=True, text=True, env=dict(os.environ, TPG_GRAM_NO_EXCHANGE="1"))
line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][0]
b = json.loads(line)
out["shard_local_step_ms"] = b["ms_per_step"]
out["shard_local_kernels_ms"] = b["kernel_ms_per_step"]

ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=G, imputed_bytes=True)
code = np.ascontiguousarray(tpg.CODE_IMPUTE_PRED)
vi = tpg.View(X, None, None, code256=code)
cnt = tpg.loci_counts(vi)
vi.free()
alt = (cnt[:, 1] + 2 * cnt[:, 2]).astype(np.int64)
poly = (alt > 0) & (alt < 2 * n)
key = np.minimum(alt, 2 * n - alt)
hist = np.bincount(key[poly], minlength=n + 1)
cost = np.where(hist > 0, 0.136 + 0.079 * ((hist + 63) // 64), 0.0)  # gramcls.hip: per class a fold, per block 64 loci
cum = np.cumsum(cost)
owner = np.minimum((cum - 1e-9) // (cum[-1] / R), R - 1).astype(int)
rows = []
for rank in range(R):
    cols = (np.where(poly & (owner[key] == rank))[0] + 1).astype(np.int32)
    v = tpg.View(X, None, cols, code256=code)
    c, s = tpg.pca_center_scale(v)
    os.environ["TPG_GRAM_CLASSES"] = "1"
    for rep in range(2):
        ctx.prof_reset()
        tpg.pca_gram(v, c, s)
        ctx.sync()
    d = ctx.prof_dump()
    rows.append({"rank": rank, "loci_owned": int(len(cols)), "classes_owned": int((hist[owner == rank] > 0).sum()),
                 "kernels_ms": {k_: round(ms, 4) for k_, (cnt_, ms) in sorted(d.items()) if k_.startswith(("gcls", "pca_gram", "pca_"))}})
    v.free()
    print("rank", rank, rows[-1], flush=True)
out["class_gram_after_exchange"] = rows
# without the exchange: a rank's own shard, class path and digit path
shard = (np.where(poly[: m // R])[0] + 1).astype(np.int32)
v = tpg.View(X, None, shard, code256=code)
c, s = tpg.pca_center_scale(v)
alone = {}
for name, env in (("classes", {"TPG_GRAM_CLASSES": "1"}), ("digits", {"TPG_GRAM_DIGITS": "1"})):
    os.environ.pop("TPG_GRAM_CLASSES", None)
    os.environ.pop("TPG_GRAM_DIGITS", None)
    os.environ.update(env)
    for rep in range(2):
        ctx.prof_reset()
        tpg.pca_gram(v, c, s)
        ctx.sync()
    alone[name] = {k_: round(ms, 4) for k_, (cnt_, ms) in sorted(ctx.prof_dump().items()) if k_.startswith(("gcls", "pca_gram", "pca_"))}
out["gram_of_the_own_shard_without_exchange"] = alone
out["not_measured"] = ("all-to-all of the packed columns (~140 MB per rank), reduce-scatter of the int32 pair counts (295 MB in all), "
                       "all-reduce of the Gram triangle (100 MB): they need the other GPUs")
os.makedirs("profiles", exist_ok=True)
with open(f"profiles/{tag}_rank_of_8_kernel_times.json", "w") as f:
    json.dump(out, f, indent=1)
print("written")
