"""Per-kernel time (HIP events of the library's profiler) of one streamed run against the number of blocks: what a block costs
beyond its share of the work.   usage (GPU box): python tools/stream_kernels_probe.py [blocks ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tidypopgen_amd as tpg  # noqa: E402

n, m, G, k = 5000, 1_000_000, 51, 20
gid = (np.arange(n) % G).astype(np.int32)
ctx = tpg.default_context()
tables = {}
os.environ.setdefault("TPG_STREAM_GRAM_BATCH", "0")  # the per-block cost is what is being measured (a synthetic store would batch its Gram)
for blocks in [int(x) for x in sys.argv[1:]] or [1, 2, 8]:
    os.environ["TPG_STREAM_BLOCKS"] = str(blocks)
    S = tpg.Stream.synth(3, n, m, npop=G, miss=0.02, imputed_bytes=True)
    for rep in range(2):
        ctx.prof_enable(rep == 1)
        ctx.prof_reset()
        S.run(pairwise=("ibs", "king", "grm"), groupIds=gid, ngroups=G, alt_freq=True, grouped_alt_freq=True, fst=("Hudson", "WC84"), k=k)
    tables[blocks] = ctx.prof_dump()
    ctx.prof_enable(False)
    S.close()
names = sorted({k_ for t in tables.values() for k_ in t}, key=lambda k_: -max(t.get(k_, (0, 0))[1] for t in tables.values()))
print(f"{'kernel':28s}" + "".join(f"{b:>5d} blk: ms (calls)  " for b in tables))
for k_ in names:
    print(f"{k_:28s}" + "".join(f"{t.get(k_, (0, 0))[1]:12.3f} ({int(t.get(k_, (0, 0))[0]):4d})  " for t in tables.values()))
print(f"{'total':28s}" + "".join(f"{sum(v[1] for v in t.values()):12.3f}         " for t in tables.values()))
