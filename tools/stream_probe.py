"""Timeline of one streamed run (tpg_stream_run) on the bench panel: who waits for whom.
usage (GPU box): python tools/stream_probe.py [bk|bed|synth] [budget_bytes] [n] [m]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tidypopgen_amd as tpg  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "bk"
budget = int(float(sys.argv[2])) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
m = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
G, k = 51, 20
gid = (np.arange(n) % G).astype(np.int32)
ctx = tpg.default_context()
if kind == "synth":
    S = tpg.Stream.synth(3, n, m, npop=G, miss=0.02, imputed_bytes=True, budget_bytes=budget)
else:
    X = tpg.FBM.synth(3, n, m, npop=G, miss=0.02, imputed_bytes=True)
    host = X.to_numpy()
    X.free()
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"probe_{os.getpid()}.{kind}")
    if kind == "bed":
        lut = np.array([3, 2, 0, 1], dtype=np.uint8)
        with open(path, "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]))
            for c0 in range(0, m, 65536):
                blk = host[:, c0:c0 + 65536]
                code = lut[np.where(blk > 3, blk - 4, blk)].T
                pad = (-n) % 4
                if pad:
                    code = np.concatenate([code, np.zeros((code.shape[0], pad), dtype=np.uint8)], axis=1)
                q = code.reshape(code.shape[0], -1, 4)
                f.write((q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8).tobytes())
    else:
        host.T.tofile(path)  # column-major bytes
    del host
    os.sync()
    S = (tpg.Stream.open_bed if kind == "bed" else tpg.Stream.open_bk)(path, n, m, budget_bytes=budget)


def run():
    t0 = time.perf_counter()
    r = S.run(pairwise=("ibs", "king", "grm"), groupIds=gid, ngroups=G, alt_freq=True, grouped_alt_freq=True,
              fst=("Hudson", "WC84"), k=k)
    return time.perf_counter() - t0, r["report"]


for i in range(3):
    if i == 2:
        os.environ["TPG_STREAM_TRACE"] = "1"
    # (the library reads the switch per run)
    dt, rep = run()
    print(f"run {i}: {dt * 1e3:.1f} ms  {rep}", flush=True)
S.close()
if kind != "synth":
    os.remove(path)
