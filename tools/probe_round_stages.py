"""Per-stage time of one value+gradient round of the headline structure with B fits in flight (HIP events around every
launch: `Problem.timing`), beside the wall time of an untimed round: where a round goes as the batch shrinks (the 64
fits per GPU of the headline job on eight GPUs, the tail of a job).
usage (GPU box): python3 tools/probe_round_stages.py [N=2048] [B ...]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from waveome_amd.engine import Context, Problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Bs = [int(v) for v in sys.argv[2:]] or [64, 512]
bench.N_HEAD = N
ctx = Context(0)
spec = bench.headline_spec()
for B in Bs:
    X, Y = bench.make_inputs(B)
    prob = Problem(ctx, X, Y.T, spec, max_batch=B)
    cols = np.arange(B, dtype=np.int32)
    U = np.tile(spec.unconstrain(), (B, 1))
    prob.value_grad(cols, U)
    t0 = time.perf_counter()
    for _ in range(5):
        prob.value_grad(cols, U)
    wall = (time.perf_counter() - t0) / 5
    prob.timing(True)
    prob.timing_reset()
    prob.value_grad(cols, U)
    st = prob.stage_times()
    prob.timing(False)
    tot = sum(v["ms"] for v in st.values())
    print(f"N={N} fits={B}: wall {wall * 1e3:.3f} ms per round ({wall * 1e3 / B:.4f} per fit), in kernels {tot:.3f} ms")
    for k, v in st.items():
        if v["launches"]:
            tf = f"{v['flops'] / (v['ms'] * 1e-3) / 1e12:6.1f} TF" if v["flops"] > 0 else ""
            print(f"   {k:24s} {v['launches']:4d} launches {v['ms']:8.3f} ms {100 * v['ms'] / tot:5.1f} %  {tf}")
    prob.close()
