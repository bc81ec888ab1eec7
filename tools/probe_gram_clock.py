"""Does the Gram pass - the first wide kernel after the host's turnaround - pay for the idle gap in front of it?
512 headline fits, value + gradient rounds from Python with the Gram and trace stages timed by HIP events per round:
back to back, with a sleep in front of every round (the clock governor sees an idle GPU), and with only the two
passes timed in a hot loop.  usage (GPU box): python3 tools/probe_gram_clock.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from waveome_amd.engine import Context, Problem

B = 512
X, Y = bench.make_inputs(B)
spec = bench.headline_spec()
prob = Problem(Context(0), X, Y.T, spec, max_batch=B)
rng = np.random.default_rng(0)
U = np.tile(spec.unconstrain(), (B, 1)) + 0.1 * rng.normal(size=(B, spec.n_trainable))
cols = np.arange(B, dtype=np.int32)
prob.value_grad(cols, U)
prob.timing(True, ["gram_assembly", "grad_trace", "lauum_gemm"])
for label, gap in (("back to back", 0.0), ("1 ms idle before each round", 0.001), ("5 ms idle", 0.005),
                   ("50 ms idle", 0.05), ("back to back again", 0.0)):
    rows = []
    for _ in range(6):
        if gap:
            time.sleep(gap)
        prob.timing_reset()
        t0 = time.perf_counter()
        prob.value_grad(cols, U)
        wall = (time.perf_counter() - t0) * 1e3
        t = prob.stage_times()
        rows.append((t["gram_assembly"]["ms"], t["grad_trace"]["ms"], t["lauum_gemm"]["ms"], wall))
    r = np.array(rows[1:])
    print(f"{label:>28}: gram {r[:, 0].mean():.3f} ms (min {r[:, 0].min():.3f})  trace {r[:, 1].mean():.3f}  "
          f"MtM {r[:, 2].mean():.3f}  round wall {r[:, 3].mean():.2f} ms", flush=True)
prob.timing(False)
prob.close()
