"""A/B of one environment switch of the library in child processes (switches are read once per process): ms per
value + gradient round at 512 / 128 / 64 fits, headline structure, N = 2048; alternating, three repetitions, and a
checksum of the losses (same bits or not).   python3 tools/probe_env_ab.py WV_TRIM 1 0"""
import os, subprocess, sys
CHILD = r'''
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from waveome_amd.engine import Context, Problem
ctx = Context(0); spec = bench.headline_spec()
X, Y = bench.make_inputs(512)
prob = Problem(ctx, X, Y.T, spec, max_batch=512)
out = []
for nb in (512, 128, 64):
    cols = np.arange(nb, dtype=np.int32); U = np.tile(spec.unconstrain(), (nb, 1))
    for _ in range(3): prob.value_grad(cols, U)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(8): f, g, st = prob.value_grad(cols, U)
    ctx.synchronize(); out.append((time.perf_counter() - t0) / 8 * 1e3)
print(" ".join(f"{v:.3f}" for v in out), float(f[:8].sum()))
'''
name, vals = sys.argv[1], sys.argv[2:]
for rep in range(3):
    for v in vals:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **{name: v}), capture_output=True, check=True)
        print(f"{name}={v}: ms per round at 512 / 128 / 64 fits, checksum: {r.stdout.decode().strip()}", flush=True)
