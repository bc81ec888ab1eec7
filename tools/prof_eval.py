"""Minimal workload for rocprofv3: a few batched value+grad evaluations at the bench size."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waveome_amd.engine import Context, Problem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
X, Y = bench.make_inputs(8)
spec = bench.headline_spec()
ctx = Context(0)
prob = Problem(ctx, X, Y.T, spec, max_batch=B)
cols = np.arange(B, dtype=np.int32) % 8
U = np.tile(spec.unconstrain(), (B, 1))
for _ in range(reps):
    prob.value_grad(cols, U)
prob.close()
