"""Launch-bound regime, seen launch by launch.
  run:        rocprofv3 --kernel-trace -d DIR -o tl -- python3 tools/timeline_one_fit.py run N [fits]
  summarise:  python3 tools/timeline_one_fit.py sum DIR/…_kernel_trace.csv > profiles/rNN_timeline_one_fit_N.txt
  list:       python3 tools/timeline_one_fit.py seq DIR/…_kernel_trace.csv    (every launch: start, duration, workgroups)
`run` does a few value+gradient evaluations of the headline kernel with `fits` (default 1) fits in flight; `sum` takes
the LAST evaluation in the trace (from its feature_kernel launch on) and prints, per kernel symbol, launches, time inside
kernels and the idle time in front of them (start minus the latest end so far, clamped at 0: overlap with the side
stream counts as no gap), then the whole evaluation."""
import csv
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(N, B):
    import numpy as np
    import bench
    from waveome_amd.engine import Context, Problem
    bench.N_HEAD = N
    X, Y = bench.make_inputs(8)
    spec = bench.headline_spec()
    prob = Problem(Context(0), X, Y.T, spec, max_batch=B)
    cols = np.arange(B, dtype=np.int32) % 8
    U = np.tile(spec.unconstrain(), (B, 1))
    for _ in range(4):
        prob.value_grad(cols, U)
    prob.close()


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("wv::", "")
    return n.replace("(anonymous namespace)::", "")


def last_evaluation(path):
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"])))
    rows.sort()
    first = max(i for i, r in enumerate(rows) if r[2].startswith("feature_kernel"))
    return rows[first:]


def sequence(path):
    ev = last_evaluation(path)
    for s, e, n, wg in ev:
        print(f"{(s - ev[0][0]) * 1e-3:9.1f} us  {(e - s) * 1e-3:7.1f} us  {wg:6d} wg  {n}")


def summarise(path):
    ev = [r[:3] for r in last_evaluation(path)]
    per = OrderedDict()
    horizon = ev[0][0]
    for s, e, n in ev:
        d = per.setdefault(n, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) * 1e-3
        d[2] += max(0, s - horizon) * 1e-3
        horizon = max(horizon, e)
    wall = (horizon - ev[0][0]) * 1e-3
    print(f"one evaluation: {len(ev)} launches, {wall:.1f} us from first start to last end")
    print(f"{'kernel':70s} {'launches':>8s} {'in kernel us':>12s} {'idle before us':>14s} {'us / launch':>11s}")
    for n, (c, t, g) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"{n[:70]:70s} {c:8d} {t:12.1f} {g:14.1f} {t / c:11.1f}")
    print(f"{'total':70s} {len(ev):8d} {sum(v[1] for v in per.values()):12.1f} {sum(v[2] for v in per.values()):14.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    elif sys.argv[1] == "seq":
        sequence(sys.argv[2])
    else:
        summarise(sys.argv[2])
