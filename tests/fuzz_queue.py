"""Randomised stress of the incremental work queue (wv_queue_*: what run_searches_async and Problem.fit_stream drive):
random task lists over three kernel structures, submitted in random pieces between random numbers of scheduling steps,
fetched with random capacities, on problems of random batch width - every fit must equal, bit for bit, the fit
Problem.fit_multi gives the same task (a slot's arithmetic does not depend on its neighbours, on when it was admitted,
or on how many fits share its launches).  WV_GROUPS (batch groups per problem) is an environment switch read when the
library creates a problem, so the script re-runs itself for 1, 2 and 3 groups.  By hand, on a GPU box:
    python tests/fuzz_queue.py [n_cases=40]"""
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def child(n_cases, seed0):
    import test_gpu_parity as T
    from waveome_amd import ModelSpec, kernels as K
    from waveome_amd.engine import Context, FitQueue, Problem, lbfgs_opts

    ctx = Context(0)
    bad = 0
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 + case)
        N = int(rng.choice([40, 64, 100, 130, 200]))
        X, Y = T._data(seed0 + case, N, P=5)
        kerns = [K.SquaredExponential(active_dims=[0]) + K.Lin(active_dims=[1]),
                 K.Matern32(active_dims=[1]) * K.Categorical(active_dims=[3]) + K.SquaredExponential(active_dims=[0]),
                 K.Periodic(K.SquaredExponential(active_dims=[0])) + K.Categorical(active_dims=[3])]
        specs = [ModelSpec(k, 4, K.GaussianLikelihood(), K.ConstantMean()) for k in kerns]
        if rng.random() < 0.5:
            for sp in specs:
                T._with_priors(sp, scale=1.0)
        T_tasks = int(rng.integers(3, 40))
        cols = rng.integers(0, 5, T_tasks).astype(np.int32)
        sidx = rng.integers(0, 3, T_tasks).astype(np.int32)
        width = max(sp.n_trainable for sp in specs)
        U0 = np.zeros((T_tasks, width))
        for t in range(T_tasks):
            sp = specs[sidx[t]]
            U0[t, :sp.n_trainable] = sp.unconstrain() + 0.3 * rng.normal(size=sp.n_trainable)
        opts = lbfgs_opts(maxiter=int(rng.choice([5, 30, 200])))
        mb = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 33]))
        ref_prob = Problem(ctx, X, Y, specs, max_batch=int(rng.choice([1, 4, 16, 64])))
        ref = ref_prob.fit_multi(cols, sidx, U0, opts)
        ref_prob.close()
        prob = Problem(ctx, X, Y, specs, max_batch=mb)
        q = FitQueue(prob, opts, width=width)
        got = {}
        order = rng.permutation(T_tasks)
        pos = 0
        steps = 0
        while len(got) < T_tasks:
            if pos < T_tasks and rng.random() < 0.6:
                n = int(rng.integers(1, 8))
                ids = order[pos:pos + n]
                q.submit(cols[ids], sidx[ids], U0[ids], tags=ids.astype(np.int64))
                pos += len(ids)
            for _ in range(int(rng.integers(0, 6))):
                done, insys = q.advance()
                steps += 1
            if rng.random() < 0.7 or pos >= T_tasks:
                r = q.fetch(cap=int(rng.integers(1, 12)))
                for k, tag in enumerate(r["tag"]):
                    got[int(tag)] = {f: r[f][k] for f in ("u", "loss", "lml", "nit", "nfev", "status")}
            if pos >= T_tasks:
                q.advance()
                steps += 1
            assert steps < 200000, "the queue does not drain"
        q.close()
        prob.close()
        ok = True
        for t in range(T_tasks):
            g = got[t]
            nt = specs[sidx[t]].n_trainable
            same = (np.array_equal(g["u"][:nt], ref["u"][t][:nt]) and g["loss"] == ref["loss"][t] and g["lml"] == ref["lml"][t]
                    and int(g["nit"]) == int(ref["nit"][t]) and int(g["nfev"]) == int(ref["nfev"][t])
                    and int(g["status"]) == int(ref["status"][t]))
            if not same:
                ok = False
                print(f"  case {seed0 + case} task {t}: queue {g['loss']!r} nit {g['nit']} vs fit_multi {ref['loss'][t]!r} nit {ref['nit'][t]}")
        bad += not ok
        print(f"case {seed0 + case}: N={N} tasks={T_tasks} max_batch={mb} maxiter={opts.maxiter} steps={steps} "
              f"{'ok' if ok else 'MISMATCH'}", flush=True)
    ctx.close()
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(1 if child(int(sys.argv[2]), int(sys.argv[3])) else 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    t0 = time.time()
    total_bad = 0
    for groups in (1, 2, 3):
        env = dict(os.environ, WV_GROUPS=str(groups))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(5000 + 1000 * groups)],
                           env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("case") or l.startswith("  case")]
        badc = sum("MISMATCH" in l for l in lines)
        total_bad += badc + (r.returncode not in (0, 1))
        print(f"WV_GROUPS={groups}: {len([l for l in lines if l.startswith('case')])} cases, {badc} with a mismatch, exit {r.returncode}")
        for l in lines:
            if "MISMATCH" in l or l.startswith("  case"):
                print("   ", l)
        if r.returncode not in (0, 1):
            print(r.stderr[-2000:])
    print(f"# {3 * n} cases in {time.time() - t0:.0f} s: {total_bad} bad")
    sys.exit(1 if total_bad else 0)
