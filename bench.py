"""bench.py - BASELINE.json metric: GP model fits/sec (N=2048, fp64, 100 L-BFGS steps).

One "step" = one pass of the hot path over one batch of synthetic tasks: every rank fits
`--tasks` (default 512 = one full batch of fits in flight; Y column) tasks of the headline workload (N=2048, D=3, kernel
sum_d SE[d] + Periodic(SE)[d], Constant mean, GPflow default start) for exactly 100 L-BFGS-B
iterations (maxiter=100, ftol=gtol=0) on its own GPU.  Weak scaling: per-GPU work is fixed.
For N>1 the driver launches this with torch.distributed.run; X and Y are generated on rank 0,
broadcast over RCCL, and the per-task score records are all-gathered after every step.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant MFMA
GEMM stage, HIP-event timed inside the timed region) and `cpu_baseline` (the NumPy/SciPy oracle
port timed on this box's host cores, N=1 only)."""
import argparse
import json
import os
import sys
import time

import numpy as np

T_PROCESS_START = time.perf_counter()
WALL_BUDGET_S = 540.0  # the driver kills the command at 600 s: optional extras are dropped when this gets close

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MATRIX_TFLOPS = 78.6  # AMD MI355X datasheet FP64 matrix; measured v_mfma_f64_4x4x4_4b_f64
                                # back-to-back issue: 73.4 TF (tools/probe_rates.py), the local guide
                                # lists no f64 matrix rate
N_HEAD, D_HEAD = 2048, 3
GEMM_STAGES = ("chol_trsm", "chol_trailing_syrk", "trtri_tri_gemm", "trtri_diag_gemm", "lauum_gemm", "lauum_gemm_few_fits")
DOMINANT_STAGE = "lauum_gemm"  # K_y^-1 = M'M: the single largest launch (N^3/3 FLOP per fit per evaluation)


def make_inputs(P, seed=9102 + 2):
    """U(-3,3) columns standardised with ddof=1; y_p = sum_d a sin(w x_d) + noise."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3, 3, size=(N_HEAD, D_HEAD))
    X = (X - X.mean(0)) / X.std(0, ddof=1)
    a = rng.normal(size=(P, D_HEAD))
    w = rng.uniform(0.5, 3.0, size=(P, D_HEAD))
    Y = np.stack([sum(a[p, d] * np.sin(w[p, d] * X[:, d]) for d in range(D_HEAD))
                  + 0.1 * rng.normal(size=N_HEAD) for p in range(P)], axis=0)  # P x N
    return X, Y


def headline_spec():
    from waveome_amd import ModelSpec
    from waveome_amd import kernels as K

    kern = K.Sum([K.SquaredExponential(active_dims=[d]) for d in range(D_HEAD)]
                 + [K.Periodic(K.SquaredExponential(active_dims=[d])) for d in range(D_HEAD)])
    return ModelSpec(kern, D_HEAD, mean_function=K.ConstantMean())


def _cpu_time_evals(spec, X, y, n_evals):
    """Seconds for n_evals value+gradient evaluations of the oracle objective (start point, then small steps)."""
    from oracle import gp_oracle as o

    u = o.pack_u(spec, o.theta_init(spec))
    t0 = time.perf_counter()
    for i in range(n_evals):
        o.objective(spec, u + 0.01 * i, X, y)
    return time.perf_counter() - t0


def _cpu_worker_main(task, n_evals, ntasks):
    """`bench.py --cpu-worker TASK EVALS NTASKS`: one process of the one-fit-per-core CPU baseline (single-threaded
    BLAS set by the parent through the environment). Prints one JSON line."""
    X, Y = make_inputs(ntasks)
    spec = headline_spec_dict()
    _cpu_time_evals(spec, X[:64], Y[task][:64], 1)  # page in NumPy / LAPACK before the clock starts
    print(json.dumps({"evals": n_evals, "seconds": _cpu_time_evals(spec, X, Y[task], n_evals)}), flush=True)


def headline_spec_dict():
    return headline_spec().to_dict()


def physical_cores():
    """(physical cores among the CPUs of this process's affinity mask, cgroup CPU quota in cores or None)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                cores.add(fh.read().strip())
        except OSError:
            cores.add(str(c))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    return max(1, len(cores)), quota


def cpu_baseline_measure(X, Y, seconds_budget=24.0):
    """SURVEY 8(d) CPU row, measurement half: the oracle port (NumPy/SciPy -> LAPACK dpotrf/dpotri, every kernel factor
    evaluated once, traces accumulated term by term) on this box's host cores, on a bounded sample of the same
    workload: a few value+gradient evaluations are timed.  Runs BEFORE the process touches the GPU, so that the
    worker processes of mode (b) are never children of a GPU-initialised (or profiler-preloaded) parent.  Two modes:
      (a) all BLAS threads on ONE fit;
      (b) one fit per core (the reference's Ray fan-out, model_search.py:474-489): child processes with
          single-threaded BLAS, one task each.
    Returns evaluations per second for both; `cpu_baseline_finish` converts to fits/s."""
    import subprocess

    spec = headline_spec_dict()
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])
    except Exception:
        blas_threads = None
    half = 0.5 * seconds_budget
    # (a) threaded BLAS, one fit
    _cpu_time_evals(spec, X[:64], Y[0][:64], 1)  # page in NumPy / LAPACK
    t_first = _cpu_time_evals(spec, X, Y[0], 1)
    k_a = int(max(1, min(8, (half - t_first) / max(t_first, 1e-3))))
    dt_a = _cpu_time_evals(spec, X, Y[0], k_a)
    mode_a = {"evals_per_s": k_a / dt_a, "blas_threads": blas_threads, "evaluations": k_a, "seconds": round(dt_a, 2)}
    # (b) one fit per PHYSICAL core of the affinity mask ("all physical cores ... one fit per core"),
    # bounded by a cgroup CPU quota when the box has one; WV_CPU_WORKERS overrides
    phys, quota = physical_cores()
    workers = int(os.environ.get("WV_CPU_WORKERS", 0)) or int(max(1, min(phys, quota or phys, 128, Y.shape[0])))
    try:
        # with every core busy an evaluation is slower than the lone threaded one above (shared L3 / memory bandwidth):
        # 2 - 3 evaluations per worker keep the pool inside its half of the budget
        k_b = 2 if workers > 32 else int(max(2, min(6, half / max(2.0 * t_first, 1e-3))))
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(w), str(k_b),
                                   str(Y.shape[0])], stdout=subprocess.PIPE, env=env, cwd=ROOT)
                 for w in range(workers)]
        outs = []
        for pr in procs:
            so, _ = pr.communicate(timeout=max(90.0, 8 * half))
            outs.append(json.loads(so.decode().strip().splitlines()[-1]))
        wall = time.perf_counter() - t0
        busy = max(o_["seconds"] for o_ in outs)  # slowest worker (interpreter start-up excluded)
        mode_b = {"evals_per_s": workers * k_b / busy, "workers": workers, "blas_threads_per_worker": 1,
                  "evaluations_per_worker": k_b, "seconds": round(busy, 2), "pool_wall_seconds": round(wall, 2)}
    except Exception as exc:  # a refused child process must not cost the bench line
        mode_b = {"error": repr(exc)}
    # the parity half of the metric ("log-marg-lik rel-err"): column 0 at the start point by the oracle port and by a
    # closed form that forms no Gram matrix (oracle/closed_forms.py); the device value is compared after the run
    from oracle import gp_oracle, closed_forms

    theta0 = np.asarray(headline_spec().theta(), dtype=np.float64)
    check = {"oracle_port": float(gp_oracle.lml(spec, theta0, X, Y[0])),
             "closed_form": float(closed_forms.headline_closed_form(theta0, X, Y[0]))}
    return {"ncpu": ncpu, "physical_cores": phys, "cgroup_cpu_quota": quota, "blas_threads": blas_threads,
            "workers": workers, "all_threads_one_fit": mode_a,
            "one_fit_per_core": mode_b, "lml_check": check}


def lml_rel_err(raw, device_lml):
    """{"value": worst relative error of the device LML of column 0 at the start point against the two CPU answers, ...}"""
    c = raw["lml_check"]
    errs = {k: abs(device_lml - v) / abs(v) for k, v in c.items()}
    return {"value": max(errs.values()), "device_lml": device_lml, "vs_oracle_port": errs["oracle_port"],
            "vs_gram_free_closed_form": errs["closed_form"], "tolerance_north_star": 1e-6,
            "what": "log marginal likelihood of bench column 0 at the GPflow default parameters: device (wv_lml) against "
                    "oracle/gp_oracle.py (dense LAPACK) and against oracle/closed_forms.py (the headline kernel's exact "
                    "low-rank series: no Gram matrix); relative"}


def cpu_baseline_finish(raw, evals_per_fit):
    """fits/s of the CPU port = measured evaluations/s / the evaluations a 100-iteration L-BFGS-B fit of this workload
    needs (`evals_per_fit`, measured on the fits of the GPU's timed region): an EXTRAPOLATION from a few timed
    evaluations, not a timed fit (a fit takes minutes on the CPU).  `value` is the better of the two modes."""
    a, b = dict(raw["all_threads_one_fit"]), dict(raw["one_fit_per_core"])
    for m in (a, b):
        if "evals_per_s" in m:
            m["fits_per_s"] = m["evals_per_s"] / evals_per_fit
    rate_a, rate_b = a.get("fits_per_s", 0.0), b.get("fits_per_s", 0.0)
    use_b = rate_b > rate_a
    k = b.get("evaluations_per_worker") if use_b else a.get("evaluations")
    return {"value": max(rate_a, rate_b), "unit": "fits/s",
            "cores": raw["workers"] if use_b else (raw["blas_threads"] or raw["ncpu"]), "kind": "port",
            "host_logical_cpus": os.cpu_count(), "cpus_in_affinity_mask": raw["ncpu"],
            "physical_cores_in_affinity_mask": raw["physical_cores"], "cgroup_cpu_quota_cores": raw["cgroup_cpu_quota"],
            "evals_per_fit_used": evals_per_fit,
            "sample": f"extrapolated from {k} timed value+gradient evaluations per core of oracle/gp_oracle.py "
                      "(NumPy/SciPy: dpotrf + dpotri, term-by-term trace) on the headline workload (N=2048, same X and "
                      "Y columns), timed before the GPU run; fits/s = evaluations/s / the evaluations per 100-iteration "
                      "fit measured on the timed GPU fits; value = better of the two modes; the one-fit-per-core pool "
                      "has one single-threaded worker per physical core of the affinity mask, capped by the cgroup CPU "
                      "quota of the box (cpu.max) when there is one - more workers than the quota only time-slice",
            "all_threads_one_fit": a, "one_fit_per_core": b}


def pmc_traffic(fits_per_launch):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside this process, so
    the figure comes from the committed rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, corrected as
    MI355X_MICROARCH.md prescribes; profiles/r05_traffic.json, made by tools/make_traffic_json.py) of the same
    kernel on the same workload, scaled to the fits one launch covers here.  None when the summary is absent."""
    for name in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as fh:
                k = json.load(fh)["kernels"]["lauum_gemm"]
        except (OSError, KeyError, ValueError):
            continue
        per_fit = (k["read_bytes"] + k["write_bytes"]) / k["fits_per_launch"]
        return per_fit * fits_per_launch, f"profiles/{name} (offline rocprofv3 PMC passes, per fit x fits per launch)"
    return None, None


def mfma_busy_profile():
    """MFMA-busy fraction of the GEMM launches of one 512-fit evaluation round (SQ_VALU_MFMA_BUSY_CYCLES /
    GRBM_GUI_ACTIVE, a rocprofv3 --pmc pass of tools/prof_eval.py 512 1; the counters cannot be read from inside this
    process), from the newest committed summary under profiles/: the Cholesky trailing update A22 -= S S' level by level
    (its own kernel symbol <..., 0, 2>; h = rows of the update, from the launch's grid), K_y^-1 = M'M and its diagonal
    launch."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_eval512_pmc_mfma_busy_by_launch.csv")))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            rows = list(csv.DictReader(fh))
    except (OSError, ValueError):
        return None
    out = {"source": "profiles/" + os.path.basename(files[-1]), "trailing_update_by_level": {}, "fits": 512}
    for r in rows:
        k, busy, us = r["kernel"], float(r["mfma_busy"]), float(r["dur_us"])
        if k.endswith("128, 2, 0, 2>"):
            tiles = int(r["grid"]) // 256 // 512          # strictly-lower 128-tiles per matrix: m (m - 1) / 2
            m = int(round((1 + (1 + 8 * tiles) ** 0.5) / 2))
            lvl = out["trailing_update_by_level"].setdefault(f"h={128 * m}", {"busy": [], "us": []})
            lvl["busy"].append(busy)
            lvl["us"].append(us)
        elif k.endswith("128, 2, 0, 1>"):
            out["mtm_lower_tiles"] = {"busy": busy, "us": us}
        elif k.endswith("128, 2, 0, 5>"):
            out["mtm_diagonal_tiles"] = {"busy": busy, "us": us}
    for lvl in out["trailing_update_by_level"].values():
        lvl["busy"] = round(sum(lvl["busy"]) / len(lvl["busy"]), 3)
        lvl["us"] = round(sum(lvl["us"]) / len(lvl["us"]), 1)
    return out


def bench_c5(args):
    """BASELINE configs[4] / SURVEY 8(d) C5: N = 16384, D = 4, 8 outputs, sum_d Matern32[d], float32 (and float64
    beside it); value-only and value+gradient evaluations timed separately, all 8 outputs in one batch."""
    from waveome_amd import ModelSpec, kernels as K
    from waveome_amd.engine import Context, Problem

    N, D, P = 16384, 4, 8
    rng = np.random.default_rng(9102 + 5)
    X = rng.uniform(-3, 3, size=(N, D))
    X = (X - X.mean(0)) / X.std(0, ddof=1)
    Y = np.stack([np.sin((1.0 + 0.2 * p) * X[:, 0]) + 0.5 * np.sin(2.1 * X[:, 1]) + 0.1 * rng.normal(size=N)
                  for p in range(P)], axis=1)
    spec = ModelSpec(K.Sum([K.Matern32(active_dims=[d]) for d in range(D)]), D, K.GaussianLikelihood(), K.ConstantMean())
    th = spec.theta()
    th[spec.noise] = 0.3
    u = np.tile(spec.unconstrain(th), (P, 1))
    cols = np.arange(P, dtype=np.int32)
    rows = {}
    for dtype in ("float32", "float64"):
        ctx = Context(0, dtype=dtype)
        prob = Problem(ctx, X, Y, spec, max_batch=P)
        for _ in range(max(1, args.warmup)):
            f, g, _ = prob.value_grad(cols, u)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            f, g, _ = prob.value_grad(cols, u)
        ctx.synchronize()
        t_vg = (time.perf_counter() - t0) / args.steps
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lml, _ = prob.lml(cols, np.tile(th, (P, 1)))
        ctx.synchronize()
        t_v = (time.perf_counter() - t0) / args.steps
        prob.close()
        rows[dtype] = {"value_grad_ms": t_vg * 1e3, "value_grad_tflops": P * N ** 3 / t_vg / 1e12,
                       "value_only_ms": t_v * 1e3, "value_only_tflops": P * N ** 3 / 3 / t_v / 1e12,
                       "loss": [float(v) for v in f], "grad_max": float(np.max(np.abs(g)))}
    rel = max(abs(a - b) / abs(b) for a, b in zip(rows["float32"]["loss"], rows["float64"]["loss"]))
    for r in rows.values():
        del r["loss"]
    out = {"metric": "GP value+gradient evaluations/sec (N=16384, fp32)", "value": P / (rows["float32"]["value_grad_ms"] * 1e-3),
           "unit": "evaluations/s", "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup),
           "ms_per_step": rows["float32"]["value_grad_ms"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "C5: N=16384 D=4, 8 outputs in one batch, sum_d Matern32[d], sigma^2=0.3; "
                                  "algorithmic FLOP: N^3 per value+gradient, N^3/3 per value-only evaluation"},
           "float32": rows["float32"], "float64": rows["float64"], "loss_rel_err_f32_vs_f64": rel,
           "roofline": {"bound": "mfma", "kernel": "all stages of one value+gradient evaluation",
                        "achieved": rows["float32"]["value_grad_tflops"], "peak": 157.3, "unit": "TFLOP/s",
                        "frac": rows["float32"]["value_grad_tflops"] / 157.3, "traffic": None}}
    print(json.dumps(out), flush=True)


def bench_c2(args):
    """BASELINE configs[1] / SURVEY 8(d) C2a beside the throughput row: ONE output, isotropic SE(all dims) + noise,
    same N and the same 100-iteration rule - the latency of a single fit (kept out of the default run so that the
    default command's kernel statistics cover the headline workload only)."""
    from waveome_amd import ModelSpec, kernels as K
    from waveome_amd.engine import Context, Problem, lbfgs_opts

    X_h, Y_h = make_inputs(8)
    ctx = Context(0)
    opts = lbfgs_opts(maxiter=100, maxfun=100000, ftol=0.0, gtol=0.0)
    c2 = ModelSpec(K.SquaredExponential(active_dims=list(range(D_HEAD))), D_HEAD, K.GaussianLikelihood(),
                   K.ConstantMean())
    p2 = Problem(ctx, X_h, Y_h.T[:, :1], c2, max_batch=1)
    u2 = c2.unconstrain()[None, :]
    for _ in range(max(1, args.warmup)):
        p2.fit([0], u2, opts)
    ctx.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        r2 = p2.fit([0], u2, opts)
    ctx.synchronize()
    dt2 = (time.perf_counter() - t2) / args.steps
    p2.close()
    out = {"metric": "GP single-output fit latency (N=2048, fp64)", "value": dt2, "unit": "s", "n_gpus": 1,
           "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": dt2 * 1e3, "higher_is_better": False,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "C2a: N=2048 D=3, 1 output, SE(all dims) + noise + Constant mean, L-BFGS-B "
                                  "(maxiter=100, ftol=gtol=0)",
                      "iterations": int(r2["nit"][0]), "evaluations": int(r2["nfev"][0]),
                      "ms_per_evaluation": dt2 * 1e3 / max(1, int(r2["nfev"][0])), "lml": float(r2["lml"][0])}}
    print(json.dumps(out), flush=True)


def launch_ranks(n, argv):
    """`python3 bench.py --gpus N ...` with no WORLD_SIZE in the environment: start the N rank processes ourselves
    (what `python -m torch.distributed.run --nproc-per-node N` does for the driver - that path keeps working, it sets
    WORLD_SIZE).  This parent never touches the GPU and never imports torch or the library: it picks a free rendezvous
    port on 127.0.0.1, starts `bench.py <same arguments>` N times as CHILD processes with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set (no exec), forwards rank 0's ONE JSON line to its own stdout (everything else the children write goes
    to stderr) and returns the worst child exit code.  If a rank dies, the others - which would wait in a collective for
    ever - are ended by PID after a grace period."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL across processes: dmabuf IPC only on this driver
    base.setdefault("OMP_NUM_THREADS", str(max(1, ncpu // n)))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", ROLE_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    import threading

    lines = []

    def read_rank0():
        # rank 0's stdout carries exactly one line (main() parks fd 1 on stderr until the JSON is written)
        for raw in procs[0].stdout:
            txt = raw.decode(errors="replace").rstrip("\n")
            if txt.startswith("{") and '"metric"' in txt:
                lines.append(txt)
            elif txt:
                print(txt, file=sys.stderr)

    reader = threading.Thread(target=read_rank0, daemon=True)
    reader.start()
    rcs = [None] * n
    killed = set()
    grace = float(os.environ.get("WV_LAUNCH_GRACE_S", "30"))
    t_fail = None
    while any(rc is None for rc in rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
        if any(rc not in (None, 0) for rc in rcs) and t_fail is None:
            t_fail = time.perf_counter()
        if t_fail is not None and time.perf_counter() - t_fail > grace:
            for i, pr in enumerate(procs):
                if rcs[i] is None and i not in killed:
                    killed.add(i)
                    pr.kill()       # the exact PID we started
        time.sleep(0.2)
    reader.join(timeout=10.0)
    line = lines[-1] if lines else None
    # the worst exit code of the ranks that ended by themselves (a rank we had to end reports the failure that stranded it)
    worst = max((abs(rc) for i, rc in enumerate(rcs) if i not in killed), default=0) or (1 if killed else 0)
    if line is not None and worst == 0:
        print(line, flush=True)
    elif worst == 0:
        print("bench.py: rank 0 exited 0 without a result line", file=sys.stderr)
        worst = 1
    else:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
    return min(worst, 255)


def _launch_selftest(mode, rank, world):
    """What a rank does under `--launch-selftest`: the rendezvous `launch_ranks` set up, for real, on the CPU."""
    import torch
    import torch.distributed as dist

    if mode == "fail" and rank == world - 1:
        sys.exit(3)
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        print("a stray line a library might print", flush=True)
        print(json.dumps({"metric": "launch-selftest", "world": dist.get_world_size(), "rank_sum": float(t.item()),
                          "local_rank_env": os.environ.get("LOCAL_RANK"), "master_addr": os.environ.get("MASTER_ADDR")}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker_main(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tasks", type=int, default=512,
                    help="fits per GPU per step (one full batch: the driver's --steps 20 --warmup 5 then takes ~6 min)")
    ap.add_argument("--batch", type=int, default=512, help="fits in flight per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the extra untimed per-stage step")
    ap.add_argument("--launch-selftest", default=None, choices=["ok", "fail"],
                    help="CPU-only check of the self-launch path (tests/test_parallel.py): the ranks meet over gloo, "
                         "all-reduce their ranks, rank 0 prints one line; 'fail' makes the last rank exit 3 instead")
    ap.add_argument("--config", default="headline", choices=["headline", "c2", "c5"],
                    help="headline: the BASELINE metric (default).  c2: BASELINE configs[1], one output, single-fit "
                         "latency.  c5: BASELINE configs[4], N=16384 float32 stress")
    args = ap.parse_args()
    if args.config == "c5":
        return bench_c5(args)
    if args.config == "c2":
        return bench_c2(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called like the N = 1 command (`python3 bench.py --gpus N ...`), without a launcher: be the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_selftest:
        return _launch_selftest(args.launch_selftest, rank, world)
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running with the launcher's {world} rank(s)",
              file=sys.stderr)

    from waveome_amd import parallel
    from waveome_amd.engine import Context, Problem, lbfgs_opts

    P = args.tasks * world
    dist = torch = None
    dev = None
    # WV_BENCH_DIST=1 (with torch.distributed.run --nproc-per-node 1): take the N > 1 path - process group, broadcast of
    # the inputs, device-pointer problem, per-step all-gather - at world size 1, which is how the RCCL path is rehearsed
    # on a one-GPU box
    distributed = world > 1 or os.environ.get("WV_BENCH_DIST") == "1"
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a five-line version banner
    # to fd 1 when the process group comes up): park the real stdout and point fd 1 at stderr until the line is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if distributed:
        import torch
        import torch.distributed as dist

        backend = os.environ.get("WV_DIST_BACKEND", "nccl")  # "gloo" only to rehearse on a 1-GPU box
        ndev = torch.cuda.device_count()
        my_dev = local_rank % max(1, ndev)
        if backend == "nccl":
            torch.cuda.set_device(my_dev)
            dist.init_process_group("nccl", device_id=torch.device("cuda", my_dev))
            dev = torch.device("cuda", my_dev)
        else:
            dist.init_process_group(backend)
        X_h, Y_h = make_inputs(P) if rank == 0 else (None, None)
        Xt = parallel.broadcast_array(X_h, 0, dev)  # RCCL broadcast of the design matrix (N x D)
        Yt = parallel.broadcast_array(Y_h, 0, dev)  # and of the outputs (P x N)
        ctx = Context(my_dev)
        spec = headline_spec()
        if dev is not None:
            torch.cuda.synchronize()
            prob = Problem(ctx, None, None, spec, max_batch=args.batch, x_device_ptr=Xt.data_ptr(),
                           y_device_ptr=Yt.data_ptr(), shape=(N_HEAD, D_HEAD, P))
        else:
            X_h, Y_h = Xt.numpy(), Yt.numpy()
            prob = Problem(ctx, X_h, Y_h.T, spec, max_batch=args.batch)
    else:
        X_h, Y_h = make_inputs(P)
        cpu_raw = None
        if not args.no_cpu_baseline:  # first, before anything touches the GPU (see cpu_baseline_measure)
            cpu_raw = cpu_baseline_measure(X_h, Y_h, seconds_budget=24.0)
        ctx = Context(0)
        spec = headline_spec()
        prob = Problem(ctx, X_h, Y_h.T, spec, max_batch=args.batch)  # inputs now resident in HBM

    my_cols = parallel.shard_tasks(P, rank, world).astype(np.int32)
    U0 = np.tile(spec.unconstrain(), (len(my_cols), 1))
    opts = lbfgs_opts(maxiter=100, maxfun=100000, ftol=0.0, gtol=0.0)

    def barrier():
        ctx.synchronize()
        if distributed:
            dist.barrier()
            if dev is not None:
                torch.cuda.synchronize()

    busy = [0.0]  # this rank's own fitting time (the per-step gather, which waits for the slowest rank, excluded)

    def step():
        t_fit = time.perf_counter()
        res = prob.fit(my_cols, U0, opts)
        busy[0] += time.perf_counter() - t_fit
        rec = np.column_stack([res["loss"], res["lml"], res["nit"], res["nfev"], res["status"]])
        return parallel.gather_records(rec, my_cols, P, device=dev), res

    for _ in range(args.warmup):
        step()
    # HIP events (on the streams the kernels run on) bracket ONLY the dominant kernel inside the timed
    # region: one launch per evaluation round. Timing every GEMM launch (~190 event pairs per round) costs
    # ~5 % of throughput, so the full per-stage breakdown comes from one extra, untimed step below.
    prob.timing(True, stages=(DOMINANT_STAGE,))
    prob.timing_reset()
    barrier()
    busy[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        table, res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_busy = [busy[0]]
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if dev is not None else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([busy[0]], dtype=torch.float64, device=dev if dev is not None else "cpu")
        allb = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allb, mine)   # report the imbalance (max / mean rank time)
        rank_busy = [float(b.item()) for b in allb]
    dom_stage = prob.stage_times()[DOMINANT_STAGE]
    # Strong scaling ("headline: 512 -> 64 / GPU"; the reference's Ray rounds, waveome/model_search.py:
    # 474-489): ONE extra, untimed step in which the 512 headline fits are split over the ranks - 512 // world fits per
    # GPU, all in flight at once.  Outside the timed region; steps / warmup / the weak-scaling value are untouched.
    strong = None
    if distributed and world > 1:
        sc = parallel.shard_tasks(min(P, 512), rank, world).astype(np.int32)
        barrier()
        t_s = time.perf_counter()
        r_s = prob.fit(sc, np.tile(spec.unconstrain(), (len(sc), 1)), opts)
        barrier()
        t_s = time.perf_counter() - t_s
        tt = torch.tensor([t_s, float(r_s["nit"].sum()) / 100.0], dtype=torch.float64, device=dev if dev is not None else "cpu")
        tmx = tt.clone()
        dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        strong = {"fits": float(tt[1].item()), "seconds": float(tmx[0].item()), "fits_per_gpu": int(len(sc)),
                  "fits_per_s": float(tt[1].item()) / float(tmx[0].item())}
    stages = None
    wall_used = time.perf_counter() - T_PROCESS_START
    if rank == 0 and not distributed and not args.no_breakdown and wall_used < WALL_BUDGET_S - 60:
        # two rounds back to back, the second one reported: the first wide kernel after an idle gap (the Gram pass) runs
        # at the clock the governor left the idle GPU with - 6.2 instead of 5.1 ms after 50 ms of idleness
        # (tools/probe_gram_clock.py) - and the rounds of a fit loop follow each other within ~0.2 ms
        prob.timing(True)
        n_b = min(len(my_cols), args.batch)
        for _ in range(2):
            prob.timing_reset()
            prob.value_grad(my_cols[:n_b], U0[:n_b])
        stages = prob.stage_times()
    prob.timing(False)

    if rank == 0:
        # a fit = 100 L-BFGS-B iterations; the few tasks whose line search terminates abnormally before
        # iteration 100 are credited only with the iterations they ran (table = last step, every step
        # repeats the same tasks from the same start)
        fits = float(table[:, 2].sum()) / 100.0 * args.steps
        nit_ok = bool(np.all(table[:, 2] == 100))
        ach = dom_stage["flops"] / (dom_stage["ms"] * 1e-3) / 1e12 if dom_stage["ms"] > 0 else 0.0
        traffic, traffic_src = pmc_traffic(args.batch)
        e2e = float(table[:, 3].sum()) * args.steps * float(N_HEAD) ** 3 / elapsed / 1e12 / world
        out = {
            "metric": "GP model fits/sec (N=2048, fp64, 100 L-BFGS steps)",
            "value": fits / elapsed, "unit": "fits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "headline N=2048 D=3 sum_d(SE[d]+Periodic(SE)[d]) + Constant mean, "
                                   "100 L-BFGS-B iterations per fit (maxiter=100, ftol=gtol=0); a fit whose line search "
                                   "cannot improve the objective any further stops before iteration 100, as SciPy's does "
                                   "(profiles/r03_fit_trajectory_parity.txt), and is credited pro rata (iterations / 100)",
                       "fits_per_gpu_per_step": args.tasks, "fits_in_flight_per_gpu": args.batch,
                       "parallelism": f"task-sharded x{world}", "all_fits_ran_100_iterations": nit_ok,
                       "strong_scaling_fits_per_s": None if strong is None else strong["fits_per_s"],
                       "strong_scaling": strong,
                       "rank_fit_seconds": [round(v, 3) for v in rank_busy],
                       "rank_time_max_over_mean": float(max(rank_busy) / (sum(rank_busy) / len(rank_busy))),
                       "mean_evals_per_fit": float(table[:, 3].mean()),
                       "evals_per_s": float(table[:, 3].sum() * args.steps / elapsed),
                       "mean_lml": float(table[:, 1].mean()),
                       "iterations_per_fit": {"min": int(table[:, 2].min()), "mean": float(table[:, 2].mean())},
                       "status_counts": {str(int(k)): int(v) for k, v in
                                         zip(*np.unique(table[:, 4], return_counts=True))}},
            "roofline": {"bound": "mfma",
                         "kernel": f"gemm_f64_kernel<false, false, 128, 2, 0, 1> + its diagonal-tile launch <..., 0, 5> "
                                   f"({DOMINANT_STAGE}: K_y^-1 = M'M, one pair of launches per evaluation round; the "
                                   f"rounds of a step's tail with fewer than ~30 fits in flight run this product on "
                                   f"the small-tile kernel, are timed as lauum_gemm_few_fits and are not counted here)",
                         "achieved": ach, "peak": PEAK_FP64_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP64_MATRIX_TFLOPS,
                         # the whole step against the same peak: N^3 FLOP per value+gradient evaluation
                         # x every evaluation of the timed region / its wall time, per GPU
                         "end_to_end_tflops": e2e, "end_to_end_frac": e2e / PEAK_FP64_MATRIX_TFLOPS,
                         "mfma_busy_by_launch": mfma_busy_profile(),
                         "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "launches": dom_stage["launches"],
                         "avg_launch_ms": dom_stage["ms"] / max(1, dom_stage["launches"]),
                         "algorithmic_flops_per_launch": dom_stage["flops"] / max(1, dom_stage["launches"])},
        }
        if stages is not None:
            tot_fl = sum(stages[k]["flops"] for k in GEMM_STAGES)
            tot_ms = sum(stages[k]["ms"] for k in GEMM_STAGES)
            out["stage_breakdown_untimed_step"] = {
                "note": "two extra value+gradient rounds after the timed region, back to back, every launch bracketed by HIP events; the second one reported",
                "all_gemm_stages_tflops": tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0,
                "ms": {k: round(v["ms"], 3) for k, v in stages.items() if v["launches"]},
                "tflops": {k: round(stages[k]["flops"] / (stages[k]["ms"] * 1e-3) / 1e12, 2)
                           for k in GEMM_STAGES if stages[k]["ms"] > 0},
                "GBps": {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) for k, v in stages.items()
                         if v["bytes"] > 0 and v["ms"] > 0}}
        if not distributed and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_finish(cpu_raw, float(table[:, 3].mean()))
            dev_lml, dev_st = prob.lml([0], spec.theta()[None, :])      # after the timed region
            if int(dev_st[0]) == 0:
                out["cpu_baseline"]["lml_rel_err"] = lml_rel_err(cpu_raw, float(dev_lml[0]))
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    prob.close()
    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
