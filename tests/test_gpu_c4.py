"""BASELINE config C4 at JOB SHAPE inside the suite: N = 1500, D = 8, 32 outputs x the 6
candidate structures = 192 fits, run to the reference's tolerances (maxiter = maxfun = 50000, SciPy's ftol / gtol:
waveome/model_classes.py:310-334) through `GPSearch.kernel_sweep` - what replaces the Ray round of
waveome/model_search.py:434-500 for the workload of examples/iHMP/ihmp_waveome_hpc_run.py:148-176.

  * one process against two ranks (gloo, both on the one GPU of the test box, tasks dealt on demand from the shared
    counter): every rank ends with the whole table, bit-identical to the single process - one checksum;
  * the imbalance of the deal (max / mean rank time) is reported and bounded;
  * a sample of 12 (column, structure) pairs - the two cheapest columns of every structure - against the oracle:
    the oracle's objective at the device's end point (1e-9), SciPy L-BFGS-B on the oracle against the device over the
    first 20 iterations (same iteration and evaluation counts, loss 1e-8), and for the four pairs whose CPU fits take
    about a minute (Lin, SE) the whole run-to-tolerance fit (end loss 1e-6 relative: the north star's bound)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOB = os.path.join(ROOT, "tests", "c4_job.py")
P_OUT = 32

ORACLE_WORKER = r'''
import os, pickle, sys
sys.path.insert(0, os.environ["WV_ROOT"])
import numpy as np
from oracle import gp_oracle as o
job = pickle.load(open(sys.argv[1], "rb"))
d, X, y = job["spec"], job["X"], job["y"]
out = {"obj_at_device_end": float(o.objective(d, job["u_end"], X, y)[0])}
r = o.fit(d, X, y, maxiter=job["cap"])
out["capped"] = {"loss": r["loss"], "nit": r["nit"], "nfev": r["nfev"]}
if job["full"]:
    r = o.fit(d, X, y)
    out["full"] = {"loss": r["loss"], "nit": r["nit"], "nfev": r["nfev"], "message": r["message"]}
pickle.dump(out, open(sys.argv[1] + ".out", "wb"))
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _job(out, env_extra=None, maxiter=None):
    env = dict(os.environ, OMP_NUM_THREADS="2", **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if not env_extra or k not in env_extra:
            env.pop(k, None)
    args = [sys.executable, JOB, out, str(P_OUT)] + ([str(maxiter)] if maxiter else [])
    return subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def test_c4_job_single_process_two_ranks_and_the_oracle(tmp_path):
    cols = ["loss", "lml", "bic", "nit", "nfev", "status"]
    # ---- the job in one process -------------------------------------------------------------------------------------
    single_path = str(tmp_path / "single.pkl")
    p = _job(single_path)
    log, _ = p.communicate(timeout=900)
    assert p.returncode == 0, log.decode()[-3000:]
    single = pickle.load(open(single_path, "rb"))
    tab = single["table"]
    assert len(tab) == P_OUT * 6 and np.all(np.isfinite(tab["loss"])) and set(tab["status"]) <= {0, 1, 2, 5}
    # ---- the same job on two ranks: tasks dealt on demand, table gathered on both --------------------------------------
    port = _free_port()
    out2 = str(tmp_path / "ranks.pkl")
    procs = [_job(out2, {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                         "MASTER_PORT": str(port), "WV_DIST_BACKEND": "gloo"}) for r in range(2)]
    for q in procs:
        log, _ = q.communicate(timeout=900)
        assert q.returncode == 0, log.decode()[-3000:]
    ranks = [pickle.load(open(f"{out2}.{r}", "rb")) for r in range(2)]
    for r in ranks:
        assert r["checksum"] == single["checksum"]                                     # ONE checksum
        np.testing.assert_array_equal(r["table"][cols].to_numpy(), tab[cols].to_numpy())   # ... and every entry
    st = ranks[0]["rank_stats"]
    assert st["mode"] == "dynamic" and sum(st["tasks_per_rank"]) == len(tab) and min(st["tasks_per_rank"]) >= len(tab) // 4
    # both ranks share one GPU here, so the times say how evenly the DEAL went, not how fast a rank is
    assert st["rank_time_max_over_mean"] < 1.4, st   # (192 fits, 469-iteration Periodic tails: 1.2 measured)
    print(f"\nC4 job shape: {len(tab)} fits; one process {single['seconds']:.1f} s; two ranks on one GPU "
          f"{[round(v, 1) for v in st['rank_seconds']]} s, drew {st['tasks_per_rank']}, max/mean "
          f"{st['rank_time_max_over_mean']:.3f}; checksum {single['checksum']:.10f}; mean iterations by structure "
          f"{tab.groupby('structure')['nit'].mean().round(1).to_dict()}")

    # ---- a sample of 12 (column, structure) pairs against the oracle ------------------------------------------------------
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import c4_job
    from waveome_amd import kernels as K
    from waveome_amd.model_classes import PSVGP

    sample = {}          # structure -> its two cheapest columns (by the device's evaluation count)
    for nm in c4_job.STRUCTURES:
        sub = tab[tab["structure"] == nm].sort_values(["nfev", "output"])
        sample[nm] = [int(o_[1:]) for o_ in sub["output"].iloc[:2]]
    columns = sorted({c for v in sample.values() for c in v})
    gps, structures = c4_job.make_search(P_OUT, columns)
    cap = 20
    full_tab = gps.kernel_sweep(structures, keep_models=True)          # (column subset x 6: other batches, same fits)
    models = dict(gps.sweep_models)
    cap_tab = gps.kernel_sweep(structures, keep_models=False, num_opt_iter=cap)
    Xn = gps.X.to_numpy(dtype=np.float64)
    jobs = []
    for nm, cs in sample.items():
        for c in cs:
            key = (f"m{c}", nm)
            row = tab[(tab["output"] == key[0]) & (tab["structure"] == nm)].iloc[0]
            # a fit depends on (column, structure) only - not on what else shared its batch, or on which rank ran it
            assert full_tab.loc[key, "loss"] == row["loss"] and full_tab.loc[key, "nfev"] == row["nfev"], key
            sp = models[key]._spec()                      # the fitted model: its parameters are the device's end point
            yk = gps.Y[key[0]].to_numpy(dtype=np.float64)
            start = PSVGP(Xn, yk[:, None], mean_function=K.ConstantMean(), kernel=structures[nm],
                          penalized_options={"penalization_factor": 0.0})._spec()   # the same structure at the defaults
            path = str(tmp_path / f"oracle_{nm}_{c}.pkl")
            pickle.dump({"spec": start.to_dict(), "X": Xn, "y": yk,
                         "u_end": np.asarray(sp.unconstrain(), dtype=np.float64), "cap": cap,
                         "full": nm in ("lin", "se")}, open(path, "wb"))
            jobs.append((key, path, float(row["loss"]), int(row["nit"]), int(row["nfev"])))
    env = dict(os.environ, WV_ROOT=ROOT, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    workers = [subprocess.Popen([sys.executable, "-c", ORACLE_WORKER, path], env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.STDOUT) for _, path, *_ in jobs]
    report = []
    for (key, path, loss, nit, nfev), w in zip(jobs, workers):
        log, _ = w.communicate(timeout=1500)
        assert w.returncode == 0, log.decode()[-3000:]
        got = pickle.load(open(path + ".out", "rb"))
        # (1) the two objectives are the same function at the device's end point
        assert got["obj_at_device_end"] == pytest.approx(loss, rel=1e-9), key
        # (2) the same optimiser decisions from the start point: 20 iterations, same counts, same loss
        dev = cap_tab.loc[key]
        assert (int(dev["nit"]), int(dev["nfev"])) == (got["capped"]["nit"], got["capped"]["nfev"]), (key, dev, got)
        assert got["capped"]["loss"] == pytest.approx(float(dev["loss"]), rel=1e-8), key
        line = f"{key[0]:>4} {key[1]:>4}: device {nit:4d} it / {nfev:4d} ev, loss {loss:.9f}; oracle at that point {got['obj_at_device_end']:.9f}"
        # (3) whole run-to-tolerance fits where the CPU can afford them
        if "full" in got:
            f = got["full"]
            assert f["loss"] == pytest.approx(loss, rel=1e-6), (key, f, loss)
            line += f"; SciPy on the oracle {f['nit']:4d} it / {f['nfev']:4d} ev, loss {f['loss']:.9f}"
        report.append(line)
    print("\n".join(report))
