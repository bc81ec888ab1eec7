"""The sharded (N > 1) paths of GPSearch on real GPU code: two ranks over gloo, both on GPU 0 (the 1-GPU test box),
must reproduce the single-process results bit for bit - tasks are independent and a slot's arithmetic does not
depend on its batch (tests/test_gpu_parity.py::test_results_do_not_depend_on_batch_composition)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, pickle, sys
import numpy as np, pandas as pd
sys.path.insert(0, os.environ["WV_ROOT"])
world = int(os.environ.get("WORLD_SIZE", "1"))
backend = os.environ.get("WV_BACKEND")
if backend:
    import torch
    import torch.distributed as dist
    if backend == "nccl" and not os.environ.get("WV_NO_SET_DEVICE"):
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world)
    from waveome_amd import parallel
    from waveome_amd.model_classes import default_context
    assert parallel.dist_active() and parallel.dist_backend() == backend
    # one process per GPU: the default context sits on LOCAL_RANK modulo the visible devices
    assert default_context().device == int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    # collectives move device memory under nccl (RCCL), host memory under gloo
    assert parallel.collective_device().type == ("cuda" if backend == "nccl" else "cpu")
from waveome_amd import kernels as K
from waveome_amd.model_search import GPSearch
if os.environ.get("WV_SMALL_CAPS"):   # structures beyond a search session's capacities: the blocking fallback fires
    from waveome_amd import kernel_search as ks
    ks._SearchSession.CAPS = dict(ks._SearchSession.CAPS, max_params=int(os.environ["WV_SMALL_CAPS"]))
rng = np.random.default_rng(3)
n = 60
X = pd.DataFrame({"t": rng.normal(size=n), "dose": rng.normal(size=n), "grp": (np.arange(n) % 3).astype(float)})
Y = pd.DataFrame({f"y{p}": np.sin((1 + 0.3 * p) * X["t"]) + 0.4 * (p % 2) * X["dose"] + 0.2 * rng.normal(size=n)
                  for p in range(5)})
out = {}
gps = GPSearch(X, Y, categorical_vars=["grp"])
gps.penalized_optimization(penalization_factor=5.0)
out["pen"] = {f: (m.fit_result["loss"], m.kernel_name) for f, m in gps.models.items()}
tab = gps.kernel_sweep({"se": K.SquaredExponential(active_dims=[0]),
                        "lin+cat": K.Sum([K.Lin(active_dims=[1]), K.Categorical(active_dims=[2])])}, keep_models=False)
out["sweep"] = tab[["loss", "bic", "nit"]].to_numpy()
gps.run_search(kernels=[K.SquaredExponential(), K.Lin()], max_depth=2, keep_only_best=False)
out["search"] = {f: {n_: m["bic"] for n_, m in gps.search_info[f]["models"].items()} for f in gps.out_names}
out["best"] = {f: gps.search_info[f]["best_model"] for f in gps.out_names}
if backend:
    sh = parallel.broadcast_inputs(X.to_numpy(float), Y.to_numpy(float))
    out["shared_on_device"] = bool(sh._Xt is not None and sh._Xt.is_cuda)
with open(os.environ["WV_OUT"] + "." + os.environ.get("RANK", "s"), "wb") as fh:
    pickle.dump(out, fh)
if backend:
    # (a barrier on a tensor of this rank's own device: dist.barrier() under nccl would use the CURRENT device)
    dist.all_reduce(torch.zeros(1, device=parallel.collective_device()))
    dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(env_extra, out):
    env = dict(os.environ, WV_ROOT=ROOT, WV_OUT=out, OMP_NUM_THREADS="2", **env_extra)
    return subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def test_two_ranks_reproduce_the_single_process_results(tmp_path):
    out = str(tmp_path / "res")
    p = _run({}, out)
    log, _ = p.communicate(timeout=600)
    assert p.returncode == 0, log.decode()[-3000:]
    port = _free_port()
    procs = [_run({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                   "MASTER_PORT": str(port), "WV_BACKEND": "gloo"}, out) for r in range(2)]
    for q in procs:
        log, _ = q.communicate(timeout=600)
        assert q.returncode == 0, log.decode()[-3000:]
    single = pickle.load(open(out + ".s", "rb"))
    for r in range(2):
        got = pickle.load(open(out + f".{r}", "rb"))
        assert got["pen"] == single["pen"]
        np.testing.assert_array_equal(got["sweep"], single["sweep"])
        assert got["search"] == single["search"] and got["best"] == single["best"]


def test_one_rank_over_rccl_reproduces_the_single_process_results(tmp_path):
    """backend "nccl" (= RCCL) at world size 1 - all the 1-GPU box allows: X / Y are broadcast as device tensors,
    the problems are created from those device buffers, the records are all-gathered on the GPU; results must
    equal the plain single-process run bit for bit."""
    out = str(tmp_path / "res")
    p = _run({}, out)
    log, _ = p.communicate(timeout=600)
    assert p.returncode == 0, log.decode()[-3000:]
    q = _run({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
              "MASTER_PORT": str(_free_port()), "WV_BACKEND": "nccl", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, out)
    log, _ = q.communicate(timeout=600)
    assert q.returncode == 0, log.decode()[-3000:]
    single = pickle.load(open(out + ".s", "rb"))
    got = pickle.load(open(out + ".0", "rb"))
    assert got["shared_on_device"] is True
    assert got["pen"] == single["pen"]
    np.testing.assert_array_equal(got["sweep"], single["sweep"])
    assert got["search"] == single["search"] and got["best"] == single["best"]


def test_two_ranks_over_rccl_without_set_device(tmp_path):
    """a caller that never calls `torch.cuda.set_device` - every collective of the library, the dealer's
    key included, must land on the rank's own GPU (`collective_device()`), not on `cuda:0` of every rank.  Needs two GPUs:
    skipped on the one-GPU test box, there for the node the scaling run gets."""
    import ctypes

    from waveome_amd import _lib

    n = ctypes.c_int32(0)   # (through the library's own runtime: importing torch here would bring a second HIP runtime
    _lib.check(_lib.load().wv_device_count(ctypes.byref(n)), "wv_device_count")   #  into the suite's process)
    if n.value < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    out = str(tmp_path / "res")
    p = _run({}, out)
    log, _ = p.communicate(timeout=600)
    assert p.returncode == 0, log.decode()[-3000:]
    port = _free_port()
    procs = [_run({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                   "MASTER_PORT": str(port), "WV_BACKEND": "nccl", "WV_NO_SET_DEVICE": "1",
                   "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, out) for r in range(2)]
    for q in procs:
        log, _ = q.communicate(timeout=600)
        assert q.returncode == 0, log.decode()[-3000:]
    single = pickle.load(open(out + ".s", "rb"))
    for r in range(2):
        got = pickle.load(open(out + f".{r}", "rb"))
        assert got["shared_on_device"] is True
        assert got["pen"] == single["pen"]
        np.testing.assert_array_equal(got["sweep"], single["sweep"])
        assert got["search"] == single["search"] and got["best"] == single["best"]


def test_sharded_search_with_a_structure_beyond_the_session_capacities(tmp_path):
    """in the unsynchronised sharded search a structure that exceeds the session's capacities is scored
    through the blocking path - on the one rank that met it, so that path must not enter a collective (it used to:
    `parallel.dynamic_map`, unmatched on the other rank).  With the capacity lowered so that every product structure
    of the search takes the fallback, two ranks must finish and reproduce the single-process trees."""
    out = str(tmp_path / "res")
    caps = {"WV_SMALL_CAPS": "5"}
    p = _run(caps, out)
    log, _ = p.communicate(timeout=600)
    assert p.returncode == 0, log.decode()[-3000:]
    port = _free_port()
    procs = [_run(dict(caps, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), WV_BACKEND="gloo"), out) for r in range(2)]
    for q in procs:
        log, _ = q.communicate(timeout=600)
        assert q.returncode == 0, log.decode()[-3000:]
    single = pickle.load(open(out + ".s", "rb"))
    for r in range(2):
        got = pickle.load(open(out + f".{r}", "rb"))
        assert got["search"] == single["search"] and got["best"] == single["best"]
