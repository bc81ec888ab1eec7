"""N>1 path on CPU: world_size-2 gloo processes shard a task list, broadcast inputs and gather
fixed-width records; the result must be bit-identical to the single-process run."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["WV_ROOT"])
import torch.distributed as dist
from waveome_amd import parallel
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = parallel.dist_info()
rng = np.random.default_rng(5)
X = rng.normal(size=(40, 3)) if rank == 0 else None
Y = rng.normal(size=(11, 40)) if rank == 0 else None
Xt = parallel.broadcast_array(X).numpy()
Yt = parallel.broadcast_array(Y).numpy()
costs = np.arange(11)[::-1] % 4
def fn(ids):
    return np.column_stack([[float(np.dot(Yt[i], Xt[:, 0])) for i in ids], [Yt[i].sum() * (i + 1) for i in ids], ids.astype(float)])
out = parallel.sharded_map(11, fn, costs=costs)
mine = parallel.shard_tasks(11, rank, world, costs)
np.save(os.environ["WV_OUT"] + f".{rank}.npy", out)
np.save(os.environ["WV_OUT"] + f".ids{rank}.npy", mine)
# fewer tasks than ranks (a search step with a single candidate): the idle rank still joins the gather
one = parallel.sharded_map(1, fn, width=3)
np.save(os.environ["WV_OUT"] + f".one{rank}.npy", one)
# GPSearch-level plumbing: device of a rank, where collectives live, and rank 0's inputs on every rank
assert parallel.dist_active() and parallel.dist_backend() == "gloo"
assert parallel.collective_device().type == "cpu"
assert parallel.local_device_index(8) == rank and parallel.local_device_index(1) == 0
assert parallel.local_device_index(4, env={"LOCAL_RANK": "6"}) == 2
mineX = np.full((40, 3), float(rank)); mineY = np.full((40, 11), float(rank))
sh = parallel.broadcast_inputs(mineX, mineY)      # rank 1's own values must be replaced by rank 0's
assert sh.X.shape == (40, 3) and sh.Y.shape == (40, 11) and np.all(sh.X == 0.0) and np.all(sh.Y == 0.0)
# dynamic dealing (parallel.dynamic_map): a task list whose heavy tasks all sit on even indices - the static
# round-robin deal gives every one of them to rank 0 - is dealt in chunks from the store's counter instead
import json, time
cost_ms = np.ones(64); cost_ms[0:32:2] = 25.0
def work(ids):
    time.sleep(float(cost_ms[ids].sum()) * 1e-3)
    return np.column_stack([ids * 2.0 + 0.5, cost_ms[ids]])
dyn = parallel.dynamic_map(64, parallel.pull_all(work), chunk=2, width=2)
dyn_stats = parallel.last_map_stats()
sta = parallel.sharded_map(64, work, width=2)
sta_stats = parallel.last_map_stats()
np.save(os.environ["WV_OUT"] + f".dyn{rank}.npy", dyn)
np.save(os.environ["WV_OUT"] + f".sta{rank}.npy", sta)
if rank == 0:
    json.dump({"dynamic": dyn_stats, "static": sta_stats}, open(os.environ["WV_OUT"] + ".stats.json", "w"))
# fewer chunks than ranks: a rank that draws nothing still joins the gather
tiny = parallel.dynamic_map(1, parallel.pull_all(work), chunk=4, width=2)
assert tiny.shape == (1, 2) and tiny[0, 0] == 0.5
# the dealer's counter lives in a store this module made with public API only (a TCPStore on rank 0, the port broadcast
# as a tensor) - never in torch's private rendezvous store - and it is made once per process group
st1 = parallel.deal_store()
assert st1 is parallel.deal_store() and type(st1).__name__ == "PrefixStore"
import torch.distributed.distributed_c10d as c10d
assert "_get_default_store" not in open(parallel.__file__).read().replace("`distributed_c10d._get_default_store`", "")
# ... or in one the caller hands over: every rank passes a handle to the same store
from torch.distributed import FileStore
parallel.set_deal_store(FileStore(os.environ["WV_OUT"] + ".filestore", world))
assert parallel.deal_store() is not st1
dyn2 = parallel.dynamic_map(64, parallel.pull_all(lambda ids: np.column_stack([ids * 2.0 + 0.5, cost_ms[ids]])), chunk=3, width=2)
assert np.array_equal(dyn2, dyn) and sum(parallel.last_map_stats()["tasks_per_rank"]) == 64
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_tasks_is_a_balanced_partition():
    from waveome_amd.parallel import shard_tasks

    costs = np.array([5, 1, 9, 9, 2, 7, 3, 3, 8, 1, 6, 4, 2], dtype=float)
    shards = [shard_tasks(len(costs), r, 4, costs) for r in range(4)]
    assert sorted(np.concatenate(shards).tolist()) == list(range(len(costs)))
    loads = [costs[s].sum() for s in shards]
    assert max(loads) - min(loads) <= costs.max()
    assert shard_tasks(5, 0, 1).tolist() == [0, 1, 2, 3, 4]


def test_no_process_group_means_device_zero_and_plain_arrays():
    from waveome_amd import parallel

    assert not parallel.dist_active() and parallel.dist_backend() is None
    assert parallel.local_device_index(8, env={"LOCAL_RANK": "3"}) == 0
    X, Y = np.arange(6.0).reshape(3, 2), np.arange(3.0).reshape(3, 1)
    sh = parallel.broadcast_inputs(X, Y)
    np.testing.assert_array_equal(sh.X, X)
    np.testing.assert_array_equal(sh.Y, Y)
    assert sh._Xt is None


def test_world_size_two_gloo_matches_single_process(tmp_path):
    port = _free_port()
    out = str(tmp_path / "rec")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WV_ROOT=ROOT, WV_OUT=out, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        log, _ = p.communicate(timeout=240)
        assert p.returncode == 0, log.decode()[-2000:]
    r0, r1 = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    np.testing.assert_array_equal(r0, r1)  # every rank holds the full gathered table
    rng = np.random.default_rng(5)
    X, Y = rng.normal(size=(40, 3)), rng.normal(size=(11, 40))
    ids = np.arange(11)
    ref = np.column_stack([[float(np.dot(Y[i], X[:, 0])) for i in ids], [Y[i].sum() * (i + 1) for i in ids], ids.astype(float)])
    np.testing.assert_array_equal(r0, ref)  # bit-identical to the 1-process result
    np.testing.assert_array_equal(np.load(out + ".one0.npy"), ref[:1])
    np.testing.assert_array_equal(np.load(out + ".one1.npy"), ref[:1])
    i0, i1 = np.load(out + ".ids0.npy"), np.load(out + ".ids1.npy")
    assert sorted(np.concatenate([i0, i1]).tolist()) == list(range(11)) and abs(len(i0) - len(i1)) <= 1
    # dynamic dealing: same table as one process would compute, on both ranks, and balanced where the static deal is not
    import json

    ids = np.arange(64)
    cost_ms = np.ones(64)
    cost_ms[0:32:2] = 25.0
    want = np.column_stack([ids * 2.0 + 0.5, cost_ms])
    for tag in ("dyn", "sta"):
        np.testing.assert_array_equal(np.load(out + f".{tag}0.npy"), want)
        np.testing.assert_array_equal(np.load(out + f".{tag}1.npy"), want)
    stats = json.load(open(out + ".stats.json"))
    assert stats["dynamic"]["mode"] == "dynamic" and sum(stats["dynamic"]["tasks_per_rank"]) == 64
    assert stats["dynamic"]["rank_time_max_over_mean"] < 1.15, stats
    assert stats["static"]["rank_time_max_over_mean"] > 1.5, stats   # (what the skew does to a static split)


def test_task_dealer_and_dynamic_map_without_a_process_group():
    from waveome_amd import parallel

    d = parallel.TaskDealer(10, 4, costs=[1, 9, 1, 1, 9, 1, 1, 1, 1, 5], world=2)
    # sorted by decreasing cost (stable): 1 4 9 0 2 3 5 6 7 8; three chunks, chunk j = every third task from j: each a
    # cross-section of the job with its heaviest task first; a pull hands out exactly one of them (sizes 4, 3, 3)
    chunks = [d.pull().tolist() for _ in range(4)]
    assert chunks == [[1, 0, 5, 8], [4, 2, 6], [9, 3, 7], []] and d.taken == 10
    assert d.first_fill == 4            # max(chunk, 3/4 of the fair share of 5)
    assert parallel.TaskDealer(3072, 32, world=8).first_fill == 288
    out = parallel.dynamic_map(7, parallel.pull_all(lambda ids: np.column_stack([ids + 0.25, -ids])), chunk=3, width=2)
    np.testing.assert_array_equal(out, np.column_stack([np.arange(7) + 0.25, -np.arange(7)]))
    st = parallel.last_map_stats()
    assert st["mode"] == "dynamic" and st["tasks_per_rank"] == [7] and st["rank_time_max_over_mean"] == 1.0
    import pytest

    with pytest.raises(RuntimeError):   # a run() that loses tasks it drew is an error, not a silent NaN row
        parallel.dynamic_map(5, lambda dealer: (dealer.pull()[:1], np.zeros((1, 1))), chunk=3, width=1)


def _bench(*args, env=None):
    e = dict(os.environ, OMP_NUM_THREADS="1", **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, timeout=240)


def test_bench_launches_its_own_ranks_when_no_launcher_did():
    """`python3 bench.py --gpus N` without WORLD_SIZE used to print "launch with
    torch.distributed.run" and exit 2.  Now the parent starts N rank children (itself, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set), forwards rank 0's one JSON line and returns the worst child exit code.  The selftest
    ranks really meet over gloo on 127.0.0.1 and all-reduce their ranks; a stray stdout line of rank 0 goes to stderr."""
    import json

    r = _bench("--gpus", "3", "--launch-selftest", "ok")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE line on stdout
    out = json.loads(lines[0])
    assert out == {"metric": "launch-selftest", "world": 3, "rank_sum": 3.0, "local_rank_env": "0", "master_addr": "127.0.0.1"}
    assert "a stray line" in r.stderr.decode()
    # the parent must stay GPU-free: it never imports torch or the library
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("def _launch_selftest")]
    assert "import torch" not in body and "waveome_amd" not in body and "os.exec" not in body


def test_bench_launcher_returns_a_failed_ranks_exit_code():
    """A rank that dies strands the others in their rendezvous: the parent ends them by PID after the grace period and
    reports the code of the rank that failed by itself; no result line."""
    r = _bench("--gpus", "2", "--launch-selftest", "fail", env={"WV_LAUNCH_GRACE_S": "2"})
    assert r.returncode == 3, (r.returncode, r.stderr.decode()[-2000:])
    assert r.stdout.decode().strip() == ""
