"""Sharding of the (Y column x kernel structure x restart) task list over the GPUs of a node.

Replaces the Ray fan-out of waveome/model_search.py:411-512 (one remote task per output,
`ray.put` of X / Y, `ray.get` of pickled models).  One process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests):

  1. rank 0 broadcasts X (N x D) and Y (P x N) once           -> `broadcast_inputs` / `broadcast_array`
  2. every rank fits its share of the tasks: a static, cost-balanced shard (`shard_tasks` / `sharded_map`: jobs
     whose tasks cost the same, e.g. the fixed-iteration bench) or chunks dealt on demand from a shared counter
     (`TaskDealer` / `dynamic_map`: run-to-tolerance jobs, whose fits differ 7x in iterations)
  3. fixed-width result records are all-gathered               -> `gather_records`
Every map also all-gathers each rank's busy time: `last_map_stats()["rank_time_max_over_mean"]` is the load
imbalance SURVEY.md 8(e) asks to be reported.

Collectives run on this rank's GPU under nccl (RCCL moves device memory only) and on the CPU under gloo
(`collective_device`); the GPU of a rank is LOCAL_RANK modulo the visible devices (`local_device_index`).
There is no data-path collective inside a fit: a single fit never spans GPUs.
torch is imported lazily so single-GPU use never pays for it."""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np


def _dist():
    import torch.distributed as dist

    return dist


def dist_info():
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    try:
        import sys

        if "torch" not in sys.modules:
            return 0, 1
        dist = _dist()
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def dist_active() -> bool:
    """True when a torch.distributed process group is initialised (a world of ONE rank included: its
    collectives still run, which is how the RCCL plumbing is rehearsed on a one-GPU box)."""
    import sys

    if "torch" not in sys.modules:
        return False
    try:
        dist = _dist()
        return bool(dist.is_available() and dist.is_initialized())
    except Exception:
        return False


def dist_backend() -> Optional[str]:
    """"nccl" (= RCCL on ROCm), "gloo", ... or None when torch.distributed is not initialised."""
    if not dist_active():
        return None
    return str(_dist().get_backend())


def local_device_index(n_devices: int, env=None) -> int:
    """The GPU ordinal of this process: LOCAL_RANK (torchrun; RANK as a fallback) modulo the visible devices, 0
    outside torch.distributed.  One process per GPU."""
    import os

    env = os.environ if env is None else env
    rank, world = dist_info()
    if not dist_active():
        return 0
    local = env.get("LOCAL_RANK", env.get("RANK", str(rank)))
    return int(local) % max(1, int(n_devices))


def collective_device(device=None):
    """Where the tensors of a collective must live: the caller's choice, else this rank's GPU under the nccl
    backend (RCCL moves device memory only), else the CPU (gloo)."""
    import torch

    if device is not None:
        return torch.device(device)
    if dist_backend() == "nccl":
        return torch.device("cuda", local_device_index(torch.cuda.device_count()))
    return torch.device("cpu")


class SharedInputs:
    """Rank 0's (X, Y) on every rank: host copies for the host-side logic and, under nccl, the device tensors the
    RCCL broadcast filled (X as N x D, Y as P x N row-major - the layout `wv_problem_create` takes), so that a
    device problem can be created from them without another host round trip."""

    def __init__(self, X, Y, Xt=None, Yt=None):
        self.X, self.Y, self._Xt, self._Yt = X, Y, Xt, Yt

    def sorted_by(self, col: Optional[int]) -> "SharedInputs":
        """The same inputs with the rows stably sorted by the labels (rint) of column `col` - the inputs themselves
        when they already are, or when col is None.  Fits do not depend on the row order (beyond rounding), but a
        kernel structure in which every term carries Categorical[col] (unit x numeric interactions,
        waveome/regularization.py:103-121; lone products in the kernel search) is block diagonal in these labels, and
        with sorted rows the device products skip the zero blocks."""
        if col is None:
            return self
        labels = np.rint(self.X[:, int(col)])
        if np.all(np.diff(labels) >= 0):
            return self
        perm = np.argsort(labels, kind="stable")
        Xt = Yt = None
        if self._Xt is not None:
            import torch

            idx = torch.as_tensor(perm, device=self._Xt.device)
            Xt, Yt = self._Xt[idx].contiguous(), self._Yt[:, idx].contiguous()
        return SharedInputs(np.ascontiguousarray(self.X[perm]), np.ascontiguousarray(self.Y[perm]), Xt, Yt)

    def problem(self, ctx, spec, max_batch, capacity=None, nan_is_missing=False):
        """engine.Problem on these inputs: from the broadcast's device tensors when they live on ctx's GPU
        (device-to-device copy inside wv_problem_create), from the host copies otherwise."""
        from .engine import Problem

        if self._Xt is not None and self._Xt.is_cuda and self._Xt.device.index == ctx.device:
            N, D = self.X.shape
            return Problem(ctx, None, None, spec, max_batch=max_batch, x_device_ptr=self._Xt.data_ptr(),
                           y_device_ptr=self._Yt.data_ptr(), shape=(N, D, self.Y.shape[1]), capacity=capacity,
                           nan_is_missing=nan_is_missing)
        return Problem(ctx, self.X, self.Y, spec, max_batch=max_batch, capacity=capacity, nan_is_missing=nan_is_missing)


def broadcast_inputs(X: np.ndarray, Y: np.ndarray, src: int = 0, device=None) -> SharedInputs:
    """Step 1 of the sharded path (replaces `ray.put` of X / Y, waveome/model_search.py:434-472): broadcast rank
    `src`'s design matrix X (N x D) and outputs Y (N x P) - over RCCL when the backend is nccl.  Every rank calls
    it with its own copies (only their shapes matter off-source); without torch.distributed it returns them."""
    rank, world = dist_info()
    if not dist_active():
        return SharedInputs(np.ascontiguousarray(X, dtype=np.float64), np.ascontiguousarray(Y, dtype=np.float64))
    import torch

    dev = collective_device(device)
    Xt = broadcast_array(X if rank == src else None, src, dev)
    Yt = broadcast_array(np.ascontiguousarray(np.asarray(Y, dtype=np.float64).T) if rank == src else None, src, dev)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return SharedInputs(Xt.cpu().numpy(), np.ascontiguousarray(Yt.cpu().numpy().T), Xt, Yt)


def shard_tasks(n_tasks: int, rank: int, world: int, costs: Optional[Sequence[float]] = None) -> np.ndarray:
    """Indices of the tasks `rank` owns: tasks sorted by decreasing cost (stable), dealt round-robin.
    Deterministic, so every rank can reconstruct every other rank's shard."""
    if costs is None:
        order = np.arange(n_tasks)
    else:
        order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    return np.sort(order[rank::world])


def broadcast_array(arr: Optional[np.ndarray], src: int = 0, device=None):
    """Broadcast a float64 array from `src`; returns a torch tensor on `device` (CPU for gloo).
    Non-source ranks pass None."""
    import torch

    dist = _dist()
    rank = dist.get_rank()
    dev = collective_device(device)
    if rank == src:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        meta = torch.tensor([a.ndim] + list(a.shape) + [0] * (4 - a.ndim), dtype=torch.int64, device=dev)
    else:
        meta = torch.zeros(5, dtype=torch.int64, device=dev)
    dist.broadcast(meta, src=src)
    nd = int(meta[0].item())
    shape = [int(v) for v in meta[1:1 + nd].tolist()]
    if rank == src:
        t = torch.from_numpy(a).to(dev)
    else:
        t = torch.empty(shape, dtype=torch.float64, device=dev)
    dist.broadcast(t, src=src)
    return t


def gather_records(local: np.ndarray, local_ids: np.ndarray, n_tasks: int, device=None) -> np.ndarray:
    """All-gather fixed-width float64 records; returns [n_tasks, width] ordered by task id on every rank."""
    rank, world = dist_info()
    local = np.ascontiguousarray(local, dtype=np.float64)
    if len(local_ids):  # (an empty shard arrives as [0, width]: nothing to infer)
        local = local.reshape(len(local_ids), -1)
    width = local.shape[1]
    if not dist_active():
        out = np.empty((n_tasks, width))
        out[np.asarray(local_ids, dtype=np.int64)] = local
        return out
    import torch

    dist = _dist()
    dev = collective_device(device)
    cap = (n_tasks + world - 1) // world  # largest shard
    buf = torch.full((cap, width + 1), -1.0, dtype=torch.float64, device=dev)
    if len(local_ids):
        buf[:len(local_ids), 0] = torch.as_tensor(np.asarray(local_ids, dtype=np.float64), device=dev)
        buf[:len(local_ids), 1:] = torch.from_numpy(local).to(dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = np.full((n_tasks, width), np.nan)
    for b in bufs:
        b = b.cpu().numpy()
        keep = b[:, 0] >= 0
        out[b[keep, 0].astype(np.int64)] = b[keep, 1:]
    return out


_LAST_STATS: dict = {}


def last_map_stats() -> dict:
    """What the most recent sharded_map / dynamic_map on this rank measured: {"mode", "rank_seconds" (busy time of
    every rank between the start of the map and the end of its own work, gather excluded), "tasks_per_rank",
    "rank_time_max_over_mean"} - the imbalance figure of SURVEY.md 8(e)."""
    return dict(_LAST_STATS)


def _record_stats(mode: str, seconds: float, n_mine: int, device=None) -> dict:
    rank, world = dist_info()
    if dist_active():
        import torch

        dev = collective_device(device)
        mine = torch.tensor([float(seconds), float(n_mine)], dtype=torch.float64, device=dev)
        bufs = [torch.empty_like(mine) for _ in range(world)]
        _dist().all_gather(bufs, mine)
        tab = np.array([b.cpu().numpy() for b in bufs])
    else:
        tab = np.array([[float(seconds), float(n_mine)]])
    secs = tab[:, 0]
    _LAST_STATS.clear()
    _LAST_STATS.update({"mode": mode, "rank_seconds": [float(v) for v in secs],
                        "tasks_per_rank": [int(v) for v in tab[:, 1]],
                        "rank_time_max_over_mean": float(secs.max() / secs.mean()) if secs.mean() > 0 else 1.0})
    return last_map_stats()


def sharded_map(n_tasks: int, fn: Callable[[np.ndarray], np.ndarray], costs=None, device=None,
                width: Optional[int] = None) -> np.ndarray:
    """Run fn(task_indices) -> [len, width] records on this rank's STATIC shard and gather all records.
    With fewer tasks than ranks some shards are empty: pass the record `width` so that those ranks
    can still take part in the gather.  Per-rank busy times -> `last_map_stats()`."""
    import time

    rank, world = dist_info()
    ids = shard_tasks(n_tasks, rank, world, costs)
    t0 = time.perf_counter()
    if len(ids) == 0:
        if width is None:
            raise ValueError("sharded_map needs n_tasks >= world_size (or an explicit record width)")
        rec = np.zeros((0, int(width)))
    else:
        rec = fn(ids)
    busy = time.perf_counter() - t0
    out = gather_records(rec, ids, n_tasks, device=device)
    _record_stats("static", busy, len(ids), device)
    return out


# ---------------------------------------------------------------------------------------------------------------
# dynamic dealing: chunks of the task list handed out on demand from a counter every rank can reach
# ---------------------------------------------------------------------------------------------------------------
_DEAL_STORE = None      # (store, id of the process group it was made for)


def set_deal_store(store) -> None:
    """Hand the dealer a key-value store every rank can reach (any `torch.distributed.Store`: the TCPStore / FileStore the
    caller passed to `init_process_group(store=...)`, for instance).  Collective in spirit: every rank must pass a handle
    to the SAME store.  Keys are kept under the prefix "waveome_amd/"."""
    global _DEAL_STORE
    from torch.distributed import PrefixStore

    _DEAL_STORE = (PrefixStore("waveome_amd", store), _group_token())


def _group_token():
    """The current default process group OBJECT (kept alive by the cache entry, so that a later group can never be mistaken
    for it through a recycled id), or None."""
    dist = _dist()
    return dist.group.WORLD if dist.is_initialized() else None


def deal_store():
    """The key-value store behind `TaskDealer`'s shared counter.  The caller's (`set_deal_store`) when one was passed
    for the current process group; otherwise one this module creates, once per process group and collectively: rank 0
    opens a `TCPStore` on a free port of its own address (MASTER_ADDR, else 127.0.0.1), the port travels to the other
    ranks as an int64 tensor on `collective_device()`, and they connect as clients.  Only public torch.distributed API:
    the process group's own rendezvous store is private to torch (`distributed_c10d._get_default_store`) and is not
    touched."""
    global _DEAL_STORE
    if _DEAL_STORE is not None and _DEAL_STORE[1] is _group_token():
        return _DEAL_STORE[0]
    import datetime
    import os

    import torch
    from torch.distributed import PrefixStore, TCPStore

    dist = _dist()
    rank, world = dist_info()
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    timeout = datetime.timedelta(seconds=300)
    server = None
    if rank == 0:
        server = TCPStore(host, 0, world, is_master=True, timeout=timeout, wait_for_workers=False)
    port = torch.tensor([server.port if rank == 0 else 0], dtype=torch.int64, device=collective_device())
    dist.broadcast(port, src=0)
    store = server if rank == 0 else TCPStore(host, int(port.item()), world, is_master=False, timeout=timeout)
    _DEAL_STORE = (PrefixStore("waveome_amd", store), _group_token())
    return _DEAL_STORE[0]


class TaskDealer:
    """Hands out the task list in chunks, on demand.  Under torch.distributed the position lives in a key-value store
    every rank reaches (`deal_store()`: a TCPStore on rank 0, or the store the caller passed; `store.add` is atomic), so a rank
    that finishes early simply draws the next chunk - what Ray's scheduler did for the reference, one task at a time
    (waveome/model_search.py:474-489, 1200-1222).  The sequence is the same on every rank; without a process group
    the counter is local.  Creating a dealer is a collective: rank 0 claims the counter's key and broadcasts it.

    What a chunk holds: the tasks are sorted by decreasing cost (stable) and chunk j takes every n_chunks-th of them
    (j, j + n_chunks, ...), heaviest first - a CROSS-SECTION of the job.  Contiguous pieces of the sorted list would
    hand the first ranks nothing but the most expensive structure: simulated with the measured evaluation counts of
    the C4 job (tools/simulate_dealing.py, profiles/r03_c4_task_table.csv; 3072 tasks on 8 ranks x 512 slots) that
    costs max / mean 1.8 where cross-sections give 1.11 and the static deal 1.14.
    `first_fill`: how many tasks a rank may hold before any of its fits has finished - three quarters of its fair share
    (the rest of the list goes to whoever finishes tasks first); `fit_stream` never holds more than that or
    max_batch + backlog, whichever is smaller."""

    def __init__(self, n_tasks: int, chunk: int, costs: Optional[Sequence[float]] = None, world: Optional[int] = None):
        self.n, self.chunk = int(n_tasks), max(1, int(chunk))
        order = (np.arange(self.n) if costs is None
                 else np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable"))
        n_chunks = max(1, -(-self.n // self.chunk))
        # piece j = every n_chunks-th task of the sorted list, starting at j (sizes differ by at most one); a pull hands
        # out exactly one piece - fixed windows over the concatenation straddled two pieces when n % chunk != 0
        pieces = [order[j::n_chunks] for j in range(n_chunks)] if self.n else []
        self.order = np.concatenate(pieces) if pieces else order
        self._bounds = np.concatenate([[0], np.cumsum([len(q) for q in pieces])]).astype(np.int64)
        world = dist_info()[1] if world is None else int(world)
        self.first_fill = max(self.chunk, int(np.ceil(0.75 * self.n / max(1, world))))
        self.taken = 0          # tasks this rank drew
        self._local = 0
        self._store = None
        self._key = None
        if dist_active():
            # The counter's key has to be the SAME on every rank.  A per-process sequence number is not: a dealer created
            # on some ranks only (an exception, a retry) would shift it silently and every rank would then deal itself the
            # whole list.  Rank 0 claims a number from the store and broadcasts it - creating a dealer is a collective
            # (all callers are), and a rank that creates one alone fails loudly in the broadcast instead.  The number
            # travels as an int64 TENSOR on `collective_device()` like every other collective of this module: an object
            # broadcast would stage on torch.cuda.current_device() under nccl, which is cuda:0 on every rank of a caller
            # that never called torch.cuda.set_device.
            import torch

            self._store = deal_store()
            n = int(self._store.add("deal/next", 1)) if dist_info()[0] == 0 else 0
            t = torch.tensor([n], dtype=torch.int64, device=collective_device())
            _dist().broadcast(t, src=0)
            self._key = f"deal/{int(t.item())}"

    def pull(self) -> np.ndarray:
        """The next piece of task indices (empty once the list is exhausted)."""
        if self._store is not None:
            j = int(self._store.add(self._key, 1)) - 1
        else:
            j, self._local = self._local, self._local + 1
        ids = self.order[self._bounds[j]:self._bounds[j + 1]] if j + 1 < len(self._bounds) else self.order[:0]
        self.taken += len(ids)
        return ids

    def close(self):
        """After every rank has stopped pulling (i.e. behind the gather of the map): the counter leaves the store."""
        if self._store is not None and dist_info()[0] == 0:
            try:
                self._store.delete_key(self._key)
            except Exception:   # (a store without delete_key: the key stays, as before)
                pass


def gather_ragged(local: np.ndarray, local_ids: np.ndarray, n_tasks: int, width: int, device=None) -> np.ndarray:
    """All-gather records of shards whose sizes are only known at run time (dynamic dealing): sizes first, then
    buffers padded to the largest shard.  Returns [n_tasks, width] ordered by task id on every rank."""
    rank, world = dist_info()
    local_ids = np.asarray(local_ids, dtype=np.int64)
    local = np.ascontiguousarray(local, dtype=np.float64).reshape(len(local_ids), int(width))
    if not dist_active():
        out = np.full((n_tasks, int(width)), np.nan)
        out[local_ids] = local
        return out
    import torch

    dist = _dist()
    dev = collective_device(device)
    cnt = torch.tensor([len(local_ids)], dtype=torch.int64, device=dev)
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cap = max(1, max(int(c.item()) for c in cnts))
    buf = torch.full((cap, int(width) + 1), -1.0, dtype=torch.float64, device=dev)
    if len(local_ids):
        buf[:len(local_ids), 0] = torch.as_tensor(local_ids.astype(np.float64), device=dev)
        buf[:len(local_ids), 1:] = torch.from_numpy(local).to(dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = np.full((n_tasks, int(width)), np.nan)
    for b in bufs:
        b = b.cpu().numpy()
        keep = b[:, 0] >= 0
        out[b[keep, 0].astype(np.int64)] = b[keep, 1:]
    return out


def gather_objects(obj, device=None) -> list:
    """All-gather one picklable object per rank (the per-column search results of the sharded kernel search: kernels,
    fit records); returns the list indexed by rank.  Pickled bytes travel as uint8 tensors on `collective_device`, so
    it works under nccl (device memory) and gloo alike; without a process group it returns [obj]."""
    if not dist_active():
        return [obj]
    import pickle

    import torch

    dist = _dist()
    rank, world = dist_info()
    dev = collective_device(device)
    blob = np.frombuffer(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL), dtype=np.uint8)
    n = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    sizes = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = max(1, max(int(v.item()) for v in sizes))
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    buf[:len(blob)] = torch.from_numpy(blob.copy()).to(dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return [pickle.loads(b.cpu().numpy()[:int(v.item())].tobytes()) for b, v in zip(bufs, sizes)]


def dynamic_map(n_tasks: int, run: Callable[["TaskDealer"], tuple], chunk: int, width: int,
                costs=None, device=None) -> np.ndarray:
    """Dynamic counterpart of `sharded_map` for jobs whose tasks differ in cost in ways nobody knows beforehand
    (fits run to SciPy's tolerances: 58 ... 395 iterations per fit in BASELINE config C4, and as much again between
    columns).

    `run(dealer)` does this rank's work: it calls `dealer.pull()` whenever it can take more tasks (an empty array =
    none left; `dealer.first_fill` bounds what it should hold before any of its tasks has finished), and returns (task
    ids, [len, width] records) for everything it drew.  Results do not depend on who fitted what (a fit depends only
    on its task), so the gathered table is bit-identical to a single-process run.  Per-rank busy times ->
    `last_map_stats()`."""
    import time

    dealer = TaskDealer(n_tasks, chunk, costs)
    if dist_active():
        # a common start line: the busy times below are compared across ranks, and a rank that leaves the dealer's
        # broadcast late (100 ms were seen under gloo) would read as imbalance of the dealing
        import torch

        _dist().all_reduce(torch.zeros(1, dtype=torch.float64, device=collective_device(device)))
    t0 = time.perf_counter()
    ids, rec = run(dealer)
    busy = time.perf_counter() - t0
    ids = np.asarray(ids, dtype=np.int64)
    if len(ids) != dealer.taken:
        raise RuntimeError(f"dynamic_map: run() returned {len(ids)} records for the {dealer.taken} tasks it drew")
    out = gather_ragged(np.asarray(rec, dtype=np.float64).reshape(len(ids), int(width)), ids, n_tasks, width, device)
    dealer.close()
    _record_stats("dynamic", busy, len(ids), device)
    return out


def pull_all(fn: Callable[[np.ndarray], np.ndarray]) -> Callable:
    """Adapter: a `run` for dynamic_map out of a plain fn(ids) -> records, one call per chunk (CPU-side work; GPU
    fits use `engine.Problem.fit_stream`, which keeps one batch full across chunks)."""
    def run(dealer):
        all_ids, recs = [], []
        while True:
            ids = dealer.pull()
            if len(ids) == 0:
                break
            all_ids.append(np.asarray(ids))
            recs.append(np.asarray(fn(np.asarray(ids)), dtype=np.float64))
        if not all_ids:
            return np.zeros(0, dtype=np.int64), np.zeros((0, 0))
        return np.concatenate(all_ids), np.concatenate(recs, axis=0)
    return run
