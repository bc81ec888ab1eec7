"""Sharding of the (Y column x kernel structure x restart) task list over the GPUs of a node.

Replaces the Ray fan-out of waveome/model_search.py:411-512 (one remote task per output,
`ray.put` of X / Y, `ray.get` of pickled models).  One process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests):

  1. rank 0 broadcasts X (N x D) and Y (P x N) once           -> `broadcast_array`
  2. every rank fits a static, cost-balanced shard of tasks    -> `shard_tasks`
  3. fixed-width result records are all-gathered               -> `gather_records`

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
    dev = torch.device(device) if device is not None else torch.device("cpu")
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
    if world == 1:
        out = np.empty((n_tasks, width))
        out[np.asarray(local_ids, dtype=np.int64)] = local
        return out
    import torch

    dist = _dist()
    dev = torch.device(device) if device is not None else torch.device("cpu")
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


def sharded_map(n_tasks: int, fn: Callable[[np.ndarray], np.ndarray], costs=None, device=None,
                width: Optional[int] = None) -> np.ndarray:
    """Run fn(task_indices) -> [len, width] records on this rank's shard and gather all records.
    With fewer tasks than ranks some shards are empty: pass the record `width` so that those ranks
    can still take part in the gather."""
    rank, world = dist_info()
    ids = shard_tasks(n_tasks, rank, world, costs)
    if len(ids) == 0:
        if width is None:
            raise ValueError("sharded_map needs n_tasks >= world_size (or an explicit record width)")
        rec = np.zeros((0, int(width)))
    else:
        rec = fn(ids)
    return gather_records(rec, ids, n_tasks, device=device)
