"""One-volume-per-GPU sharding of a time series.

Volumes of a series are independent given (reference_proc, reference_raw, weight, w_init, params)
-- the reference itself farms them out to processes (multiprocessing_3d.py:286-318).  Here: one
process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in
CPU tests), volume t -> rank ``t mod world``.  The ONLY collective on the path is the broadcast of
the fixed-reference payload from rank 0; results stay on the rank that computed them (each rank
writes its own slices of the output), so there is no gather on the data path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def shard_indices(n_volumes: int, rank: int, world: int) -> List[int]:
    """Static block-cyclic partition: volume t belongs to rank t % world."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_volumes, world))


def _dist():
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return None
    return dist


def broadcast_reference(payload: Optional[Dict[str, Optional[np.ndarray]]], src: int = 0,
                        device: Optional[str] = None) -> Dict[str, Optional[np.ndarray]]:
    """Broadcast {name: float32 array or None} from `src` as ONE packed buffer.

    With the nccl backend the buffer lives on `device` (cuda:LOCAL_RANK) so the transfer is a
    single RCCL broadcast over xGMI; with gloo it is a CPU tensor.  Returns the payload on every
    rank (the same dict on `src`).  Without an initialised process group this is the identity."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        if payload is None:
            raise ValueError("payload required on a single rank")
        return payload
    import torch
    rank = dist.get_rank()
    meta = None
    if rank == src:
        if payload is None:
            raise ValueError("payload required on the source rank")
        meta = [(k, None if v is None else tuple(v.shape)) for k, v in payload.items()]
    box = [meta]
    dist.broadcast_object_list(box, src=src)  # shapes only (a few bytes)
    meta = box[0]
    total = int(sum(int(np.prod(s)) for _, s in meta if s is not None))
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device(device if device is not None else (f"cuda:{torch.cuda.current_device()}" if use_cuda else "cpu"))
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    if rank == src:
        host = np.concatenate([np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
                               for v in payload.values() if v is not None]) if total else np.zeros(0, np.float32)
        flat.copy_(torch.from_numpy(host))
    dist.broadcast(flat, src=src)  # <- the path's single data collective
    if rank == src:
        return payload
    host = flat.cpu().numpy()
    out: Dict[str, Optional[np.ndarray]] = {}
    off = 0
    for k, s in meta:
        if s is None:
            out[k] = None
        else:
            n = int(np.prod(s))
            out[k] = host[off:off + n].reshape(s).copy()
            off += n
    return out


def process_series_sharded(batch: Optional[np.ndarray], batch_proc: Optional[np.ndarray],
                           reference_raw: Optional[np.ndarray], reference_proc: Optional[np.ndarray],
                           w_init: Optional[np.ndarray], flow_params: Optional[dict],
                           interpolation_method: str = "cubic", executor=None,
                           n_volumes: Optional[int] = None,
                           load_volume=None) -> Tuple[List[int], np.ndarray, np.ndarray]:
    """Register this rank's shard of a series.

    Rank 0 passes the reference payload (reference_raw/proc, w_init and the 4-D weight inside
    flow_params); other ranks may pass None for those and receive them by broadcast.  Each rank
    supplies its volumes either as full arrays `batch`/`batch_proc` (T,Z,Y,X,C) indexed by global t,
    or through `load_volume(t) -> (raw, proc)` with `n_volumes` (data-parallel loading).

    Returns (global indices handled here, registered (n_local,Z,Y,X,C), flows (n_local,Z,Y,X,3))."""
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    payload = None
    if rank == 0:
        fp = dict(flow_params or {})
        weight = fp.pop("weight", None)
        payload = {"reference_raw": reference_raw, "reference_proc": reference_proc, "w_init": w_init,
                   "weight": None if weight is None else np.asarray(weight)}
        scalars = fp
    else:
        scalars = None
    payload = broadcast_reference(payload, src=0)
    if dist and world > 1:
        box = [scalars]
        dist.broadcast_object_list(box, src=0)
        scalars = box[0]
    fp = dict(scalars)
    if payload["weight"] is not None:
        fp["weight"] = payload["weight"]

    T = int(n_volumes if n_volumes is not None else batch.shape[0])
    mine = shard_indices(T, rank, world)
    if load_volume is not None:
        pairs = [load_volume(t) for t in mine]
        local_raw = np.stack([p[0] for p in pairs]) if pairs else None
        local_proc = np.stack([p[1] for p in pairs]) if pairs else None
    else:
        local_raw = batch[mine]
        local_proc = batch_proc[mine]
    if executor is None:
        from .executor import HipExecutor3D
        executor = HipExecutor3D()
        executor.setup()
    ref_raw = payload["reference_raw"]
    ref_proc = payload["reference_proc"]
    if not mine:
        Z, Y, X, nc = ref_proc.shape
        return mine, np.empty((0, Z, Y, X, nc), np.float32), np.empty((0, Z, Y, X, 3), np.float32)
    registered, flows = executor.process_batch(local_raw, local_proc, ref_raw, ref_proc, payload["w_init"],
                                               None, None, interpolation_method=interpolation_method,
                                               flow_params=fp)
    return mine, registered, flows
