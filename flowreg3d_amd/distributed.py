"""One-volume-per-GPU sharding of a time series.

Volumes of a series are independent given (reference_proc, reference_raw, weight, w_init, params)
-- the reference itself farms them out to processes (multiprocessing_3d.py:286-318).  Here: one
process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in
CPU tests), volume t -> rank ``t mod world``.  The ONLY data collective on the path is the broadcast of
the packed fixed-reference payload from rank 0 (preceded by one ``broadcast_object_list`` of a few
hundred bytes: array shapes and the scalar solver parameters); results stay on the rank that computed
them (each rank writes its own slices of the output), so there is no gather on the data path.
The N > 1 RCCL path has been rehearsed with gloo only (CPU tests, and two ranks sharing one GPU); it runs
under RCCL for the first time on the driver's 8-GPU node.
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


def local_device_index() -> int:
    """The GPU this rank uses -- the same rule as ``_lib.init`` (FR3D_DEVICE, else LOCAL_RANK, modulo the
    number of visible devices), so torch's current device and the engine's device cannot disagree."""
    import os
    from . import _lib
    idx = int(os.environ.get("FR3D_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    n = _lib.device_count()
    return idx % n if n > 0 else idx


def _bind_device(device: Optional[str]):
    """torch.device for collectives on the GPU: cuda:<local_device_index()>, made torch's current device.
    Raises if the engine is already initialised on another GPU (one rank must not straddle two devices)."""
    import torch
    from . import _lib
    idx = int(str(device).split(":")[1]) if device is not None and ":" in str(device) else local_device_index()
    if _lib._inited_device is not None and _lib._inited_device != idx:
        raise RuntimeError(f"engine is initialised on GPU {_lib._inited_device} but this rank's collectives would run on "
                           f"GPU {idx}: set FR3D_DEVICE / LOCAL_RANK consistently (one process per GPU)")
    torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


def _broadcast_packed(payload, scalars, src: int, device: Optional[str], on_device: bool):
    """-> (meta [(name, shape|None)], flat float32 tensor, scalars).  Two collectives in all: one
    broadcast_object_list with the shapes AND the scalar parameters (a few hundred bytes), one broadcast of
    the packed float32 payload -- the path's single data collective."""
    import torch
    dist = _dist()
    rank = dist.get_rank()
    box = [None]
    if rank == src:
        if payload is None:
            raise ValueError("payload required on the source rank")
        box = [([(k, None if v is None else tuple(v.shape)) for k, v in payload.items()], scalars)]
    dist.broadcast_object_list(box, src=src)
    meta, scalars = box[0]
    total = int(sum(int(np.prod(s)) for _, s in meta if s is not None))
    use_cuda = dist.get_backend() == "nccl"
    dev = _bind_device(device) if use_cuda else torch.device("cpu")
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    if rank == src:
        host = np.concatenate([np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
                               for v in payload.values() if v is not None]) if total else np.zeros(0, np.float32)
        flat.copy_(torch.from_numpy(host))
    dist.broadcast(flat, src=src)  # <- the path's single data collective (RCCL over xGMI with backend "nccl")
    if on_device and not use_cuda:
        flat = flat.to(_bind_device(device))  # gloo rehearsal of the device-resident path on a GPU box
    _wait_for_payload(flat)
    return meta, flat, scalars


def _wait_for_payload(flat):
    """RCCL only ENQUEUES the broadcast on torch's communication stream; the engine reads the payload on its
    own non-blocking HIP stream (engine.hip: hipStreamNonBlocking), which no torch stream orders.  Wait on the
    host until the buffer is complete before any engine call may see its address (gloo rehearsal: the upload
    of the received CPU tensor goes through the same wait)."""
    if flat.is_cuda:
        import torch
        torch.cuda.synchronize(flat.device)


def _unpack_host(meta, flat):
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


def broadcast_reference(payload: Optional[Dict[str, Optional[np.ndarray]]], src: int = 0,
                        device: Optional[str] = None) -> Dict[str, Optional[np.ndarray]]:
    """Broadcast {name: float32 array or None} from `src` as ONE packed buffer and return it as host arrays.

    With the nccl backend the buffer lives on this rank's GPU (``local_device_index()``, the engine's own
    rule, made torch's current device) so the transfer is a single RCCL broadcast over xGMI; with gloo it
    is a CPU tensor.  Without an initialised process group this is the identity."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        if payload is None:
            raise ValueError("payload required on a single rank")
        return payload
    meta, flat, _ = _broadcast_packed(payload, None, src, device, False)
    if dist.get_rank() == src:
        return payload
    return _unpack_host(meta, flat)


def process_series_sharded(batch: Optional[np.ndarray], batch_proc: Optional[np.ndarray],
                           reference_raw: Optional[np.ndarray], reference_proc: Optional[np.ndarray],
                           w_init: Optional[np.ndarray], flow_params: Optional[dict],
                           interpolation_method: str = "cubic", executor=None,
                           n_volumes: Optional[int] = None,
                           load_volume=None, device_payload: Optional[bool] = None,
                           window: Optional[int] = None
                           ) -> Tuple[List[int], np.ndarray, np.ndarray]:
    """Register this rank's shard of a series.

    Rank 0 passes the reference payload (reference_raw/proc, w_init and the 4-D weight inside
    flow_params); other ranks may pass None for those and receive them by broadcast.  Each rank
    supplies its volumes either as full arrays `batch`/`batch_proc` (T,Z,Y,X,C) indexed by global t,
    or through `load_volume(t) -> (raw, proc)` with `n_volumes` (data-parallel loading).

    `window`: with `load_volume`, the rank loads and registers its shard `window` volumes at a time (host memory holds
    one window of inputs; BASELINE config 4: 64 time points, 8 per rank, windows of one lock-step batch).

    `device_payload` (default: True with the nccl backend and the built-in executor): the broadcast buffer
    stays in HBM and the engine reads reference, weight and w_init from it (``fr3d_process_batch_raw_dev``) --
    no host copy of the payload on the receiving ranks.  With gloo it can be forced on to rehearse that path
    on a GPU box (the received CPU tensor is uploaded once).

    Returns (global indices handled here, registered (n_local,Z,Y,X,C), flows (n_local,Z,Y,X,3))."""
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    payload = scalars = None
    if rank == 0:
        fp = dict(flow_params or {})
        weight = fp.pop("weight", None)
        if weight is not None and np.asarray(weight).ndim < 4:
            # per-channel / per-voxel weights travel as the (Z,Y,X,C) field get_displacement builds from them
            # (core/optical_flow_3d.py:351-381), so every rank -- and the device-resident path -- sees one layout
            from .core import expand_weight
            Z, Y, X, nc = np.asarray(reference_proc).shape
            weight = expand_weight(weight, Z, Y, X, nc)
        payload = {"reference_raw": reference_raw, "reference_proc": reference_proc, "w_init": w_init,
                   "weight": None if weight is None else np.asarray(weight)}
        scalars = fp
    on_device = False
    flat = meta = None
    if dist and world > 1:
        if device_payload is None:
            device_payload = executor is None and dist.get_backend() == "nccl"
        on_device = bool(device_payload)
        meta, flat, scalars = _broadcast_packed(payload, scalars, 0, None, on_device)
        if not on_device:
            payload = payload if rank == 0 else _unpack_host(meta, flat)
    fp = dict(scalars)

    T = int(n_volumes if n_volumes is not None else batch.shape[0])
    mine = shard_indices(T, rank, world)
    if executor is None:
        from .executor import HipExecutor3D
        executor = HipExecutor3D(device=local_device_index() if (dist and world > 1) else None)
        executor.setup()

    def windows():
        """this rank's volumes as (raw, proc) stacks, one window at a time.  Window k+1 is loaded on a host thread
        while the engine computes window k (the reference's executors overlap reader and workers the same way,
        parallelization/multiprocessing_3d.py:286-318): with window < shard the GPU no longer idles during load_volume.
        Host memory holds two windows of inputs."""
        if load_volume is None:
            if mine:
                yield batch[mine], batch_proc[mine]
            return
        step = len(mine) if not window else max(1, int(window))
        starts = list(range(0, len(mine), max(step, 1)))

        def load(i):
            pairs = [load_volume(t) for t in mine[i:i + step]]
            return np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])

        if len(starts) <= 1:
            for i in starts:
                yield load(i)
            return
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1, thread_name_prefix="fr3d-load") as pool:
            nxt = pool.submit(load, starts[0])
            for n, i in enumerate(starts):
                cur = nxt.result()
                if n + 1 < len(starts):
                    nxt = pool.submit(load, starts[n + 1])  # runs while the caller processes `cur`
                yield cur

    def collect(run):
        regs, flws = [], []
        for raw, proc in windows():
            r, f = run(raw, proc)
            regs.append(r)
            flws.append(f)
        return (np.concatenate(regs), np.concatenate(flws)) if regs else (None, None)

    if on_device:
        # device addresses of the payload's fields inside the broadcast buffer
        ptrs, shapes, off = {}, {}, 0
        for k, shp in meta:
            if shp is None:
                ptrs[k] = None
            else:
                ptrs[k] = flat.data_ptr() + 4 * off
                shapes[k] = shp
                off += int(np.prod(shp))
        Z, Y, X, nc = shapes["reference_proc"]
        if not mine:
            return mine, np.empty((0, Z, Y, X, nc), np.float32), np.empty((0, Z, Y, X, 3), np.float32)
        registered, flows = collect(lambda raw, proc: executor.process_batch_device_refs(
            raw, proc, ptrs, (Z, Y, X, nc), interpolation_method=interpolation_method, flow_params=fp))
        del flat
        return mine, registered, flows
    if payload["weight"] is not None:
        fp["weight"] = payload["weight"]
    ref_raw = payload["reference_raw"]
    ref_proc = payload["reference_proc"]
    if not mine:
        Z, Y, X, nc = ref_proc.shape
        return mine, np.empty((0, Z, Y, X, nc), np.float32), np.empty((0, Z, Y, X, 3), np.float32)
    registered, flows = collect(lambda raw, proc: executor.process_batch(
        raw, proc, ref_raw, ref_proc, payload["w_init"], None, None, interpolation_method=interpolation_method,
        flow_params=fp))
    return mine, registered, flows
