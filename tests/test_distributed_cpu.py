"""CPU, world_size 2 over gloo: the one-volume-per-GPU sharding and the reference broadcast.
The GPU executor is replaced by a stand-in built on the oracle, so the test checks the N>1 plumbing
(partition, single payload broadcast, results stay local) against a sequential run."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleExecutor:
    """process_batch with the per-volume body of sequential_3d.py:148-175, on the CPU oracle."""

    def process_batch(self, batch, batch_proc, reference_raw, reference_proc, w_init, gd, im,
                      interpolation_method="cubic", progress_callback=None, **kwargs):
        from oracle import oracle
        fp = kwargs["flow_params"]
        T = batch.shape[0]
        reg = np.empty_like(batch)
        flows = np.empty(batch.shape[:4] + (3,), np.float32)
        for t in range(T):
            f = oracle.get_displacement(reference_proc, batch_proc[t], uvw=w_init.copy(), **fp).astype(np.float32)
            r = oracle.imregister_wrapper(batch[t], f[..., 0], f[..., 1], f[..., 2], reference_raw,
                                          interpolation_method)
            flows[t] = f
            reg[t] = r.reshape(reg[t].shape)
        return reg, flows


def _series():
    from flowreg3d_amd.synthetic import make_pair
    fixed, _, _ = make_pair((10, 14, 14), seed=5)
    vols = [make_pair((10, 14, 14), seed=5, scale=0.2 * (t + 1))[1] for t in range(5)]
    batch = np.stack(vols)[..., None].astype(np.float32)
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=3, iterations=6, min_level=0, levels=3, eta=0.8, a_smooth=1.0,
              a_data=0.45, weight=np.full((10, 14, 14, 1), 1.0))
    return fixed[..., None].astype(np.float32), batch, np.zeros((10, 14, 14, 3), np.float32), fp


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from flowreg3d_amd.distributed import process_series_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fixed, batch, w0, fp = _series()
        if rank == 0:
            mine, reg, flows = process_series_sharded(batch, batch, fixed, fixed, w0, fp, executor=OracleExecutor())
        else:  # non-source ranks own only their volumes; the reference arrives by broadcast
            mine, reg, flows = process_series_sharded(batch, batch, None, None, None, None,
                                                      executor=OracleExecutor())
        q.put((rank, mine, reg, flows))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_series_equals_sequential_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fixed, batch, w0, fp = _series()
    reg_seq, flows_seq = OracleExecutor().process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp)
    seen = []
    for rank, mine, reg, flows in got:
        assert mine == list(range(rank, 5, 2))
        assert np.array_equal(reg, reg_seq[mine]) and np.array_equal(flows, flows_seq[mine])
        seen += mine
    assert sorted(seen) == list(range(5))


def test_broadcast_is_identity_without_process_group():
    from flowreg3d_amd.distributed import broadcast_reference
    p = {"a": np.ones(3, np.float32), "b": None}
    assert broadcast_reference(p) is p


def _prefetch_worker(rank, world, port, q):
    """windows of one volume: the loader must be called for window k+1 BEFORE process_batch of window k returns"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import threading
    import time
    import torch.distributed as dist
    from flowreg3d_amd.distributed import process_series_sharded, shard_indices
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fixed, batch, w0, fp = _series()
        events = []
        lock = threading.Lock()

        def load_volume(t):
            with lock:
                events.append(("load", t, threading.get_ident()))
            return batch[t], batch[t]

        class Slow(OracleExecutor):
            def process_batch(self, b, bp, *a, **k):
                with lock:
                    events.append(("begin", None, threading.get_ident()))
                out = super().process_batch(b, bp, *a, **k)
                time.sleep(0.3)  # the next window's load has ample time to start
                with lock:
                    events.append(("end", None, threading.get_ident()))
                return out

        args = (None, None, fixed, fixed, w0, fp) if rank == 0 else (None, None, None, None, None, None)
        mine, reg, flows = process_series_sharded(*args, executor=Slow(), load_volume=load_volume, n_volumes=5, window=1)
        assert mine == shard_indices(5, rank, world)
        q.put((rank, mine, events, reg, flows))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_loader_prefetches_the_next_window_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_prefetch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fixed, batch, w0, fp = _series()
    seq_reg, seq_flows = OracleExecutor().process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp)
    for rank, mine, events, reg, flows in got:
        kinds = [e[0] for e in events]
        # every volume of the shard was loaded exactly once, in order
        assert [e[1] for e in events if e[0] == "load"] == mine
        # window k+1's load is called BEFORE process_batch of window k returns, on another thread
        main_thread = next(e[2] for e in events if e[0] == "begin")
        for k in range(len(mine) - 1):
            e = [i for i, x in enumerate(kinds) if x == "end"][k]
            nxt = next(i for i, x in enumerate(events) if x[0] == "load" and x[1] == mine[k + 1])
            assert nxt < e, (rank, k, kinds)
            assert events[nxt][2] != main_thread
        # and the results are those of the sequential run
        assert np.array_equal(flows, seq_flows[mine]) and np.array_equal(reg, seq_reg[mine])
