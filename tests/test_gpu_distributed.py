"""-m gpu: the N > 1 paths on a one-GPU box, two ranks over gloo sharing device 0 (the RCCL call itself
first runs on the driver's 8-GPU node; everything around it is the same code):

  * process_series_sharded with the built-in HipExecutor3D, once with host payload arrays and once with the
    broadcast buffer kept in HBM (device_payload=True -> fr3d_process_batch_raw_dev reads reference, weight and
    w_init from it): each rank's shard equals the single-process result bit for bit;
  * bench.py's own N > 1 branch under torch.distributed.run (FR3D_DIST_BACKEND=gloo, cfg1 workload): exactly
    one JSON line, from rank 0, with n_gpus 2 and the world size / backend recorded in `config`.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
from flowreg3d_amd.distributed import process_series_sharded
from flowreg3d_amd.synthetic import make_pair
dist.init_process_group("gloo")
rank = dist.get_rank()
shape = (12, 20, 18)
fixed, _, _ = make_pair(shape, seed=5, channels=2)
batch = np.stack([make_pair(shape, seed=5, channels=2, scale=0.2 * (t + 1))[1] for t in range(5)]).astype(np.float32)
w0 = np.full(shape + (3,), 0.1, np.float32)
fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=3, iterations=8, min_level=0, levels=3, eta=0.8, a_smooth=1.0,
          a_data=0.45, weight=np.array([0.7, 0.3]))
out = {{}}
for mode in (False, True):
    if rank == 0:
        mine, reg, flows = process_series_sharded(batch, batch, fixed, fixed, w0, fp, device_payload=mode)
    else:
        mine, reg, flows = process_series_sharded(batch, batch, None, None, None, None, device_payload=mode)
    out["mine"] = np.array(mine)
    out["reg%d" % mode] = reg
    out["flows%d" % mode] = flows
np.savez({out!r} + "_%d.npz" % rank, **out)
dist.barrier()
dist.destroy_process_group()
"""


def test_sharded_series_two_ranks_host_and_device_payload(hip, tmp_path):
    from flowreg3d_amd.executor import HipExecutor3D
    from flowreg3d_amd.synthetic import make_pair
    script = tmp_path / "worker.py"
    out = str(tmp_path / "res")
    script.write_text(WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, FR3D_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    shape = (12, 20, 18)
    fixed, _, _ = make_pair(shape, seed=5, channels=2)
    batch = np.stack([make_pair(shape, seed=5, channels=2, scale=0.2 * (t + 1))[1] for t in range(5)]).astype(np.float32)
    w0 = np.full(shape + (3,), 0.1, np.float32)
    fp = dict(alpha=(0.25, 0.25, 0.25), update_lag=3, iterations=8, min_level=0, levels=3, eta=0.8, a_smooth=1.0,
              a_data=0.45, weight=np.array([0.7, 0.3]))
    reg_seq, flows_seq = HipExecutor3D().process_batch(batch, batch, fixed, fixed, w0, None, None, flow_params=fp)
    seen = []
    for rank in range(2):
        g = np.load(out + "_%d.npz" % rank)
        mine = g["mine"].tolist()
        assert mine == list(range(rank, 5, 2))
        for mode in (0, 1):
            assert np.array_equal(g["flows%d" % mode], flows_seq[mine]), (rank, mode)
            assert np.array_equal(g["reg%d" % mode], reg_seq[mine]), (rank, mode)
        seen += mine
    assert sorted(seen) == list(range(5))


def test_bench_n2_branch_prints_one_line_from_rank0():
    env = dict(os.environ, FR3D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--workload", "cfg1", "--condition", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["world_size"] == 2 and d["config"]["dist_backend"] == "gloo"
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d and "cfg3" not in d  # rank 0 at N = 1 only
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["launches"] > 0
    # what the first RCCL run is checked by: every rank's own rate and the one data collective
    pr = d["config"]["per_rank_volumes_per_sec"]
    assert len(pr) == 2 and all(v > 0 for v in pr) and d["value"] <= sum(pr) * (1 + 1e-9)
    assert d["config"]["broadcast"]["bytes"] == 32 * 64 * 64 * 4 and d["config"]["broadcast"]["ms"] > 0


def test_bench_refuses_a_world_size_that_does_not_match_gpus():
    env = dict(os.environ, FR3D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1",
           "--warmup", "0", "--workload", "cfg1", "--condition", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "does not match" in (r.stderr + r.stdout)


CFG4_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
from flowreg3d_amd.distributed import process_series_sharded
from flowreg3d_amd.synthetic import fast_pair, SOLVER_DEFAULTS
dist.init_process_group("gloo")
rank = dist.get_rank()
n, T = 256, 8  # BASELINE config 4 as stated: 256^3 volumes of one series against a fixed reference, volume t -> rank t % N
loaded = []
def load_volume(t):
    # data-parallel loading: every rank produces (here: synthesises) only its own time points
    loaded.append(t)
    s = (1.7 * (0.6 + 0.1 * t), -1.1 * (0.6 + 0.1 * t), 0.6 * (0.6 + 0.1 * t))
    _, moving, _ = fast_pair((n, n, n), shift=s)
    return moving[..., None], moving[..., None]
fixed = fast_pair((n, n, n))[0][..., None] if rank == 0 else None
fp = dict(SOLVER_DEFAULTS, levels=4) if rank == 0 else None
w0 = np.zeros((n, n, n, 3), np.float32) if rank == 0 else None
mine, reg, flows = process_series_sharded(None, None, fixed, fixed, w0, fp, n_volumes=T, load_volume=load_volume,
                                          window=2, device_payload=True)
assert loaded == mine == list(range(rank, T, 2)), (loaded, mine)
np.savez({out!r} + "_%d.npz" % rank, mine=np.array(mine), mean_flow=flows[:, 48:-48, 48:-48, 48:-48].mean(axis=(1, 2, 3)), flow0=flows[0, ::8, ::8, ::8],
         reg_shape=np.array(reg.shape))
dist.barrier()
dist.destroy_process_group()
"""


def test_cfg4_as_stated_two_ranks_windows_and_device_payload(hip, tmp_path):
    """BASELINE config 4's harness end to end on two gloo ranks sharing the GPU: 256^3 time points of one series, volume
    t -> rank t % N, each rank loads ONLY its own time points (load_volume) in windows of 2, the reference payload is
    broadcast once and read from HBM (device_payload), results stay on the rank.  Every volume recovers its motion
    and rank 0's first volume equals the single-process result bit for bit."""
    import flowreg3d_amd as fr
    from flowreg3d_amd.synthetic import fast_pair, SOLVER_DEFAULTS
    script = tmp_path / "worker4.py"
    out = str(tmp_path / "c4")
    script.write_text(CFG4_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, FR3D_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", FR3D_BATCH="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    n = 256
    for rank in range(2):
        g = np.load(out + "_%d.npz" % rank)
        mine = g["mine"].tolist()
        assert mine == list(range(rank, 8, 2)) and g["reg_shape"].tolist() == [4, n, n, n, 1]
        for q, t in enumerate(mine):
            want = np.array([1.7, -1.1, 0.6]) * (0.6 + 0.1 * t)
            # interior mean (the faces the shifted volume has left are cropped); consecutive time points differ by 0.17 in
            # x, so this identifies the time point; rank 0's first volume is checked bit for bit below
            assert np.abs(g["mean_flow"][q] - want).max() < 0.085, (t, g["mean_flow"][q], want)  # half the spacing
    fixed, moving, _ = fast_pair((n, n, n), shift=(1.7 * 0.6, -1.1 * 0.6, 0.6 * 0.6))
    single = fr.get_displacement(fixed, moving, **dict(SOLVER_DEFAULTS, levels=4)).astype(np.float32)
    assert np.array_equal(np.load(out + "_0.npz")["flow0"], single[::8, ::8, ::8])

