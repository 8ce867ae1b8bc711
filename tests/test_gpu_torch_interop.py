"""-m gpu: device pointers owned by PyTorch-ROCm (the bench's RCCL-broadcast reference lives in a
torch tensor) are usable by the engine's *_dev entry points -- both must share one HIP runtime."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_torch_tensor_pointers_feed_the_engine(hip, oracle):
    torch = pytest.importorskip("torch")
    from flowreg3d_amd import _lib
    lib = _lib.init(0)
    rng = np.random.default_rng(0)
    Z, Y, X = 10, 12, 14
    vol = rng.random((Z, Y, X, 1), dtype=np.float32)
    ref = rng.random((Z, Y, X, 1), dtype=np.float32)
    flow = ((rng.random((Z, Y, X, 3), dtype=np.float32) - 0.5) * 3).astype(np.float32)
    tv, tr, tf = (torch.from_numpy(a).to("cuda:0") for a in (vol, ref, flow))
    out = torch.empty((Z, Y, X, 1), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    _lib.check(lib.fr3d_warp_dev(tv.data_ptr(), _lib.F32, tf.data_ptr(), _lib.F32, tr.data_ptr(), Z, Y, X, 1, 3,
                                 out.data_ptr()))
    got = out.cpu().numpy()
    want = oracle.imregister_wrapper(vol, flow[..., 0], flow[..., 1], flow[..., 2], ref)
    assert np.abs(got[..., 0] - want).max() <= 1.2e-7


def test_process_batch_dev_on_torch_memory(hip, oracle):
    torch = pytest.importorskip("torch")
    from flowreg3d_amd import _lib
    from flowreg3d_amd.synthetic import make_pair
    lib = _lib.init(0)
    fixed, moving, _ = make_pair((12, 18, 20), seed=9, scale=0.4)
    kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=15, min_level=0, levels=2, eta=0.8, a_smooth=1.0, a_data=0.45)
    params = _lib.make_params(n_channels=1, **kw)
    T = 3
    tb = torch.from_numpy(np.stack([moving] * T)[..., None].copy()).to("cuda:0")
    tfix = torch.from_numpy(fixed[..., None].copy()).to("cuda:0")
    flows = torch.empty((T, 12, 18, 20, 3), dtype=torch.float32, device="cuda:0")
    regs = torch.empty((T, 12, 18, 20, 1), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    lib.fr3d_set_batch(2)  # exercises a full lock-step batch (2) and a remainder (1)
    try:
        _lib.check(lib.fr3d_process_batch_dev(C.byref(params), tb.data_ptr(), tb.data_ptr(), tfix.data_ptr(),
                                              tfix.data_ptr(), None, None, T, 12, 18, 20, 1, 3, flows.data_ptr(),
                                              regs.data_ptr(), C.cast(None, _lib.PROGRESS_FN), None))
    finally:
        lib.fr3d_set_batch(0)
    f = flows.cpu().numpy()
    want = oracle.get_displacement(fixed, moving, **kw)
    for t in range(T):  # identical volumes -> identical flows, whatever batch slot solved them
        assert np.array_equal(f[t], f[0])
    epe = np.linalg.norm(f[0] - want, axis=-1)
    assert epe.mean() < 1e-4, (epe.mean(), epe.max())
