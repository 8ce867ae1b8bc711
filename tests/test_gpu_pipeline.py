"""-m gpu: the batch driver -- preprocessing, w_init bootstrap and propagation,
executor, statistics -- against the outputs of the REFERENCE's own compensate_arr_3D (tests/golden/drv_*.npz) and,
for a larger series, against the driver loop restated on the CPU oracle (pinned to the same fixtures on the CPU)."""
import numpy as np
import pytest

from driver_cases import CASES, check_against_reference, load_case, oracle_driver

pytestmark = pytest.mark.gpu


def _video(T=6, shape=(10, 16, 18), C=1):
    from flowreg3d_amd.synthetic import make_pair
    fixed, _, _ = make_pair(shape, seed=31, channels=C)
    vols = [make_pair(shape, seed=31, channels=C, scale=0.15 * (t + 1))[1] for t in range(T)]
    v = np.stack(vols).astype(np.float32)
    if C == 1:
        v, fixed = v[..., None], fixed[..., None]
    return (v * 1000 + 100).astype(np.float32), (fixed * 1000 + 100).astype(np.float32)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["parity", "default"])
def test_compensate_arr_matches_reference_compensate_arr(hip, name, mode):
    """pipeline.compensate_arr_3D against the REFERENCE's own compensate_arr_3D outputs (tests/golden/drv_*.npz,
    tools/gen_driver_golden.py; sequential executor): serial and executor bootstrap of w_init, w_init rolled across
    batches, update_initialization_w=False, two channels with 1-D weights, uint16 series with output_typename, the
    4-D squeeze path, linear interpolation.  "parity" = fp64 solver storage at the reference's cross-executor
    tolerance (rtol 1e-5 / atol 1e-6 on `registered`); "default" = the library's automatic solver mode at the
    north-star flow bound."""
    import dataclasses
    from flowreg3d_amd.pipeline import compensate_arr_3D
    g, meta, opt = load_case(name)
    if mode == "parity":
        opt = dataclasses.replace(opt, solver_fp64=2)
    seen = []
    reg, w, stats = compensate_arr_3D(g["video"], g["reference"], opt, progress_callback=lambda a, b: seen.append((a, b)),
                                      return_stats=True)
    T = g["w"].shape[0]
    assert seen[-1] == (T, T) and [a for a, _ in seen] == sorted(a for a, _ in seen)
    assert list(g["progress"][-1]) == [T, T]
    check_against_reference(g, reg, w, stats, None, 1e-5 if mode == "parity" else 1e-4, f"{name}/{mode}",
                            parity_grade=mode == "parity")


@pytest.mark.parametrize("name", ["drv_t7_b5", "drv_t7_b3", "drv_noinit", "drv_c2_u16"])
def test_device_sink_driver_matches_reference_compensate_arr(hip, name):
    """the device-resident driver (run(..., sink="device"): series, flows, w_init means and statistics stay in HBM)
    against the same reference outputs, fp64 solver storage."""
    import dataclasses
    from driver_cases import OUT_DTYPES
    from flowreg3d_amd.pipeline import BatchMotionCorrectorHip
    g, meta, opt = load_case(name)
    bc = BatchMotionCorrectorHip(dataclasses.replace(opt, solver_fp64=2))
    sink = bc.run(g["video"], g["reference"], sink="device")
    try:
        reg, w = sink.registered(), sink.flows()
    finally:
        sink.free()
    if opt.output_typename in OUT_DTYPES:
        reg = reg.astype(OUT_DTYPES[opt.output_typename])
    check_against_reference(g, reg, w, bc.stats, bc.w_init, 1e-5, f"{name}/device sink")


def test_compensate_arr_matches_oracle_driver(hip, oracle):
    """a larger series than the reference fixtures afford (pure-Python reference: ~10 us per voxel update), against
    the oracle-driven loop that tests/test_driver_golden_cpu.py pins to the reference"""
    from flowreg3d_amd.pipeline import Options, compensate_arr_3D
    video, ref = _video()
    opt = Options(min_level=0, levels=3, iterations=12, update_lag=4, buffer_size=4, output_typename=None,
                  sigma=[[1.0, 0.8, 0.6, 0.5]], solver_fp64=2)
    seen = []
    reg, w, stats = compensate_arr_3D(video, ref, opt, progress_callback=lambda a, b: seen.append((a, b)),
                                      return_stats=True)
    assert reg.shape == video.shape and reg.dtype == video.dtype and w.shape == video.shape[:4] + (3,)
    assert seen[-1] == (video.shape[0], video.shape[0]) and [a for a, _ in seen] == sorted(a for a, _ in seen)
    reg_o, w_o, st_o, _ = oracle_driver(oracle, video, ref, opt)
    epe = np.linalg.norm(w.astype(np.float64) - w_o, axis=-1)
    assert epe.mean() < 1e-4, (epe.mean(), epe.max())
    assert np.abs(reg - reg_o).max() < 0.5  # intensities ~100..1100: 5e-4 relative
    for k in ("mean_disp", "max_disp", "mean_translation"):
        assert np.allclose(getattr(stats, k), st_o[k], rtol=1e-4, atol=1e-5), k
    assert np.allclose(stats.mean_div, st_o["mean_div"], rtol=1e-3, atol=2e-6)


def test_compensate_arr_shapes_and_dtypes(hip):
    from flowreg3d_amd.pipeline import Options, compensate_arr_3D
    video, ref = _video(T=3)
    opt = Options(min_level=0, levels=2, iterations=5, buffer_size=2)
    reg, w = compensate_arr_3D(video[..., 0], ref[..., 0], opt)          # (T,Z,Y,X) + 3-D reference
    assert reg.shape == video.shape[:4] and reg.dtype == np.float64 and w.shape == video.shape[:4] + (3,)
    reg1, w1 = compensate_arr_3D(video[0, ..., 0], ref[..., 0], Options(min_level=0, levels=2, iterations=5,
                                                                        output_typename="uint16"))
    assert reg1.shape == video.shape[1:4] and reg1.dtype == np.uint16 and w1.shape == video.shape[1:4] + (3,)
    with pytest.raises(ValueError):
        compensate_arr_3D(np.zeros((0, 4, 4, 4, 1)), ref, opt)


def test_flow_statistics_vs_numpy(hip):
    from flowreg3d_amd.pipeline import flow_statistics
    rng = np.random.default_rng(0)
    w = rng.standard_normal((2, 7, 9, 11, 3)).astype(np.float32)
    md, mx, dv, tr = flow_statistics(w)
    mag = np.sqrt(w[..., 0] ** 2 + w[..., 1] ** 2 + w[..., 2] ** 2)
    assert np.allclose(md, mag.mean(axis=(1, 2, 3)), rtol=1e-6)
    assert np.array_equal(mx.astype(np.float32), mag.max(axis=(1, 2, 3)))
    for t in range(2):
        div = np.gradient(w[t, ..., 0], axis=2) + np.gradient(w[t, ..., 1], axis=1) + np.gradient(w[t, ..., 2], axis=0)
        assert abs(dv[t] - float(div.mean())) < 1e-6


@pytest.mark.parametrize("dtype,update_ref", [(np.float32, False), (np.uint16, True)])
def test_device_sink_gives_the_host_path_results(hip, dtype, update_ref):
    """f-4: run(..., sink="device") keeps registered / w of the series in HBM (DeviceSink) and runs
    preprocessing, flow, warp, the w_init means, the statistics and update_reference on device-resident data:
    bit-identical to the host-array driver (same kernels, same order; np.mean(axis=0) of the float32 flows
    restated as float32 accumulation in stack order)."""
    from scipy.ndimage import gaussian_filter
    from flowreg3d_amd.pipeline import BatchMotionCorrectorHip, Options
    rng = np.random.default_rng(12)
    shape = (8, 18, 16)
    ref = gaussian_filter(rng.random(shape), 1.5)[..., None]
    ref = (ref - ref.min()) / (ref.max() - ref.min())
    video = np.stack([np.roll(ref, (t % 3) - 1, axis=2) * 0.9 + 0.02 * t for t in range(7)], 0)
    if dtype == np.uint16:
        video, ref = (video * 3000 + 100).astype(dtype), ref * 3000 + 100
    else:
        video = video.astype(dtype)
    kw = dict(levels=2, min_level=0, iterations=6, buffer_size=3, weight=[1.0], sigma=[[1.0, 1.0, 1.0, 0.1]],
              update_reference=update_ref)
    calls = []
    host = BatchMotionCorrectorHip(Options(**kw))
    reg_h, w_h = host.run(video, ref, sink="host_arrays")  # the array-at-a-time driver (custom executors, other dtypes)
    # the default host sink streams the batches through the device driver and fetches each batch's outputs
    stream = BatchMotionCorrectorHip(Options(**kw))
    reg_s, w_s = stream.run(video, ref)
    assert np.array_equal(w_s, w_h) and np.array_equal(reg_s, reg_h) and reg_s.dtype == dtype
    assert np.array_equal(stream.w_init, host.w_init)
    dev = BatchMotionCorrectorHip(Options(**kw))
    dev.register_progress_callback(lambda d, t: calls.append((d, t)))
    sink = dev.run(video, ref, sink="device")
    try:
        assert sink.filled == 7 and sink.nbytes == reg_h.nbytes + w_h.nbytes
        assert np.array_equal(sink.flows(), w_h)
        assert np.array_equal(sink.registered(), reg_h) and sink.registered().dtype == dtype
        assert np.array_equal(sink.flows(2, 5), w_h[2:5])
    finally:
        sink.free()
    assert np.array_equal(dev.w_init, host.w_init)
    assert calls and calls[-1] == (7, 7)
    for a, b in ((dev.stats.mean_disp, host.stats.mean_disp), (dev.stats.mean_div, host.stats.mean_div)):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-15)
    if update_ref:
        assert np.array_equal(dev.reference_proc, host.reference_proc)


def test_batch_driver_with_two_engine_lanes_is_bit_identical(hip):
    """Two engine lanes (the default) under the whole batch driver (host arrays and device sink): same registered series,
    flows, running w_init and statistics as with one lane; an invalid lane count leaves the setting alone."""
    from scipy.ndimage import gaussian_filter
    from flowreg3d_amd import _lib
    from flowreg3d_amd.pipeline import BatchMotionCorrectorHip, Options
    rng = np.random.default_rng(5)
    shape = (8, 16, 18)
    ref = gaussian_filter(rng.random(shape), 1.5)[..., None]
    ref = (ref - ref.min()) / (ref.max() - ref.min())
    video = np.stack([np.roll(ref, (t % 3) - 1, axis=1) * 0.95 + 0.01 * t for t in range(9)], 0).astype(np.float32)
    kw = dict(levels=2, min_level=0, iterations=6, buffer_size=5, weight=[1.0], sigma=[[1.0, 1.0, 1.0, 0.1]])
    lib = _lib.init(0)
    assert lib.fr3d_set_lanes(1) == 2
    try:
        one = BatchMotionCorrectorHip(Options(**kw))
        reg1, w1 = one.run(video, ref)
    finally:
        assert lib.fr3d_set_lanes(3) == 1 and lib.fr3d_set_lanes(2) == 1  # 3 is ignored
    try:
        two = BatchMotionCorrectorHip(Options(**kw))
        reg2, w2 = two.run(video, ref)
        dev = BatchMotionCorrectorHip(Options(**kw))
        sink = dev.run(video, ref, sink="device")
        try:
            assert np.array_equal(sink.flows(), w1) and np.array_equal(sink.registered(), reg1)
        finally:
            sink.free()
    finally:
        assert lib.fr3d_set_lanes(2) == 2
    assert np.array_equal(reg1, reg2) and np.array_equal(w1, w2)
    assert np.array_equal(one.w_init, two.w_init) and np.array_equal(one.w_init, dev.w_init)
    np.testing.assert_array_equal(one.stats.mean_disp, two.stats.mean_disp)
