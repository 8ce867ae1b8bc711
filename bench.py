#!/usr/bin/env python3
"""Benchmark of the hot path: volumes/sec of (3-D flow solve + final warp) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg1]

One "step" = one volume of a synthetic time series registered against a fixed reference:
get_displacement (5-level pyramid at 256^3) + cubic compensation warp, i.e. one iteration of the
executor body (flowreg3d motion_correction/parallelization/sequential_3d.py:148-160), through the
C ABI entry fr3d_process_batch_dev with every input already resident in HBM.  With N > 1 (launched
by torch.distributed.run, one rank per GPU) the series is sharded volume-per-GPU, the fixed
reference is broadcast once from rank 0 over RCCL and nothing else is exchanged ("scaling": weak).

Prints ONE JSON line on rank 0 (schema in the task contract) with `roofline` for the SOR sweep
kernel (HIP events on the engine's stream) and `cpu_baseline` (the C oracle on a bounded sample, one
core and all cores).  At N = 1 the default run adds two legs to the same line, after the timed region of
the headline workload: `host_path` (NumPy arrays in, NumPy arrays out through fr3d_process_batch: the
PCIe-inclusive rate of the drop-in entry, never `value`), `pipeline` (the whole drop-in driver compensate_arr_3D:
preprocessing, w_init bootstrap, executor, statistics -- host arrays and device sink), `single_pair` (one
get_displacement call, the reference's API unit, at 256^3 and 512^3), `cfg3` (the 512^3 six-level configuration on
SURVEY 8d's input recipe, 4 timed steps at lock-step batch 4, with its own `roofline`), `cfg5` (two channels) and
`a_smooth_0.5` (the psi_smooth solver path on the cfg2 geometry); `--no-extras` skips them.
`roofline.frac` prices the sweep on SURVEY 8d's contract bytes (4 B per value: 76 B per voxel update for one channel),
whatever the storage format; `frac_on_storage_basis` prices the same time on the mode's own format.  The timed region runs the library's default of two engine lanes; `one_lane`
is the profiled one-lane pass right after it, on which `roofline` is measured (--lanes 1: the timed region itself).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (Z, Y, X, levels, description)      "N-level pyramid" => levels = N-1, min_level = 0
    "cfg1": (32, 64, 64, 2, "64x64x32 pair, 3-level pyramid"),
    "cfg2": (256, 256, 256, 4, "256^3 single-channel fp32, 5-level pyramid"),
    "cfg3": (512, 512, 512, 5, "512^3 single-channel fp32, 6-level pyramid"),
    # BASELINE config 5's geometry with the 13-solve schedule its parity fixture uses (tests/golden/fullsize_cfg5.npz)
    "cfg5": (256, 512, 512, 12, "256x512x512 two-channel fp32, 13-level pyramid"),
}
CHANNELS = {"cfg5": 2}
# cfg4 = cfg2 on a 64-timepoint series sharded over N GPUs: `--workload cfg2 --gpus N --steps 8`.
# cfg5 (512x512x256, two channels) is a parity case (tests/test_gpu_fullsize_parity.py); the default run times it as an extra leg.
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


_REF_CACHE = {}  # (Z, Y, X, recipe) -> reference volume: the 512^3 texture costs half a minute of host time, once per process


def solver_kwargs(levels, a_smooth=1.0):
    return dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=100, min_level=0, levels=levels,
                eta=0.8, a_smooth=a_smooth, a_data=0.45)


class DevArray:
    """A float32 device buffer owned through the engine's allocator."""

    def __init__(self, lib, shape):
        self.lib, self.shape = lib, tuple(shape)
        self.nbytes = int(np.prod(shape)) * 4
        self.ptr = lib.fr3d_dev_malloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("fr3d_dev_malloc failed")

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.nbytes == self.nbytes
        rc = self.lib.fr3d_h2d(self.ptr, a.ctypes.data, self.nbytes)
        assert rc == 0
        return self

    def download(self):
        out = np.empty(self.shape, np.float32)
        rc = self.lib.fr3d_d2h(out.ctypes.data, self.ptr, self.nbytes)
        assert rc == 0
        return out

    def free(self):
        if self.ptr:
            self.lib.fr3d_dev_free(self.ptr)
            self.ptr = None


def _cpu_volume(args):
    """one volume of the CPU path (flow solve + compensation warp) on one core; -> seconds"""
    shape, levels = args
    from oracle import oracle
    from flowreg3d_amd.synthetic import fast_pair
    fixed, moving, _ = fast_pair(shape)
    kw = solver_kwargs(levels)
    t0 = time.perf_counter()
    flow = oracle.get_displacement(fixed, moving, **kw).astype(np.float32)
    oracle.imregister_wrapper(moving, flow[..., 0], flow[..., 1], flow[..., 2], fixed)
    return time.perf_counter() - t0


def _cpu_share():
    """CPU cores this process may really use: affinity mask, cgroup quota, FR3D_CPU_WORKERS; at most 16
    worker processes unless told otherwise (the CPU share of a one-GPU box of the pool)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("FR3D_CPU_WORKERS")
    return max(1, int(env)) if env else min(n, 16)


def cpu_baseline(workload, sample_edge):
    """The CPU oracle (restatement of the reference's NumPy/Numba path; kind "port") timed on this host:
    one volume on one core, then one volume per core on every core at once (one process per volume,
    as the reference's MultiprocessingExecutor3D runs it, parallelization/multiprocessing_3d.py:309-318).
    The sample is a cube of `sample_edge` voxels with the workload's pyramid and solver parameters (the
    oracle needs 3.5-7 minutes for one 256^3 volume, tests/golden/fullsize_cfg2.npz metadata); `value`
    scales the all-core sample rate to the workload's voxel count."""
    from oracle import oracle
    oracle.build()
    Z, Y, X, levels, _ = WORKLOADS[workload]
    e = min(sample_edge, Z, Y, X)
    shape = (e, e, e) if workload != "cfg1" else (Z, Y, X)
    scale = (shape[0] * shape[1] * shape[2]) / float(Z * Y * X)
    t1 = _cpu_volume((shape, levels))
    cores = _cpu_share()
    # memory: the oracle holds ~300 B per voxel of the finest level; keep the concurrent volumes inside half the RAM
    try:
        ram = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
        cores = max(1, min(cores, int(0.5 * ram / (320.0 * shape[0] * shape[1] * shape[2]))))
    except (ValueError, OSError):
        pass
    # one child interpreter per volume (started before this process touches the GPU, see main()); each prints
    # the seconds its volume took
    code = f"import bench; print(bench._cpu_volume((({shape[0]}, {shape[1]}, {shape[2]}), {levels})))"
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, text=True)
             for _ in range(cores)]
    per = []
    for pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("CPU baseline worker failed")
        per.append(float(out.strip().splitlines()[-1]))
    t_all = time.perf_counter() - t0
    rate_all = cores / t_all
    return {"value": rate_all * scale, "unit": "volumes/sec", "cores": cores, "kind": "port",
            "sample": f"{shape[0]}x{shape[1]}x{shape[2]} pair, same pyramid/solver parameters: 1 volume on 1 core {t1:.1f} s; "
                      f"{cores} volumes on {cores} cores (one process per volume) {t_all:.1f} s wall, "
                      f"{min(per):.1f}-{max(per):.1f} s each; rates scaled by voxel count to {Z}x{Y}x{X}",
            "value_1core": scale / t1, "sample_volumes_per_sec_1core": 1.0 / t1,
            "fullsize_1core_measured": "one whole 256^3 volume (flow solve only) measured 206 s on one core of a GPU host "
                                       "(round 1) and 414-460 s in the 8-core build container (tests/golden/fullsize_cfg2*.npz "
                                       "metadata): the voxel-count scaling of the sample flatters the CPU by ~15 %",
            "fullsize_oracle_seconds_1core": fullsize_cpu_seconds(),
            "sample_volumes_per_sec_all_cores": rate_all, "host_cpus": os.cpu_count()}


def fullsize_cpu_seconds():
    """Seconds one core needed for ONE whole volume's flow solve at full size: the CPU-oracle runs that produced the
    parity fixtures (tests/golden/fullsize_*.npz metadata; 8-core build container, one core per run).  Measured, not
    scaled: 512^3 costs ~2700-3700 s, i.e. 2.7e-4 .. 3.7e-4 volumes/s on one core."""
    out = {}
    for case in ("cfg2_recipe", "cfg3", "cfg3_recipe", "cfg5"):
        try:
            z = np.load(os.path.join(ROOT, "tests", "golden", f"fullsize_{case}.npz"))
            meta = json.loads(bytes(z["meta"]).decode())
            out[case] = {"shape_zyx": meta["shape_zyx"], "channels": meta["channels"],
                         "seconds": round(float(meta["oracle_seconds_1core"]), 1)}
        except (OSError, KeyError, ValueError):
            continue
    return out


SOLVER_NAMES = ("fp32 storage, fp32 update arithmetic", "fp32 storage, fp64 update arithmetic",
                "fp64 storage and arithmetic", "packed 42-bit storage (three values per 16 B), fp64 update arithmetic")
STORAGE_BYTES = (4.0, 4.0, 8.0, 16.0 / 3.0)  # per stored solver value


def parity_record(workload, mode, recipe_inputs):
    """Mean flow end-point error against the CPU path at full size as MEASURED by tests/test_gpu_fullsize_parity.py
    (which appends to gpurun_out/parity_fullsize.json; the committed copy is profiles/parity_fullsize.json).  A property
    of (inputs, solver mode), not re-measured by a bench run: the CPU side of one 256^3 volume takes minutes.
    -> (value or None, where it comes from)"""
    try:
        with open(os.path.join(ROOT, "profiles", "parity_fullsize.json")) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None, "profiles/parity_fullsize.json missing"
    for case in ((workload + "_recipe", workload) if recipe_inputs else (workload, workload + "_recipe")):
        e = rec.get(f"{case}/mode{mode}")
        if e and workload == "cfg5":
            return e["lattice_mean_epe"], (f"profiles/parity_fullsize.json[{case}/mode{mode}]: lattice mean EPE vs the CPU oracle on the "
                                           "expansion + rotation pair of tests/golden/fullsize_cfg5.npz; this run TIMES a "
                                           "translated stand-in texture of the same geometry")
        if e:
            same = case.endswith("_recipe") == bool(recipe_inputs)
            return e["lattice_mean_epe"], (f"profiles/parity_fullsize.json[{case}/mode{mode}]: lattice mean EPE vs the CPU oracle, "
                                           + ("same input recipe as this run" if same else "NOT this run's input recipe"))
    return None, f"no full-size measurement recorded for {workload} in solver mode {mode}"


def sweep_source_hash():
    """identifies the sweep kernel the PMC traffic record belongs to (sources, not the binary: rebuilding unrelated
    kernels must not invalidate it)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("k_sor.hip", "k_sor_core.h", "fr3d_internal.h"):
        with open(os.path.join(ROOT, "flowreg3d_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def resolved_mode(solver_fp64, nvox, channels=1, a_smooth=1.0):
    """FR3D_SOLVER_AUTO as the engine resolves it (flowreg3d_amd/csrc/engine.hip: solver_mode)."""
    m = solver_fp64
    if m < 0 and a_smooth != 1.0:
        m = 2 if (channels >= 2 or nvox > (1 << 25)) else 1
    if m < 0:
        m = 2 if channels >= 2 else (3 if nvox > (1 << 22) else 1)
    return 2 if (m == 3 and a_smooth != 1.0) else m


def measure(lib, _lib, workload, K, W, batch_arg, condition, solver_fp64, rank, world, dist, dev_index, fast_inputs,
            a_smooth=1.0, lanes=2):
    """Warm up, condition, time EXACTLY K steps (volumes per rank) of `workload`; -> dict of results.
    Inputs are generated once and are resident in HBM before the timed region starts.
    lanes = 2 (the library's default): the timed region runs two engine lanes (two lock-step half batches in flight on two
    HIP streams); the per-kernel HIP-event times -- `roofline`, `kernel_ms_per_step` -- are then taken on a profiled pass
    of the same K steps directly after it, which the library runs on ONE lane (the event spans of two lanes overlap and
    are not kernel times).  lanes = 1: the timed region itself is the profiled pass."""
    lib.fr3d_set_lanes(lanes)
    from flowreg3d_amd.synthetic import fast_pair, flow_gt, texture
    Z, Y, X, levels, desc = WORKLOADS[workload]
    nch = CHANNELS.get(workload, 1)
    nv = Z * Y * X
    T = K + W
    params = _lib.make_params(n_channels=nch, solver_fp64=None if solver_fp64 < 0 else solver_fp64,
                              **solver_kwargs(levels, a_smooth))
    mode = resolved_mode(solver_fp64, nv, nch, a_smooth)

    def reference_volume():
        # texture(): blurred noise + blobs (SURVEY 8d's recipe, the inputs the parity records were measured on);
        # fast_inputs: fast_pair's O(N) stand-in
        key = (Z, Y, X, nch, not fast_inputs)
        if key not in _REF_CACHE:
            chans = [fast_pair((Z, Y, X), seed=1234 + c)[0] if fast_inputs else texture((Z, Y, X), seed=1234 + c) for c in range(nch)]
            _REF_CACHE[key] = chans[0] if nch == 1 else np.ascontiguousarray(np.stack(chans, -1))
        return _REF_CACHE[key]

    # ---- fixed reference: generated on rank 0, broadcast over RCCL/xGMI -----------------------
    ref_dev = DevArray(lib, (Z, Y, X, nch))
    ref_t = None
    bcast_ms = None
    if world > 1:
        import torch
        ref_t = torch.empty((Z, Y, X, nch), dtype=torch.float32, device=f"cuda:{dev_index}")
        if rank == 0:
            ref_t.copy_(torch.from_numpy(reference_volume().reshape(Z, Y, X, nch)))
        torch.cuda.synchronize()
        dist.barrier()
        t_b = time.perf_counter()
        dist.broadcast(ref_t, src=0)  # the path's only collective
        torch.cuda.synchronize()      # the engine reads the buffer on its own stream: complete before any engine call
        bcast_ms = 1e3 * (time.perf_counter() - t_b)
        fixed_ptr = ref_t.data_ptr()
    else:
        ref_dev.upload(reference_volume())
        fixed_ptr = ref_dev.ptr

    # ---- this rank's shard of the time series: moving_t = warp(fixed, -flow_gt * s_t) on the GPU --
    batch = DevArray(lib, (T, Z, Y, X, nch))
    flows = DevArray(lib, (T, Z, Y, X, 3))
    regs = DevArray(lib, (T, Z, Y, X, nch))
    gflow = DevArray(lib, (Z, Y, X, 3))
    for i in range(T):
        t_global = rank + world * i
        s = np.sin(2.0 * np.pi * (t_global + 1) / 64.0) + 0.35
        gflow.upload(-flow_gt((Z, Y, X), scale=float(s)))
        _lib.check(lib.fr3d_warp_dev(fixed_ptr, _lib.F32, gflow.ptr, _lib.F32, fixed_ptr, Z, Y, X, nch, 3,
                                     batch.ptr + i * nv * 4 * nch))
    gflow.free()

    # default lock-step batch: 8 at 256^3, 4 at 512^3 (the compact solver slabs of a 512^3 volume take 16 GB with
    # fp32 storage, 33 GB with fp64 storage; batch 8 fits too and runs at the same rate per volume)
    batch_vols = max(1, min(K, batch_arg if batch_arg > 0 else (4 if nv > (1 << 25) else 8)))
    lib.fr3d_set_batch(batch_vols)  # warm-up and timed run use the same lock-step batch / workspace

    def run(first, count, prof):
        lib.fr3d_prof_enable(1 if prof else 0)
        if prof:
            lib.fr3d_prof_reset()
        _lib.check(lib.fr3d_process_batch_dev(
            C.byref(params), batch.ptr + first * nv * 4 * nch, batch.ptr + first * nv * 4 * nch, fixed_ptr, fixed_ptr,
            None, None, count, Z, Y, X, nch, 3, flows.ptr + first * nv * 12, regs.ptr + first * nv * 4 * nch,
            C.cast(None, _lib.PROGRESS_FN), None))

    def barrier():
        lib.fr3d_sync()
        if world > 1:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    if W > 0:
        # warm-up runs at least one full lock-step batch so that every workspace slab the timed
        # steps use has been touched (fresh device memory is slower on first use); the volumes it
        # borrows from the timed range are recomputed inside the timed region
        run(0, max(W, batch_vols), False)
        # An idle MI355X needs 15 s and more of load before its memory system runs at full rate (the SOR
        # kernel measures 3.5 TB/s in the first seconds of a fresh box and 3.85 TB/s from then on,
        # whatever the binary): keep repeating the warm-up batch, untimed, for --condition seconds
        # so that the timed steps see the steady state a long series runs in.
        t_c = time.perf_counter()
        while condition > 0 and time.perf_counter() - t_c < condition:
            run(0, max(W, batch_vols), False)
            lib.fr3d_sync()
    barrier()
    t0 = time.perf_counter()
    run(W, K, lanes == 1)
    barrier()
    elapsed = time.perf_counter() - t0
    one_lane = None
    if lanes != 1:
        run(W, K, True)  # the one-lane workspace (twice the lock-step batch per lane) is allocated here, untimed
        lib.fr3d_sync()
        t1 = time.perf_counter()
        run(W, K, True)  # run() resets the brackets: the statistics below are this pass alone
        lib.fr3d_sync()
        serial = time.perf_counter() - t1
        one_lane = {"value": K / serial, "unit": "volumes/sec per GPU", "ms_per_step": 1e3 * serial / K,
                    "what": "the same K steps on one engine lane with profiling brackets, directly after the timed region: "
                            "the pass `roofline`, `kernel_ms_per_step` and `roofline_stages` are measured on"}
    stats = _lib.prof_get()
    lib.fr3d_prof_enable(0)
    ran = int(lib.fr3d_last_solver_mode())  # what the library actually resolved FR3D_SOLVER_AUTO to
    if ran != mode:
        print(f"note: solver mode {ran} ran where bench.py expected {mode} (memory-driven fallback?)", file=sys.stderr)
        mode = ran
    per_rank = None
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, float(elapsed))  # timing only: every rank's own K-step time (the step time is their maximum)
        per_rank = [float(x) for x in every]
        elapsed = max(per_rank)
    par, par_src = parity_record(workload, mode, not fast_inputs)
    res = {"elapsed": elapsed, "stats": stats, "batch_vols": batch_vols, "desc": desc, "mode": mode, "lanes": lanes,
           "one_lane": one_lane,
           "parity_mean_epe_vs_cpu": par, "parity_source": par_src,
           "per_rank_volumes_per_sec": None if per_rank is None else [round(K / t, 3) for t in per_rank],
           "broadcast": None if bcast_ms is None else {"bytes": nv * 4, "ms": round(bcast_ms, 3),
                                                        "what": "fixed reference volume, rank 0 -> all (RCCL over xGMI)"}}
    if rank == 0:
        sor = stats["sor"]
        achieved = sor["algo_bytes"] / (sor["ms"] * 1e-3) / 1e9 if sor["ms"] > 0 else 0.0
        # HBM traffic of the SOR kernel: separate rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950
        # correction + WRITE_SIZE), stored per voxel update in profiles/pmc_traffic.json by
        # tools/make_pmc_traffic.py -- a constant of the committed kernel, NOT measured by this run
        traffic, traffic_source = None, "no PMC record for this workload / solver mode"
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pj = json.load(fh)
            pmc = pj.get(f"{workload}/mode{mode}") if a_smooth == 1.0 else None  # (single-channel records; none for cfg5)
            if pmc and pj.get("_sweep_source_hash") != sweep_source_hash():
                traffic_source = ("profiles/pmc_traffic.json was taken on other sweep-kernel sources "
                                  f"({pj.get('_sweep_source_hash')} != {sweep_source_hash()}): not quoted")
            elif pmc:
                traffic = pmc["bytes_per_update"] * sor["units"] / max(sor["launches"], 1)
                traffic_source = f"profiles/pmc_traffic.json ({pj.get('_taken_at', 'unknown round')}; same sweep-kernel " \
                                 f"sources as this run): {pmc['bytes_per_update']:.1f} B per voxel update x this run's " \
                                 "updates per launch -- separate rocprofv3 --pmc passes, not measured by this run"
        except (OSError, ValueError, KeyError):
            traffic = None
        channels = nch
        storage_basis = sor["algo_bytes"] / max(sor["units"], 1)
        vals = 10 * channels + (9 if a_smooth == 1.0 else 17)
        # SURVEY 8d prices the sweep at 4 B per value (76 B per voxel update for C = 1, a_smooth = 1) whatever the storage
        # format: `frac` is on that basis, so that rounds and storage formats compare; the storage-format figure is beside it
        contract_basis = 4.0 * vals
        secs = sor["ms"] * 1e-3
        achieved = contract_basis * sor["units"] / secs / 1e9 if secs > 0 else 0.0
        per_launch = sor["units"] / max(sor["launches"], 1)
        res["roofline"] = {"bound": "hbm",
                           "kernel": "k_sor_step (SOR hyperplane sweep)" if a_smooth == 1.0 else
                                     "k_smooth_psi_only + k_smooth_sweep_only (psi_smooth SOR sweep, a_smooth != 1: three launches per step, counted as one)",
                           "algo_bytes_per_update": contract_basis,
                           "basis": f"SURVEY 8d: {vals} values per voxel update ({'9 J + w psi + 3 L + 3 d read, 3 d written' if a_smooth == 1.0 else '9 J + w psi + 3 u + 3 d + psi_s read, 3 d written; psi_s: 3 u + 3 d read, 1 written'}) "
                                    "x 4 B (the fp32 figure of the contract, independent of this mode's storage format)",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS,
                           "storage_bytes_per_update": storage_basis,
                           "storage_basis": f"{vals} values x {STORAGE_BYTES[mode]:.3g} B per value in this mode's storage format",
                           "achieved_on_storage_basis": storage_basis * sor["units"] / secs / 1e9 if secs > 0 else 0.0,
                           "frac_on_storage_basis": (storage_basis * sor["units"] / secs / 1e9 / HBM_PEAK_GBS) if secs > 0 else 0.0,
                           "traffic": traffic, "traffic_source": traffic_source,
                           "traffic_over_algorithmic": None if traffic is None else traffic / (contract_basis * per_launch),
                           "algo_bytes_per_launch": contract_basis * per_launch,
                           "avg_launch_us": 1e3 * sor["ms"] / max(sor["launches"], 1),
                           "launches": sor["launches"],
                           "sor_ms_per_step": sor["ms"] / K,
                           "measured_on": "the timed region (one engine lane)" if lanes == 1 else
                                          "a profiled one-lane pass of the same K steps directly after the timed region (see `one_lane`: "
                                          "sor_ms_per_step is a part of one_lane.ms_per_step, not of the two-lane ms_per_step); "
                                          "the timed region runs two lanes, whose kernels overlap in time"}
        res["kernel_ms_per_step"] = {k: round(v["ms"] / K, 3) for k, v in stats.items()}
        # the other stages of the path against the same HBM roofline, algorithmic bytes as in
        # DESIGN.md section 5 (warp: 24 B/voxel; the median is compute-bound and listed for completeness)
        res["roofline_stages"] = {k: {"achieved": round(v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "unit": "GB/s",
                                      "frac": round(v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                  for k, v in stats.items()
                                  if k in ("warp", "prefilter", "tensor", "resize", "median", "preproc") and v["ms"] > 0}
    for a in (batch, flows, regs, ref_dev):
        a.free()
    del ref_t
    return res


def host_path(workload, n_vol, solver_fp64):
    """The drop-in entry as the reference calls it: NumPy arrays in, NumPy arrays out through
    HipExecutor3D.process_batch -> fr3d_process_batch_raw (pageable host memory, PCIe both ways)."""
    from flowreg3d_amd.executor import HipExecutor3D
    from flowreg3d_amd.synthetic import fast_pair
    Z, Y, X, levels, _ = WORKLOADS[workload]
    fixed, moving, _ = fast_pair((Z, Y, X))
    batch = np.ascontiguousarray(np.stack([moving] * n_vol)[..., None])
    fp = dict(solver_kwargs(levels), weight=np.array([1.0]), solver_fp64=solver_fp64)
    w0 = np.zeros((Z, Y, X, 3), np.float32)
    ref = fixed[..., None]
    best = None
    with HipExecutor3D() as ex:
        reg = flows = None
        for _ in range(2):  # first call allocates staging buffers and faults in the output arrays
            del reg, flows  # the previous call's 2 GB of results are released outside the timed call (munmap: ~0.1 s)
            t0 = time.perf_counter()
            out = ex.process_batch(batch, batch, ref, ref, w0, None, None, flow_params=fp)
            dt = time.perf_counter() - t0
            reg, flows = out
            del out
            best = dt if best is None else min(best, dt)
    return {"value": n_vol / best, "unit": "volumes/sec", "volumes": n_vol,
            "what": f"{workload}: HipExecutor3D.process_batch on NumPy arrays ({batch.nbytes >> 20} MiB in x2, "
                    f"{(reg.nbytes + flows.nbytes) >> 20} MiB out, pageable memory), second of two calls"}


def make_series(lib, _lib, shape, n_vol):
    """n_vol moving volumes of the synthetic series (device warp of SURVEY 8d's texture) as one host array, + the reference"""
    from flowreg3d_amd.synthetic import flow_gt, texture
    Z, Y, X = shape
    nv = Z * Y * X
    key = (Z, Y, X, 1, True)
    if key not in _REF_CACHE:
        _REF_CACHE[key] = texture((Z, Y, X), seed=1234)
    ref = _REF_CACHE[key]
    ref_dev = DevArray(lib, (Z, Y, X, 1)).upload(ref)
    gflow = DevArray(lib, (Z, Y, X, 3))
    one = DevArray(lib, (Z, Y, X, 1))
    series = np.empty((n_vol, Z, Y, X, 1), np.float32)
    for t in range(n_vol):
        s = np.sin(2.0 * np.pi * (t + 1) / 64.0) + 0.35
        gflow.upload(-flow_gt((Z, Y, X), scale=float(s)))
        _lib.check(lib.fr3d_warp_dev(ref_dev.ptr, _lib.F32, gflow.ptr, _lib.F32, ref_dev.ptr, Z, Y, X, 1, 3, one.ptr))
        series[t] = one.download()
    for a in (ref_dev, gflow, one):
        a.free()
    return series, ref


def pipeline_leg(lib, _lib, n_vol=16, buffer_size=8):
    """The whole drop-in driver (flowreg3d_amd.pipeline.compensate_arr_3D: the reference's compensate_arr_3D ->
    BatchMotionCorrector.run, motion_correction/compensate_recording_3D.py:229-254,431-555): preprocessing (normalise +
    Gaussian, fp64), the w_init bootstrap (the first batch is solved twice: from zero, then from the mean flow), the
    executor, the w_init roll, statistics.  Host arrays in and out, then the device-resident sink."""
    from flowreg3d_amd.pipeline import BatchMotionCorrectorHip, Options, compensate_arr_3D
    Z, Y, X, levels, _ = WORKLOADS["cfg2"]
    series, ref = make_series(lib, _lib, (Z, Y, X), n_vol)
    opt = Options(alpha=(0.25, 0.25, 0.25), weight=[1.0], levels=levels, min_level=0, eta=0.8, update_lag=5, iterations=100,
                  a_smooth=1.0, a_data=0.45, buffer_size=buffer_size, output_typename="single")
    solves = n_vol + min(22, buffer_size, n_vol)  # every volume once + the bootstrap pass over the first batch
    lib.fr3d_prof_enable(0)
    t0 = time.perf_counter()
    reg, w = compensate_arr_3D(series, ref[..., None], opt)
    t_host = time.perf_counter() - t0
    del reg, w
    # device sink, with the stage brackets on (one lane) for the preprocessing kernel's share
    lib.fr3d_prof_enable(1)
    lib.fr3d_prof_reset()
    t0 = time.perf_counter()
    sink = BatchMotionCorrectorHip(opt).run(series, ref[..., None], sink="device")
    lib.fr3d_sync()
    t_dev_prof = time.perf_counter() - t0
    stats = _lib.prof_get()
    lib.fr3d_prof_enable(0)
    sink.free()
    t0 = time.perf_counter()
    sink = BatchMotionCorrectorHip(opt).run(series, ref[..., None], sink="device")
    lib.fr3d_sync()
    t_dev = time.perf_counter() - t0
    sink.free()
    pre = stats["preproc"]
    nvox = float(Z) * Y * X
    # per volume the reference's preprocessing reads the raw volume and writes the processed one once per separable
    # pass; algorithmic bytes as for the resampler: (N_in + N_out) values per pass, fp64 between the passes
    return {"workload": f"cfg2 geometry, {n_vol} volumes of the synthetic series, buffer_size {buffer_size}: "
                        f"{solves} flow solves (bootstrap pass over the first batch + every volume)",
            "value": n_vol / t_host, "unit": "volumes/sec", "seconds": t_host,
            "solves_per_sec": solves / t_host,
            "what": "compensate_arr_3D on NumPy arrays (float32 series in, float32 registered + flows out; pageable memory)",
            "device_sink": {"value": n_vol / t_dev, "unit": "volumes/sec", "seconds": t_dev, "solves_per_sec": solves / t_dev,
                            "what": "BatchMotionCorrectorHip.run(sink='device'): batches uploaded once, everything else resident in HBM",
                            "seconds_with_stage_brackets_one_lane": t_dev_prof},
            "preproc_ms_per_volume": pre["ms"] / max(n_vol + 1, 1),
            "preproc": {"ms": pre["ms"], "launches": pre["launches"], "volumes": n_vol + 1,
                        "achieved": round(pre["algo_bytes"] / (pre["ms"] * 1e-3) / 1e9, 1) if pre["ms"] > 0 else None,
                        "unit": "GB/s", "frac": round(pre["algo_bytes"] / (pre["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pre["ms"] > 0 else None,
                        "basis": "per separable pass one read and one write of the volume in the pass's own types "
                                 f"({nvox:.0f} voxels per volume; the reference volume is preprocessed once too)"},
            "kernel_ms_per_volume_device_sink": {k: round(v["ms"] / n_vol, 3) for k, v in stats.items()}}


def single_pair_leg(lib, _lib, workload, solver_fp64):
    """One get_displacement call -- the reference's API unit (core/optical_flow_3d.py:319) -- on device-resident
    volumes, batch of one; and the time to the FIRST result of a lock-step batch beside it (256^3)."""
    from flowreg3d_amd.synthetic import flow_gt, texture
    Z, Y, X, levels, desc = WORKLOADS[workload]
    nv = Z * Y * X
    key = (Z, Y, X, 1, True)
    if key not in _REF_CACHE:
        _REF_CACHE[key] = texture((Z, Y, X), seed=1234)
    fixed = DevArray(lib, (Z, Y, X, 1)).upload(_REF_CACHE[key])
    gflow = DevArray(lib, (Z, Y, X, 3)).upload(-flow_gt((Z, Y, X)))
    moving = DevArray(lib, (Z, Y, X, 1))
    flow = DevArray(lib, (Z, Y, X, 3))
    _lib.check(lib.fr3d_warp_dev(fixed.ptr, _lib.F32, gflow.ptr, _lib.F32, fixed.ptr, Z, Y, X, 1, 3, moving.ptr))
    params = _lib.make_params(n_channels=1, solver_fp64=None if solver_fp64 < 0 else solver_fp64, **solver_kwargs(levels))
    times = []
    for _ in range(4):  # the first call allocates the workspace
        lib.fr3d_sync()
        t0 = time.perf_counter()
        _lib.check(lib.fr3d_get_displacement_dev(C.byref(params), fixed.ptr, moving.ptr, Z, Y, X, 1, None, None, flow.ptr))
        lib.fr3d_sync()
        times.append(time.perf_counter() - t0)
    out = {"workload": f"{workload}: {desc}; one fr3d_get_displacement_dev call (flow solve only, no final warp)",
           "seconds": min(times[1:]), "value": 1.0 / min(times[1:]), "unit": "volumes/sec",
           "solver": SOLVER_NAMES[int(lib.fr3d_last_solver_mode())], "calls_s": [round(t, 4) for t in times]}
    if nv <= (1 << 24):
        # latency of the first result when 8 volumes are solved in lock step on one lane (fr3d_set_batch lowers it)
        n = 8
        batch = DevArray(lib, (n, Z, Y, X, 1))
        for t in range(n):
            _lib.check(lib.fr3d_warp_dev(fixed.ptr, _lib.F32, gflow.ptr, _lib.F32, fixed.ptr, Z, Y, X, 1, 3, batch.ptr + t * nv * 4))
        flows = DevArray(lib, (n, Z, Y, X, 3))
        regs = DevArray(lib, (n, Z, Y, X, 1))
        stamps = []
        cb = _lib.PROGRESS_FN(lambda k, _u: stamps.append(time.perf_counter()))
        lat = {}
        for lanes in (1, 2):
            lib.fr3d_set_lanes(lanes)
            lib.fr3d_set_batch(n)
            for rep in range(2):
                del stamps[:]
                t0 = time.perf_counter()
                _lib.check(lib.fr3d_process_batch_dev(C.byref(params), batch.ptr, batch.ptr, fixed.ptr, fixed.ptr, None, None, n,
                                                      Z, Y, X, 1, 3, flows.ptr, regs.ptr, cb, None))
                lib.fr3d_sync()
                t_all = time.perf_counter() - t0
            lat[f"lanes{lanes}"] = {"first_result_s": round(stamps[0] - t0, 4) if stamps else None, "all_8_s": round(t_all, 4)}
        lib.fr3d_set_lanes(2)
        out["lockstep_batch_of_8"] = lat
        for a in (batch, flows, regs):
            a.free()
    for a in (fixed, gflow, moving, flow):
        a.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=128, help="edge of the CPU-baseline cube")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs of the default run (host-array path, the 512^3 line)")
    ap.add_argument("--solver-fp64", type=int, default=-1, choices=(-1, 0, 1, 2, 3),
                    help="-1 = the library's choice (FR3D_SOLVER_AUTO: for one channel fp32 storage with fp64 update arithmetic up "
                         "to 2^22 voxels, packed 42-bit storage above -- the cheapest mode measured to stay within 1e-4 voxels of "
                         "the CPU path with margin); 0 fp32 storage+update, 1 fp32 storage with fp64 update arithmetic, 2 fp64 "
                         "storage, 3 packed 42-bit storage with fp64 update arithmetic")
    ap.add_argument("--batch", type=int, default=0,
                    help="volumes solved in lock step per GPU (shared launches); 0 = 8 at 256^3, 4 at 512^3")
    ap.add_argument("--a-smooth", type=float, default=1.0,
                    help="smoothness exponent (1.0 = the pipeline's OFOptions default and every BASELINE configuration; any "
                         "other value, e.g. get_displacement's own default 0.5, runs the psi_smooth solver path)")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2),
                    help="engine lanes of the timed region (fr3d_set_lanes; 2 = the library's default).  Per-kernel times are "
                         "always taken on one lane: with 2 on a profiled pass directly after the timed region")
    ap.add_argument("--condition", type=float, default=30.0,
                    help="seconds of untimed warm-up work before the timed steps (0 = only the W warm-up steps)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # The CPU baseline runs FIRST, before this process touches the GPU: its all-core leg starts one worker
    # process per core, and a process that has initialised HIP must not spawn/exec (rank 0 at N = 1 only).
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, args.cpu_sample)

    dist = None
    dev_index = local_rank
    backend = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  FR3D_DIST_BACKEND=gloo lets the N>1 path be rehearsed on a
        # one-GPU box (ranks then share device 0); it is never used for reported numbers.
        backend = os.environ.get("FR3D_DIST_BACKEND", "nccl")
        dev_index = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()
    if world > 1 and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match the process group's world size {world}")

    from flowreg3d_amd import _lib
    lib = _lib.init(dev_index)
    K, W = args.steps, args.warmup
    m = measure(lib, _lib, args.workload, K, W, args.batch, args.condition, args.solver_fp64, rank, world, dist,
                dev_index, fast_inputs=False, a_smooth=args.a_smooth, lanes=args.lanes)

    if rank == 0:
        elapsed = m["elapsed"]
        achieved = m["roofline"]["achieved"]
        # what a plain y += x stream reaches on this device right now (1 GiB arrays, after the timed
        # steps): the practical ceiling behind the nominal 8 TB/s
        stream = C.c_double(0.0)
        _lib.check(lib.fr3d_stream_probe(1 << 28, 20, C.byref(stream)))
        rstream = C.c_double(0.0)  # read-only stream (the sweep's real traffic is 83 % reads)
        _lib.check(lib.fr3d_read_probe(1 << 26, 20, C.byref(rstream)))
        m["roofline"].update(stream_measured=round(stream.value, 1), read_stream_measured=round(rstream.value, 1),
                             frac_of_stream_measured=round(achieved / stream.value, 4) if stream.value > 0 else None)
        out = {
            "metric": "volumes/sec (3D flow solve + warp)",
            "value": (K * world) / elapsed,
            "unit": "volumes/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if m["mode"] == 0 else "f64",  # update arithmetic; storage format in config.solver
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {m['desc']}; iterations=100, update_lag=5, eta=0.8, "
                                   f"alpha=0.25, a_data=0.45, a_smooth={args.a_smooth:g}; lexicographic-exact SOR",
                       "solver": SOLVER_NAMES[m["mode"]] + (" (library's automatic choice)" if args.solver_fp64 < 0 else ""),
                       "parity_mean_epe_vs_cpu_path": m["parity_mean_epe_vs_cpu"], "parity_source": m["parity_source"],
                       "per_rank_volumes_per_sec": m["per_rank_volumes_per_sec"], "broadcast": m["broadcast"],
                       "volumes_per_gpu_per_step": 1, "lockstep_batch": m["batch_vols"],
                       "engine_lanes": f"{m['lanes']} (fr3d_set_lanes; the lock-step batch is split between the lanes)",
                       "untimed_conditioning_s": args.condition if W > 0 else 0.0,
                       "sharding": f"volume-per-GPU x{world}",
                       "world_size": world, "dist_backend": backend if world > 1 else "none (single process)",
                       "collectives": "one broadcast of the fixed reference (+ barriers and one all_gather_object of the per-rank "
                                      "step times, timing only)" if world > 1 else "none",
                       "device": lib.fr3d_device_info().decode()},
            "roofline": m["roofline"],
            "one_lane": m["one_lane"],
            "kernel_ms_per_step": m["kernel_ms_per_step"],
            "roofline_stages": m["roofline_stages"],
        }
        if world == 1 and not args.no_extras and args.workload == "cfg2":
            # (1) the host-array entry (PCIe both ways) -- reported beside `value`, never as `value`
            out["host_path"] = host_path("cfg2", 8, None if args.solver_fp64 < 0 else args.solver_fp64)
            # (1b) the whole drop-in driver, and one get_displacement call (the reference's API unit)
            out["pipeline"] = pipeline_leg(lib, _lib)
            out["single_pair"] = {"cfg2": single_pair_leg(lib, _lib, "cfg2", args.solver_fp64)}
            # (2) the headline workload with fp32 solver storage (the mode SURVEY 8d's 76 B / update figure is defined on;
            # measured parity 8.6e-5 at 256^3: inside the bound, by a margin too thin for a default) and the 512^3
            # configuration the roofline target is stated on, in three storage modes: the library's choice (packed
            # 42-bit storage), fp32 storage (parity 1.5e-4, above the 1e-4 bound) and fp64 storage (reference-grade).
            # The workspace of the previous leg is released first.
            for key, wl, md, cond, nst in (("cfg2_fp32_storage", "cfg2", 1, 10.0, 8), ("cfg3", "cfg3", -1, 8.0, 4),
                                           ("cfg3_fp32_storage", "cfg3", 1, 5.0, 4), ("cfg3_fp64_storage", "cfg3", 2, 5.0, 4),
                                           ("cfg5", "cfg5", -1, 0.0, 2)):
                _lib.shutdown()
                lib = _lib.init(dev_index)
                if key == "cfg5":  # the 512^3 reference volume is still cached here
                    out["single_pair"]["cfg3"] = single_pair_leg(lib, _lib, "cfg3", args.solver_fp64)
                    _lib.shutdown()
                    lib = _lib.init(dev_index)
                # cfg5 on the O(N) stand-in texture per channel (its parity record is on make_pair's inputs: stated)
                c3 = measure(lib, _lib, wl, nst, 1, 0, cond, md, 0, 1, None, dev_index, fast_inputs=wl == "cfg5", lanes=args.lanes)
                out[key] = {"workload": f"{wl}: {c3['desc']}; same solver parameters", "value": nst / c3["elapsed"],
                            "unit": "volumes/sec", "steps": nst, "warmup": 1, "ms_per_step": 1e3 * c3["elapsed"] / nst,
                            "lockstep_batch": c3["batch_vols"], "untimed_conditioning_s": cond,
                            "dtype": "f32" if c3["mode"] == 0 else "f64",
                            "solver": SOLVER_NAMES[c3["mode"]] + (" (library's automatic choice)" if md < 0 else " (forced)"),
                            "parity_mean_epe_vs_cpu_path": c3["parity_mean_epe_vs_cpu"], "parity_source": c3["parity_source"],
                            "roofline": c3["roofline"], "one_lane": c3["one_lane"],
                            "kernel_ms_per_step": c3["kernel_ms_per_step"], "roofline_stages": c3["roofline_stages"]}
            # (3) the psi_smooth solver path (a_smooth != 1; get_displacement's own default is 0.5, no BASELINE
            # configuration uses it): cfg2 geometry, same parameters otherwise
            _lib.shutdown()
            lib = _lib.init(dev_index)
            sm = measure(lib, _lib, "cfg2", 8, 1, 0, 0.0, args.solver_fp64, 0, 1, None, dev_index, fast_inputs=True,
                         a_smooth=0.5, lanes=args.lanes)
            out["a_smooth_0.5"] = {"workload": "cfg2 geometry with a_smooth=0.5 (psi_smooth re-evaluated every iteration)",
                                   "value": 8 / sm["elapsed"], "unit": "volumes/sec", "steps": 8, "warmup": 1,
                                   "ms_per_step": 1e3 * sm["elapsed"] / 8, "lockstep_batch": sm["batch_vols"],
                                   "solver": SOLVER_NAMES[sm["mode"]], "parity_mean_epe_vs_cpu_path": parity_record("cfg2_asmooth05", sm["mode"], False)[0],
                                   "parity_source": parity_record("cfg2_asmooth05", sm["mode"], False)[1],
                                   "roofline": sm["roofline"], "one_lane": sm["one_lane"],
                                   "kernel_ms_per_step": sm["kernel_ms_per_step"]}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
