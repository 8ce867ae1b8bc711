"""Deterministic synthetic volumes and motion for tests and bench.py.

Pure NumPy/SciPy, no reference import: smooth dense texture (Gaussian-blurred PCG64 noise plus a
few Gaussian blobs), ground-truth flow = translation + small rotation about z through the centre
(closed form of the reference's Translational/Rotational3DFlowAugmentor,
motion_generation/motion_generators.py:69-180), moving = backward cubic warp of fixed.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def texture(shape, seed=1234, sigma=2.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    vol = ndimage.gaussian_filter(rng.random(shape, dtype=np.float32), sigma, mode="reflect")
    Z, Y, X = shape
    zz, yy, xx = np.meshgrid(np.arange(Z, dtype=np.float32), np.arange(Y, dtype=np.float32),
                             np.arange(X, dtype=np.float32), indexing="ij", sparse=True)
    vol = (vol - vol.min()) / (vol.max() - vol.min())
    s = max(Z, 4) / 10.0
    for _ in range(8):
        cz, cy, cx = rng.random(3) * np.array([Z, Y, X])
        vol = vol + 0.5 * np.exp(-((zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)).astype(np.float32)
    vol = (vol - vol.min()) / (vol.max() - vol.min())
    return vol.astype(np.float32)


def flow_gt(shape, translation=(1.7, -1.1, 0.6), rot_deg=1.5, scale=1.0):
    """(Z,Y,X,3) float32 displacement [dx,dy,dz]: translation + rotation about the z axis."""
    Z, Y, X = shape
    yy, xx = np.meshgrid(np.arange(Y, dtype=np.float64), np.arange(X, dtype=np.float64), indexing="ij")
    th = np.deg2rad(rot_deg * scale)
    cy, cx = (Y - 1) / 2.0, (X - 1) / 2.0
    rx = (np.cos(th) - 1.0) * (xx - cx) - np.sin(th) * (yy - cy)
    ry = np.sin(th) * (xx - cx) + (np.cos(th) - 1.0) * (yy - cy)
    f = np.empty((Z, Y, X, 3), np.float32)
    f[..., 0] = (rx + translation[0] * scale)[None]
    f[..., 1] = (ry + translation[1] * scale)[None]
    f[..., 2] = translation[2] * scale
    return f


def flow_expansion_rotation(shape, expansion=(0.010, 0.008, 0.006), rot_deg=(2.0, 2.0, 2.0), scale=1.0):
    """(Z,Y,X,3) float32 [dx,dy,dz]: anisotropic expansion about the centre ("injection/recoil"
    stand-in, cf. Expansion3DFlowAugmentor, motion_generation/motion_generators.py:236-301) plus small
    rotations about all three axes (Rotational3DFlowAugmentor :69-152), closed form."""
    Z, Y, X = shape
    zz, yy, xx = np.meshgrid(np.arange(Z, dtype=np.float64), np.arange(Y, dtype=np.float64),
                             np.arange(X, dtype=np.float64), indexing="ij", sparse=True)
    cz, cy, cx = (Z - 1) / 2.0, (Y - 1) / 2.0, (X - 1) / 2.0
    px, py, pz = xx - cx, yy - cy, zz - cz
    ax, ay, az = (np.deg2rad(a * scale) for a in rot_deg)  # about x, y, z
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx - np.eye(3)
    f = np.empty((Z, Y, X, 3), np.float32)
    for d in range(3):
        f[..., d] = (R[d, 0] * px + R[d, 1] * py + R[d, 2] * pz
                     + scale * expansion[d] * (px, py, pz)[d]).astype(np.float32)
    return f


def backward_warp(vol, flow, order=3):
    """vol sampled at x + flow (same convention as imregister_wrapper)."""
    Z, Y, X = vol.shape
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    coords = [zz + flow[..., 2], yy + flow[..., 1], xx + flow[..., 0]]
    return ndimage.map_coordinates(vol.astype(np.float64), coords, order=order, mode="nearest").astype(np.float32)


def make_pair(shape, seed=1234, channels=1, scale=1.0, cheap=False, motion="rigid"):
    """-> fixed, moving (Z,Y,X[,C]) float32 in [0,1] and the ground-truth flow (Z,Y,X,3).

    moving(x) = fixed(x - flow), so get_displacement(fixed, moving) ~ flow.  ``cheap`` uses
    linear interpolation (fast for 256^3/512^3 bench volumes)."""
    gt = flow_gt(shape, scale=scale) if motion == "rigid" else flow_expansion_rotation(shape, scale=scale)
    fs, ms = [], []
    for c in range(channels):
        f = texture(shape, seed + c)
        fs.append(f)
        ms.append(backward_warp(f, -gt, order=1 if cheap else 3))
    if channels == 1:
        return fs[0], ms[0], gt
    return np.stack(fs, -1), np.stack(ms, -1), gt


def _lerp_shift(vol, shift, axis):
    """vol sampled at index + shift along one axis, linear interpolation, edge-clamped (O(N))."""
    n = vol.shape[axis]
    pos = np.clip(np.arange(n, dtype=np.float64) + shift, 0, n - 1)
    i0 = np.floor(pos).astype(np.int64)
    i1 = np.minimum(i0 + 1, n - 1)
    w = (pos - i0).astype(np.float32)
    shp = [1, 1, 1]
    shp[axis] = n
    w = w.reshape(shp)
    return np.take(vol, i0, axis=axis) * (1.0 - w) + np.take(vol, i1, axis=axis) * w


def fast_pair(shape, shift=(1.7, -1.1, 0.6), seed=1234, block=128):
    """O(N) stand-in for make_pair at 256^3 / 512^3 test sizes: a periodic blurred-noise block tiled
    over the volume plus one broad bump (breaks the periodicity), and a pure translation
    moving(x) = fixed(x - shift) by separable linear interpolation.  -> fixed, moving, flow"""
    Z, Y, X = shape
    rng = np.random.Generator(np.random.PCG64(seed))
    b = min(block, Z, Y, X)
    base = ndimage.gaussian_filter(rng.random((b, b, b), dtype=np.float32), 2.0, mode="wrap")
    base = (base - base.min()) / (base.max() - base.min())
    reps = [-(-n // b) for n in shape]
    vol = np.tile(base, reps)[:Z, :Y, :X]
    g = [np.exp(-0.5 * ((np.arange(n, dtype=np.float32) - 0.45 * n) / (0.3 * n)) ** 2) for n in shape]
    fixed = (0.7 * vol + 0.3 * g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]).astype(np.float32)
    moving = fixed
    for axis, d in zip((2, 1, 0), shift):  # shift = (dx, dy, dz)
        moving = _lerp_shift(moving, -d, axis)
    flow = np.empty(shape + (3,), np.float32)
    flow[...] = np.asarray(shift, np.float32)
    return fixed, moving.astype(np.float32), flow


def epe(a, b, crop=0):
    """mean / max end-point error between two (Z,Y,X,3) flows, optionally cropped per side."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if crop:
        s = (slice(crop, -crop),) * 3
        a, b = a[s], b[s]
    d = np.linalg.norm(a - b, axis=-1)
    return float(d.mean()), float(d.max())


# OFOptions solver defaults (motion_correction/OF_options_3D.py:155-174), the parameters every BASELINE
# configuration is quoted with
SOLVER_DEFAULTS = dict(alpha=(0.25, 0.25, 0.25), update_lag=5, iterations=100, min_level=0, eta=0.8,
                       a_smooth=1.0, a_data=0.45)


def fullsize_case(name, warp=None):
    """Deterministic full-size inputs of BASELINE configs 2, 3 and 5 (the cases whose CPU-path flows
    are sampled in tests/golden/fullsize_*.npz): -> fixed, moving, flow_gt, get_displacement kwargs.
    `warp`: imregister_wrapper of the side that generates the inputs (recipe cases only)."""
    if name == "cfg2":
        fixed, moving, gt = fast_pair((256, 256, 256))
        return fixed, moving, gt, dict(SOLVER_DEFAULTS, levels=4)
    if name == "cfg2_asmooth05":
        # config 2's volume and pyramid with get_displacement's own default a_smooth = 0.5 (the psi_smooth
        # solver path; no BASELINE configuration uses it)
        fixed, moving, gt = fast_pair((256, 256, 256))
        return fixed, moving, gt, dict(SOLVER_DEFAULTS, levels=4, a_smooth=0.5)
    if name == "cfg3":
        fixed, moving, gt = fast_pair((512, 512, 512))
        return fixed, moving, gt, dict(SOLVER_DEFAULTS, levels=5)
    if name in ("cfg2_recipe", "cfg3_recipe", "cfg2_recipe_s135"):
        # SURVEY section 8d's input recipe exactly as bench.py generates it: texture() (blurred PCG64 noise + 8 blobs),
        # translation (1.7,-1.1,0.6) + 1.5 degree rotation about z, moving = the path's own cubic compensation warp of
        # fixed by -flow with fixed as the out-of-bounds fill (imregister_wrapper, core/optical_flow_3d.py:22-74).
        # `warp(f2, u, v, w, f1)` is that function: the CPU restatement's in the build container, the engine's on the GPU box
        # (they agree to the last float32 bit on all but ~1 % of the voxels, which is why the fixture checks the moving
        # volume on a sample instead of by checksum).
        if warp is None:
            raise ValueError("recipe cases need the imregister_wrapper to generate the moving volume with")
        # "_s135": the largest motion of bench.py's synthetic series (time points scale the field by sin(2 pi t/64) + 0.35)
        n = 512 if name == "cfg3_recipe" else 256
        fixed = texture((n, n, n), seed=1234)
        gt = flow_gt((n, n, n), scale=1.35 if name.endswith("_s135") else 1.0)
        moving = np.asarray(warp(fixed, -gt[..., 0], -gt[..., 1], -gt[..., 2], fixed), dtype=np.float32).reshape(fixed.shape)
        return fixed, moving, gt, dict(SOLVER_DEFAULTS, levels=4 if n == 256 else 5)
    if name in ("thr_160x176x176", "thr_200"):
        # single-channel volumes just above FR3D_SOLVER_AUTO's switch from fp32 to packed solver storage (2^22 voxels):
        # 4.96 M and 8 M voxels, a non-cubic and a cubic one (the switch point was pinned at 128^3 and 256^3 only)
        shape = (160, 176, 176) if name == "thr_160x176x176" else (200, 200, 200)
        fixed, moving, gt = fast_pair(shape)
        return fixed, moving, gt, dict(SOLVER_DEFAULTS, levels=4)
    if name == "cfg5_levels8":
        # the survey's own config-5 schedule (levels=8, min_level=0 -> 9 solves); the pyramid does
        # not capture the 23-voxel corner motion (CPU and GPU both end 3.45 voxels from the ground truth), which is
        # why cfg5 proper uses 13 levels -- kept as a parity case: GPU == CPU whatever the schedule
        fixed, moving, gt = make_pair((256, 512, 512), seed=1234, channels=2, motion="expansion", cheap=True)
        return fixed, moving, gt, dict(SOLVER_DEFAULTS, levels=8, weight=np.array([0.5, 0.5]))
    if name == "cfg5":
        fixed, moving, gt = make_pair((256, 512, 512), seed=1234, channels=2, motion="expansion", cheap=True)
        # levels=12 -> 13 solves down to 18x35x35: the 2-degree rotations move the corners of a 256x512x512
        # volume by up to 23 voxels, which a 9-solve pyramid (coarsest 43x86x86, round 1) does not capture --
        # neither on the CPU nor on the GPU
        return fixed, moving, gt, dict(SOLVER_DEFAULTS, levels=12, weight=np.array([0.5, 0.5]))
    raise ValueError(name)
