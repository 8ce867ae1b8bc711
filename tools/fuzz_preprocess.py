"""Randomised preprocessing inputs (dtypes, shapes with and without a time axis, per-channel sigmas,
normalisation modes) -- device normalize + Gaussian filter against the oracle restatement.
usage (GPU box): python tools/fuzz_preprocess.py [n_cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowreg3d_amd import _lib, preprocess as pp
from oracle import oracle


def run(n_cases=40, seed=0, verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    rng = np.random.default_rng(seed)
    _lib.init()
    bad = 0
    for case in range(n_cases):
        zyx = tuple(int(v) for v in rng.choice([1, 2, 3, 5, 8, 13, 21, 40], size=3))
        C = int(rng.choice([1, 2, 3]))
        T = int(rng.choice([1, 3, 5])) if rng.random() < 0.7 else None
        shape = ((T,) if T is not None else ()) + zyx + (C,)
        dt = rng.choice([np.uint8, np.uint16, np.int16, np.float32, np.float64])
        if np.issubdtype(dt, np.integer):
            info = np.iinfo(dt)
            arr = rng.integers(max(info.min, -3000), min(info.max, 3000), size=shape).astype(dt)
        else:
            arr = (rng.standard_normal(shape) * 50 + 10).astype(dt)
        mode = str(rng.choice(["together", "separate"]))
        sig = rng.uniform(0.3, 2.5, size=(C, int(rng.choice([3, 4])))) if rng.random() < 0.6 else rng.uniform(0.3, 2.5, size=3)
        try:
            if arr.size == 0:
                got = pp.apply_gaussian_filter(pp.normalize(arr.astype(np.float64), channel_normalization=mode), sig)
                ok = got.shape == arr.shape
            else:
                want = oracle.apply_gaussian_filter(oracle.normalize(arr.astype(np.float64), channel_normalization=mode), sig)
                got = pp.apply_gaussian_filter(pp.normalize(arr.astype(np.float64), channel_normalization=mode), sig)
                ok = got.shape == want.shape and np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
                if hasattr(pp, "preprocess_frames"):
                    fused = pp.preprocess_frames(arr, sigma=sig, channel_normalization=mode)
                    ok = ok and np.abs(fused - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
        except Exception as e:  # noqa
            ok = False
            say("EXC", repr(e))
        bad += not ok
        say("%s case %2d shape %s %s %s sigma %s" % ("ok " if ok else "BAD", case, shape, np.dtype(dt).name, mode, np.shape(sig)), flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("cases %d bad %d" % (n, run(n, sd)))
