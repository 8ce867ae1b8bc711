#!/usr/bin/env python3
"""Where do the shipped modes' 2.8e-4 on config 5 come from?  On a half-scale config-5 volume (two channels) the
verification mode (= the CPU path, bit for bit) is compared with (a) itself with the fp32 level tail of the shipped modes
(experiment build, FR3D_VERIFY_TAIL32=1), (b) the shipped fp64-storage and packed modes.
usage (GPU box): FR3D_LIB=.../libflowreg3d_hip_exp.so python tools/experiments/cfg5_decompose.py [scale]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import flowreg3d_amd as fr
from flowreg3d_amd.synthetic import make_pair, SOLVER_DEFAULTS
Z, Y, X, levels, which, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
fixed, moving, gt = make_pair((Z, Y, X), seed=1234, channels=2, motion="expansion", cheap=True)
kw = dict(SOLVER_DEFAULTS, levels=levels, weight=np.array([0.5, 0.5]))
if which == "verify":
    f = fr.get_displacement_verify(fixed, moving, **kw)
else:
    f = fr.get_displacement(fixed, moving, solver_fp64=int(which), **kw)
np.save(out, f)
""" % ROOT


def main():
    sc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
    Z, Y, X = int(256 * sc), int(512 * sc), int(512 * sc)
    levels = 12 if sc >= 1 else 10
    runs = [("verify", {}), ("verify_tail32", {"FR3D_VERIFY_TAIL32": "1"}), ("2", {}), ("3", {}), ("1", {})]
    flows = {}
    for name, env in runs:
        out = f"/tmp/cfg5dec_{name}.npy"
        which = "verify" if name.startswith("verify") else name
        r = subprocess.run([sys.executable, "-c", CHILD, str(Z), str(Y), str(X), str(levels), which, out],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
        if r.returncode != 0:
            print(json.dumps({"run": name, "error": r.stderr[-400:]}), flush=True)
            continue
        flows[name] = np.load(out)
    ref = flows["verify"]
    for name, f in flows.items():
        if name == "verify":
            continue
        d = np.linalg.norm(f - ref, axis=-1)
        print(json.dumps({"shape": [Z, Y, X], "levels": levels, "run": name, "mean_epe_vs_verify": float(d.mean()),
                          "max": float(d.max()), "p99": float(np.quantile(d, 0.99))}), flush=True)


if __name__ == "__main__":
    main()
