#!/usr/bin/env python3
"""Full-size parity fixtures for BASELINE configs 2, 3 and 5: run the CPU oracle (oracle/, the
pinned restatement of the reference's get_displacement) ONCE, in the build container, on the
deterministic synthetic inputs of each configuration and commit a strided / cropped SAMPLE of its
flow field plus a checksum of the inputs under tests/golden/.  tests/test_gpu_fullsize_parity.py
regenerates the same inputs on the GPU box, verifies the checksum, runs the HIP path in its
default (benched) solver mode at the full 100 iterations and compares on the sample.

  python tools/gen_fullsize_golden.py cfg2|cfg2_asmooth05|cfg3|cfg5 [--out tests/golden]

Cost on one core of the build container: cfg2 (256^3) ~4 min / 3 GB, cfg5 (256x512x512, C=2)
~20 min / 25 GB, cfg3 (512^3) ~30 min / 22 GB.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from flowreg3d_amd.synthetic import epe, fullsize_case  # noqa: E402


def input_digest(fixed, moving):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(fixed).tobytes())
    h.update(np.ascontiguousarray(moving).tobytes())
    return h.hexdigest()


def sample(flow, stride, block):
    """strided lattice + one central block of a (Z,Y,X,3) flow."""
    Z, Y, X, _ = flow.shape
    lat = np.ascontiguousarray(flow[::stride, ::stride, ::stride])
    z0, y0, x0 = (Z - block) // 2, (Y - block) // 2, (X - block) // 2
    blk = np.ascontiguousarray(flow[z0:z0 + block, y0:y0 + block, x0:x0 + block])
    return lat, blk, (z0, y0, x0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case", choices=("cfg2", "cfg2_asmooth05", "cfg3", "cfg5", "cfg2_recipe", "cfg3_recipe", "cfg2_recipe_s135", "cfg5_levels8",
                                     "thr_160x176x176", "thr_200"))
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--stride", type=int, default=8)
    ap.add_argument("--block", type=int, default=32)
    ap.add_argument("--ppow", action="store_true",
                    help="run the `ppow` build of the oracle (portable pow in the psi nonlinearities) and keep the lattice in "
                         "float64: the sample the engine's verification mode is compared with bit for bit -> fullsize_<case>_ppow.npz")
    args = ap.parse_args()
    from oracle import oracle
    oracle.build()
    if args.ppow:
        oracle.use_build("ppow")

    fixed, moving, gt, kw = fullsize_case(args.case, warp=oracle.imregister_wrapper)
    digest = input_digest(fixed, moving)
    print(f"{args.case}: inputs {fixed.shape} sha256 {digest[:16]}..., running the oracle", flush=True)
    t0 = time.time()
    flow = oracle.get_displacement(fixed, moving, **kw)
    dt = time.time() - t0
    lat, blk, org = sample(flow, args.stride, args.block)
    crop = 8
    meta = {"case": args.case, "shape_zyx": list(fixed.shape[:3]), "channels": 1 if fixed.ndim == 3 else fixed.shape[3],
            "params": {k: (np.asarray(v).tolist() if hasattr(v, "__len__") else v) for k, v in kw.items()},
            "inputs_sha256": digest, "fixed_sha256": hashlib.sha256(np.ascontiguousarray(fixed).tobytes()).hexdigest(),
            "stride": args.stride, "block": args.block, "block_origin_zyx": list(org),
            "oracle_seconds_1core": dt,
            "epe_oracle_vs_gt_mean_interior8": epe(flow, gt, crop)[0],
            "epe_oracle_vs_gt_max_interior8": epe(flow, gt, crop)[1],
            "flow_mean": [float(x) for x in flow.mean(axis=(0, 1, 2))],
            "oracle_build": "ppow (portable pow in psi)" if args.ppow else "default (C library pow)",
            "generator": "tools/gen_fullsize_golden.py; oracle = oracle/fr3d_oracle.c (pinned by tests/golden/*.npz)"}
    np.savez_compressed(os.path.join(args.out, f"fullsize_{args.case}{'_ppow' if args.ppow else ''}.npz"),
                        lattice=lat.astype(np.float64 if args.ppow else np.float32), block=blk.astype(np.float64),
                        gt_lattice=np.ascontiguousarray(gt[::args.stride, ::args.stride, ::args.stride]).astype(np.float32),
                        moving_lattice=np.ascontiguousarray(moving[::args.stride, ::args.stride, ::args.stride]),
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(json.dumps(meta), flush=True)


if __name__ == "__main__":
    main()
