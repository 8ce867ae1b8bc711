#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC summaries of a round.

usage: tools/make_pmc_traffic.py r01       (reads profiles/r01_pmc_{fetch,write}_cfg{2,3}.txt)

HBM bytes of k_sor_step = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, both
in KiB, divided by the voxel updates of the run (sum over pyramid levels of voxels x iterations, one
volume).  The correction is checked in the same pass on k_axpy (reads two 4-B streams, writes one):
its raw FETCH_SIZE equals its WRITE_SIZE, i.e. half of what it reads.  bench.py multiplies
bytes_per_update by the updates per launch to fill roofline.traffic.
"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # WORKLOADS, solver_kwargs
from flowreg3d_amd.core import pyramid_schedule


def counter(path, kernel, name):
    for line in open(path):
        if kernel in line and name in line:
            m = re.search(r"dispatches\s+(\d+)\s+sum\s+([0-9.e+]+)", line)
            return int(m.group(1)), float(m.group(2))
    raise SystemExit(f"{kernel}/{name} not found in {path}")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of "
                    "`bench.py --workload W --steps 1 --warmup 0 --batch 1 --solver-fp64 M --lanes 1 --no-extras`, kernel k_sor_step; "
                    "keys are workload/mode<M> (fr3d_params.solver_fp64: 1 fp32 storage, 2 fp64 storage, 3 packed 42-bit "
                    "storage); FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated in the same pass on k_axpy, which "
                    "reports exactly 1/2 of a known 4-B-per-lane coalesced stream); KiB units; made by "
                    "tools/make_pmc_traffic.py " + tag}
    import subprocess
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = "?"
    out["_taken_at"] = f"round {tag.lstrip('r0') or '?'}, sources as of commit {head or '?'}"
    out["_sweep_source_hash"] = bench.sweep_source_hash()  # bench.py quotes the record only for these kernel sources
    for wl, mode in (("cfg2", 3), ("cfg2", 1), ("cfg3", 3), ("cfg3", 1), ("cfg3", 2)):
        pf = os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_{wl}_m{mode}.txt")
        pw = os.path.join(ROOT, "profiles", f"{tag}_pmc_write_{wl}_m{mode}.txt")
        if not (os.path.exists(pf) and os.path.exists(pw)):
            continue
        Z, Y, X, levels, _ = bench.WORKLOADS[wl]
        kw = bench.solver_kwargs(levels)
        sizes, _ = pyramid_schedule(Z, Y, X, kw["eta"], kw["levels"], kw["min_level"])
        updates = sum(z * y * x for z, y, x in sizes) * kw["iterations"]
        n, fetch = counter(pf, "k_sor_step", "FETCH_SIZE")
        _, write = counter(pw, "k_sor_step", "WRITE_SIZE")
        _, af = counter(pf, "k_axpy", "FETCH_SIZE")
        _, aw = counter(pw, "k_axpy", "WRITE_SIZE")
        total = (2.0 * fetch + write) * 1024.0
        out[f"{wl}/mode{mode}"] = {"launches": n, "voxel_updates": updates, "fetch_size_kib_raw": fetch, "write_size_kib": write,
                                   "hbm_bytes_total": total, "bytes_per_update": total / updates,
                                   "algorithmic_bytes_per_update": 19 * bench.STORAGE_BYTES[mode],
                                   # axpy reads 2 streams and writes 1: corrected fetch / write must be 2
                                   "axpy_calibration_fetch_over_write": af / aw}
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: round(v["bytes_per_update"], 1) for k, v in out.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main()
