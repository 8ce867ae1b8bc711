#!/usr/bin/env python3
"""per-level comparison of the verification mode and the ppow oracle through their debug dumps"""
import os, sys, tempfile, glob, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
d = tempfile.mkdtemp(prefix="vdump_")
os.environ["FR3D_ORACLE_DUMP"] = d
os.environ["FR3D_VERIFY_DUMP"] = d
os.environ["FR3D_LIB"] = os.path.join(ROOT, "flowreg3d_amd", "lib", "libflowreg3d_hip_exp.so")
import flowreg3d_amd as fr
from flowreg3d_amd.synthetic import make_pair
from oracle import oracle
oracle.build(); oracle.use_build("ppow")
shape = tuple(int(v) for v in sys.argv[1].split(","))
its, levels = int(sys.argv[2]), int(sys.argv[3])
fixed, moving, _ = make_pair(shape, seed=7, cheap=True)
kw = dict(alpha=(0.25,) * 3, update_lag=5, iterations=its, min_level=0, levels=levels, eta=0.8, a_smooth=1.0, a_data=0.45)
want = oracle.get_displacement(fixed, moving, **kw)
got = fr.get_displacement_verify(fixed, moving, **kw)
print("final differ", int((got != want).sum()))
sizes, _ = fr.pyramid_schedule(*shape, 0.8, levels, 0)
sizes = list(sizes)  # coarse -> fine
for li, (lz, ly, lx) in enumerate(sizes):
    L = len(sizes) - 1 - li
    nl = lz * ly * lx
    inner = (slice(1, -1),) * 3
    def o(tag, ch, shp): return np.fromfile(f"{d}/o_L{L}_{tag}{ch}.bin", np.float64).reshape(shp)
    wo = o("warped", 0, (lz, ly, lx)).astype(np.float32)
    wg = np.fromfile(f"{d}/g_L{L}_warpedf0.bin", np.float32).reshape(lz, ly, lx)
    uo = o("uinit", 0, (lz + 2, ly + 2, lx + 2))[inner].astype(np.float32)
    ug = np.fromfile(f"{d}/g_L{L}_uinitf0.bin", np.float32).reshape(3, lz, ly, lx)[0]
    ro = o("res", 0, (lz + 2, ly + 2, lx + 2, 3))[inner]
    rg = np.moveaxis(np.fromfile(f"{d}/g_L{L}_res0.bin", np.float64).reshape(3, lz, ly, lx), 0, -1)
    line = [f"level {L} {lz}x{ly}x{lx}: uinit differ {int((uo != ug).sum())}, warped differ {int((wo != wg).sum())}, increments differ {int((ro != rg).sum())}"]
    for c in range(3):
        fo = o("u", c, (lz + 2, ly + 2, lx + 2))[inner]
        fg = np.fromfile(f"{d}/g_L{L}_u{c}.bin", np.float64).reshape(lz, ly, lx)
        line.append(f"u{c} differ {int((fo != fg).sum())}")
    print(", ".join(line), flush=True)
# the warp difference at the finest level
lz, ly, lx = sizes[-1]
wo = np.fromfile(f"{d}/o_L0_warped0.bin", np.float64).reshape(lz, ly, lx)
wg = np.fromfile(f"{d}/g_L0_warpedf0.bin", np.float32).reshape(lz, ly, lx)
uo = np.fromfile(f"{d}/o_L0_uinit0.bin", np.float64).reshape(lz + 2, ly + 2, lx + 2)
ug = np.fromfile(f"{d}/g_L0_uinitf0.bin", np.float32).reshape(3, lz, ly, lx)
for idx in np.argwhere(wo.astype(np.float32) != wg)[:4]:
    z, y, x = idx
    print("voxel", (z, y, x), "oracle", repr(float(wo[z, y, x])), "gpu", repr(float(wg[z, y, x])), "flow (u,v,w)", [repr(float(ug[c, z, y, x])) for c in range(3)],
          "fixed there", repr(float(fixed[z, y, x])), "moving there", repr(float(moving[z, y, x])))
    u_, v_, w_ = (np.float32(ug[c, z, y, x]) for c in range(3))
    print("   sample position (x+u, y+v, z+w) as float32:", repr(float(np.float32(np.float64(x) + np.float64(u_)))), repr(float(np.float32(np.float64(y) + np.float64(v_)))), repr(float(np.float32(np.float64(z) + np.float64(w_)))), "bounds", lx - 1, ly - 1, lz - 1)
