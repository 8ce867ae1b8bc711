#!/usr/bin/env python3
"""Longer runs of the randomised GPU-vs-oracle generator than the test suite affords (tools/fuzz_vs_oracle.py):
fp64 storage, packed storage, both through the window sweep, the default mode and the verification mode with a_smooth drawn too.
   tools/experiments/fuzz_long.py [cases per configuration]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fuzz_vs_oracle as fz
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
total_bad = 0
for name, kw in (("fp64 storage", dict(mode=2)), ("packed storage", dict(mode=3)), ("default mode", dict(mode=None)),
                 ("fp64 storage, window sweep", dict(mode=2, sweep=2)), ("packed storage, window sweep", dict(mode=3, sweep=2)),
                 ("verification mode incl. a_smooth != 1", dict(mode="verify", verify_smooth=True))):
    bad, worst = fz.run(n_cases=n, seed=1000 + len(name), verbose=False, **kw)
    print(f"{name}: {n} cases, {bad} bad, worst scaled mean EPE {worst:.2e}", flush=True)
    total_bad += bad
sys.exit(1 if total_bad else 0)
