#!/usr/bin/env python3
"""Host <-> device copy rates on this box (pageable / pinned, one and several copier threads) and the cost of pinning:
what bounds the host-array path of fr3d_process_batch_raw.  usage (GPU box): python tools/experiments/pcie_probe.py"""
import json
import threading
import time

import numpy as np
import torch

N = 512 << 20  # bytes per buffer
dev = torch.device("cuda:0")
d = torch.empty(N, dtype=torch.uint8, device=dev)
d2 = torch.empty(N, dtype=torch.uint8, device=dev)
page = torch.from_numpy(np.ones(N, np.uint8))
page2 = torch.from_numpy(np.ones(N, np.uint8))
pin = torch.empty(N, dtype=torch.uint8).pin_memory()
pin2 = torch.empty(N, dtype=torch.uint8).pin_memory()
pin.fill_(1), pin2.fill_(1)


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def par(fns):
    def run():
        th = [threading.Thread(target=f) for f in fns]
        [x.start() for x in th]
        [x.join() for x in th]
    return run


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def on(stream, fn):
    def f():
        with torch.cuda.stream(stream):
            fn()
        stream.synchronize()
    return f


out = {}
out["h2d_pageable_GBs"] = N / t(lambda: d.copy_(page)) / 1e9
out["h2d_pinned_GBs"] = N / t(lambda: d.copy_(pin, non_blocking=True)) / 1e9
out["d2h_pageable_GBs"] = N / t(lambda: page.copy_(d)) / 1e9
out["d2h_pinned_GBs"] = N / t(lambda: pin.copy_(d, non_blocking=True)) / 1e9
out["h2d_pageable_2threads_GBs"] = 2 * N / t(par([on(s1, lambda: d.copy_(page)), on(s2, lambda: d2.copy_(page2))])) / 1e9
out["h2d_pinned_2streams_GBs"] = 2 * N / t(par([on(s1, lambda: d.copy_(pin, non_blocking=True)), on(s2, lambda: d2.copy_(pin2, non_blocking=True))])) / 1e9
out["bidirectional_pinned_GBs_each"] = N / t(par([on(s1, lambda: d.copy_(pin, non_blocking=True)), on(s2, lambda: pin2.copy_(d2, non_blocking=True))])) / 1e9
fresh = np.ones(N, np.uint8)
rt = torch.cuda.cudart()
t0 = time.perf_counter()
rc = rt.cudaHostRegister(fresh.ctypes.data, N, 0)
out["host_register_512MiB_ms"] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter()
rt.cudaHostUnregister(fresh.ctypes.data)
out["host_unregister_512MiB_ms"] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter()
x = np.empty(N, np.uint8)
x[::4096] = 1
out["fault_in_fresh_512MiB_ms"] = 1e3 * (time.perf_counter() - t0)
print(json.dumps({k: round(v, 2) for k, v in out.items()}))
