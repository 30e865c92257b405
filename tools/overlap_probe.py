#!/usr/bin/env python3
"""Does the chip hide one MSM's bucket grouping (memory-bound, ~10 ms) under another MSM's accumulation (VALU-bound, ~94 ms)?

Two independent contexts (own streams, own work buffers) over the same 2^npow bases run the same MSM
  serial      one after the other on one host thread                        -> 2 T
  concurrent  from two host threads at once (ctypes releases the GIL)       -> 2 T - (what the hardware overlapped)
If the concurrent pair is not faster than the serial pair, a software pipeline that groups batch b + 1 (or the upper windows)
while batch b accumulates has nothing to gain either.  usage: tools/overlap_probe.py [npow] [reps]
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import entries_amd as ea

npow = int(sys.argv[1]) if len(sys.argv) > 1 else 26
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << npow
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1)).to(dev)
bases = tile.repeat(n >> 15, 1).contiguous()
sc = [bench.uniform_scalars(n, bench.R377_TOP, dev, 7 + i) for i in range(2)]
ctxs = []
for i in range(2):
    c = ea.MultiScalarMultContext("bls12_377_g1")
    c.set_bases(bases)
    ctxs.append(c)
del bases
ref = [ctxs[i].run(sc[i])[0] for i in range(2)]
torch.cuda.synchronize()
# run_device works on the stream the scalars are current on: one stream per context (the second at high priority, i.e. on
# its own hardware queue), or both MSMs would simply queue behind each other
streams = [torch.cuda.Stream(), torch.cuda.Stream(priority=-1)]


def run_on(i):
    with torch.cuda.stream(streams[i]):
        return ctxs[i].run(sc[i])[0]


def serial():
    return [run_on(i) for i in range(2)]


def concurrent():
    out = [None, None]

    def work(i):
        out[i] = run_on(i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out


for name, fn in (("serial", serial), ("concurrent", concurrent), ("serial", serial), ("concurrent", concurrent)):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    dt = (time.perf_counter() - t0) / reps * 1e3
    tm = ctxs[0].last_timings()
    print("%-10s 2 x 2^%d: %.2f ms per pair of MSMs  same=%s  (ctx0 stages: digits %.2f sort %.2f accumulate %.2f)" %
          (name, npow, dt, r == ref, tm["digits"], tm["sort"], tm["accumulate"]), flush=True)
