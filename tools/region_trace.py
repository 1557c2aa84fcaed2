#!/usr/bin/env python3
"""Where do the 20 timed steps of `bench.py --steps 20 --warmup 5` spend more than the steady-state step?  Per-step GPU events + host timestamps
around run.step() for the timed region (same Runner as bench.py), printed per step.  usage: python tools/region_trace.py [--steps 20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
import bench, torch
a = bench.parse()
torch.cuda.set_device(0)
run = bench.Runner(a, 0, 1)
if a.prewarm > 0:
    bench._prewarm_clocks(torch, run, a.prewarm)
for rep in range(3):
    for _ in range(a.warmup):
        run.step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for s in range(a.steps):
        h0 = time.perf_counter()
        run.step()
        host.append((time.perf_counter() - h0) * 1e6)
        ev[s + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    gpu = [1e3 * ev[s].elapsed_time(ev[s + 1]) for s in range(a.steps)]
    print(f"rep {rep}: wall {wall / a.steps:.1f} us/step; GPU event span {sum(gpu) / a.steps:.1f} us/step")
    print("   gpu us per step :", " ".join(f"{g:5.0f}" for g in gpu))
    print("   host us per call:", " ".join(f"{h:5.0f}" for h in host))
run.close()
