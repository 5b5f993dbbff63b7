#!/usr/bin/env python3
"""One run of the product's loop (solver_loop_sweep) on rect M x M from -l1:M with the reference's messages and a time
stamp per line; the solver's own progress lines go to stderr (keeps a long last bound from looking hung).
usage: gpu_ladder_one.py M LIMIT"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from timberborn_support_solver_amd import PLATFORMS_DEFAULT, Encoding, Mi355Sat, PlatformLimits, WorldGrid, solver_loop_sweep
m, limit = int(sys.argv[1]), float(sys.argv[2])
g = WorldGrid.rect(m, m)
e = Encoding.encode(PLATFORMS_DEFAULT, g)
t0 = time.perf_counter()
hist = solver_loop_sweep(g, e, PlatformLimits({(1, 1): m}), out=lambda l: print(f"[{time.perf_counter()-t0:7.1f}s] {l}", flush=True),
                         time_limit=limit, make_solver=lambda: Mi355Sat(verbose=1))
print("history", [(h["k"], h["result"].name, h["count"], round(h["seconds"], 1)) for h in hist], f"total {time.perf_counter()-t0:.1f}s", flush=True)
st = hist[-1].get("stats") or {}
print({k: st.get(k) for k in ("conflicts", "propagations", "shared_exported", "shared_imported", "kernel_seconds", "workers")}, flush=True)
