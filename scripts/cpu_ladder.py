#!/usr/bin/env python3
"""CPU restatement's sequential decreasing-k loop (oracle/cdcl.c, 1 core): per-rung verdict, conflicts, seconds.
usage: cpu_ladder.py SIZE [K0] [1x1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as ora
from timberborn_support_solver_amd import PLATFORMS_DEFAULT, Encoding, PlatformLayout, PlatformLimits, WorldGrid
m = int(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else m
defs = [(1, 1)] if len(sys.argv) > 3 else PLATFORMS_DEFAULT
g = WorldGrid.rect(m, m)
e = Encoding.encode(defs, g)
t0 = time.perf_counter()
while True:
    ck = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    o = ora.OracleSolver(); o.add_cnf(ck.lits, ck.offsets)
    t = time.perf_counter(); r = o.solve(); dt = time.perf_counter() - t
    st = o.stats()
    cnt = PlatformLayout.from_assignment(o.model(ck.n_vars)[:e.n_vars], e).platform_count() if r == 10 else None
    print(f"rect {m} k={k}: {'SAT' if r == 10 else 'UNSAT'} count={cnt} conflicts={st['conflicts']} props={st['propagations']} {dt:.2f}s (total {time.perf_counter()-t0:.2f}s)", flush=True)
    if r != 10 or cnt == 0: break
    k = cnt - 1
