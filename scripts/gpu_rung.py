#!/usr/bin/env python3
"""One rung (rect SIZE, at-most-K) under several solver configurations: time to verdict, conflicts, per-worker rate.
usage: gpu_rung.py SIZE K TIME_LIMIT "workers=256,share_lbd=4;workers=1024" [1x1]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from timberborn_support_solver_amd import PLATFORMS_DEFAULT, Encoding, Mi355Sat, PlatformLimits, WorldGrid
m, k, limit = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv) for c in sys.argv[4].split(";")]
defs = [(1, 1)] if len(sys.argv) > 5 else PLATFORMS_DEFAULT
lib = None
if os.environ.get("BENCH_LIB"):
    import ctypes
    lib = ctypes.CDLL(os.path.abspath(os.environ["BENCH_LIB"]))
g = WorldGrid.rect(m, m)
e = Encoding.encode(defs, g)
cnf = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
print(f"rect {m} k={k}: vars {cnf.n_vars} clauses {cnf.n_clauses}", flush=True)
for cfg in cfgs:
    s = Mi355Sat(_lib_override=lib, **cfg)
    s.add_cnf(cnf.lits, cnf.offsets)
    it = s.interrupter()
    tm = threading.Timer(limit, it.interrupt); tm.start()
    t = time.perf_counter(); r = s.solve(); dt = time.perf_counter() - t
    tm.cancel()
    st = s.stats()
    W = cfg.get("workers", 0)
    print(f"{cfg}: {r.name} {dt:.2f}s kernel={st['kernel_seconds']:.2f}s conflicts={st['conflicts']:.3e} "
          f"({st['conflicts']/max(st['kernel_seconds'],1e-9):.3e}/s) props={st['propagations']:.3e} learnts={st['learnts']} "
          f"exp={st['shared_exported']} imp={st['shared_imported']} units={st['shared_imported_units']} restarts={st['restarts']} "
          f"simp: units={st['simp_units']} equiv={st['simp_equivalences']} removed={st['simp_clauses_removed']} eliminated={st['simp_eliminated']}", flush=True)
    s.close()
