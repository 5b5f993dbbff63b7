#!/usr/bin/env python3
"""Throughput of one search slice vs worker count / LDS-assignment variant (rect 64x64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from timberborn_support_solver_amd import *
import ctypes
size, k, slc = 64, 46, 50
grid = WorldGrid.rect(size, size)
enc = Encoding.encode(PLATFORMS_DEFAULT, grid)
cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
cases = []
for occ in (2, 3, 4):
    for W in (1280, occ * 1024, occ * 1536):
        cases.append((occ, -1, W))
cases += [(2, 1, 1536), (4, 1, 1536)]
libs = {}
for occ, lds, W in cases:
    if occ not in libs:
        libs[occ] = ctypes.CDLL(os.path.join(ROOT, "timberborn_support_solver_amd", f"libmi355sat_occ{occ}.so"))
    print(f"occ{occ}", end=" ")
    s = Mi355Sat(workers=W, slice_ms=400, conflict_budget=1, lds_val=lds, _lib_override=libs[occ])
    s.add_cnf(cnf.lits, cnf.offsets)
    r = s.solve()
    st = s.stats()
    print(f"lds_val={lds} W={W}: kernel={st['kernel_seconds']:.3f}s props={st['propagations']} props/s={st['propagations']/st['kernel_seconds']:.3e} "
          f"confl/s={st['conflicts']/st['kernel_seconds']:.3e} alg GB/s={algorithmic_bytes(st)/st['kernel_seconds']/1e9:.1f}", flush=True)
    s.close()
