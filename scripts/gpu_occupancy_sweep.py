#!/usr/bin/env python3
"""Throughput of one search slice vs worker count / LDS-assignment variant (rect 64x64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from timberborn_support_solver_amd import *
size, k, slc = 64, 46, 50
grid = WorldGrid.rect(size, size)
enc = Encoding.encode(PLATFORMS_DEFAULT, grid)
cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
for lds, W in [(1, 1280), (1, 1536), (-1, 1280), (-1, 2048), (-1, 3072), (-1, 4096), (-1, 6144)]:
    s = Mi355Sat(workers=W, slice_conflicts=slc, conflict_budget=W * slc, lds_val=lds)
    s.add_cnf(cnf.lits, cnf.offsets)
    r = s.solve()
    st = s.stats()
    print(f"lds_val={lds} W={W}: kernel={st['kernel_seconds']:.3f}s props={st['propagations']} props/s={st['propagations']/st['kernel_seconds']:.3e} "
          f"confl/s={st['conflicts']/st['kernel_seconds']:.3e} alg GB/s={algorithmic_bytes(st)/st['kernel_seconds']/1e9:.1f}", flush=True)
    s.close()
