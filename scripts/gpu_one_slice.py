#!/usr/bin/env python3
"""One search-kernel slice on rect SIZE (for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from timberborn_support_solver_amd import *
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 46
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
slc = int(sys.argv[4]) if len(sys.argv) > 4 else 50
lds = int(sys.argv[5]) if len(sys.argv) > 5 else 0
grid = WorldGrid.rect(size, size)
enc = Encoding.encode(PLATFORMS_DEFAULT, grid)
cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
s = Mi355Sat(workers=W, slice_conflicts=slc, conflict_budget=W * slc, lds_val=lds)
s.add_cnf(cnf.lits, cnf.offsets)
r = s.solve()
st = s.stats()
print(r.name, f"kernel={st['kernel_seconds']:.3f}s props={st['propagations']} props/s={st['propagations']/st['kernel_seconds']:.3e} steps={st['bcp_steps']}")
