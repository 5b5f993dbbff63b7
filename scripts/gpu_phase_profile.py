#!/usr/bin/env python3
"""Per-phase cycle shares of the search kernel (diagnostic build libmi355sat_prof.so)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from timberborn_support_solver_amd import *
prof = ctypes.CDLL(os.path.join(ROOT, "timberborn_support_solver_amd", "libmi355sat_prof.so"))
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 46
W = int(sys.argv[3]) if len(sys.argv) > 3 else 3072
slc = int(sys.argv[4]) if len(sys.argv) > 4 else 500   # slice length in ms
grid = WorldGrid.rect(size, size)
enc = Encoding.encode(PLATFORMS_DEFAULT, grid)
cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
for kw in (dict(),):
    s = Mi355Sat(workers=W, slice_ms=slc, conflict_budget=1, verbose=1, ramp=-1, share=-1, _lib_override=prof, **kw)
    s.add_cnf(cnf.lits, cnf.offsets)
    t = time.time(); r = s.solve(); dt = time.time() - t
    st = s.stats()
    print(kw, r.name, f"kernel={st['kernel_seconds']:.3f}s props={st['propagations']} conflicts={st['conflicts']} "
          f"props/s={st['propagations']/st['kernel_seconds']:.3e} steps={st['bcp_steps']} watch/prop={st['n_watch']/st['propagations']:.2f} "
          f"cl_lit/prop={st['n_cl_lit']/st['propagations']:.2f} move/prop={st['n_move']/st['propagations']:.3f}", flush=True)
    s.close()
