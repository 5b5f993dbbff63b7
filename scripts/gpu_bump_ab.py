"""A/B of decision-queue bump policies (builds with -DMS_BUMP_MODE=n): conflicts a small portfolio needs
for the same verdicts, exchange and ramp off.  GPU box only."""
import ctypes
import os
import sys
import time

sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, WorldGrid  # noqa: E402
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402

libs = sys.argv[1:] or [""]
W = 8
for size, k in [(20, 6), (24, 8)]:
    g = WorldGrid.rect(size, size)
    e = Encoding.encode(PLATFORMS_DEFAULT, g)
    c = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    for lib in libs:
        for seed in (1, 2):
            s = Mi355Sat(workers=W, slice_ms=20, share=-1, ramp=-1, conflict_budget=300000 * W, seed=seed,
                         _lib_override=ctypes.CDLL(os.path.abspath(lib)) if lib else None)
            s.add_cnf(c.lits, c.offsets)
            t0 = time.perf_counter()
            r = s.solve()
            dt = time.perf_counter() - t0
            st = s.stats()
            print(f"rect {size} k={k} {os.path.basename(lib) or 'default':24s} seed {seed}: {r.name:11s} {dt:6.1f}s  {st['conflicts']/W:9.0f} conflicts/worker  "
                  f"{st['propagations']/max(1,st['conflicts']):7.0f} props/conflict  learnt len {st['learnt_literals']/max(1,st['learnts']):.1f}", flush=True)
            s.close()
