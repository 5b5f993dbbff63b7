"""Search quality of ONE device worker against the CPU restatement: conflicts needed for the same verdict
(exchange / ramp / portfolio all off).  GPU box only."""
import sys
import time

sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, WorldGrid  # noqa: E402
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402

for size, k in [(20, 6), (24, 8), (24, 9)]:
    g = WorldGrid.rect(size, size)
    e = Encoding.encode(PLATFORMS_DEFAULT, g)
    c = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    for W in (1, 2, 4):
        s = Mi355Sat(workers=W, slice_ms=50, share=-1, ramp=-1, conflict_budget=400000 * W)
        s.add_cnf(c.lits, c.offsets)
        t0 = time.perf_counter()
        r = s.solve()
        dt = time.perf_counter() - t0
        st = s.stats()
        print(f"rect {size} k={k} workers={W}: {r.name} in {dt:.1f}s, conflicts {st['conflicts']} ({st['conflicts']/W:.0f} per worker), "
              f"props {st['propagations']:.3e}, restarts {st['restarts']}, learnts kept {st['learnts']}", flush=True)
        s.close()
