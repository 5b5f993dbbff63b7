import sys, time
sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, WorldGrid
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT
for size, k in [(32, 120), (64, 200)]:
    g = WorldGrid.rect(size, size); e = Encoding.encode(PLATFORMS_DEFAULT, g)
    c = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    for rep in range(2):
        t0 = time.perf_counter()
        s = Mi355Sat(verbose=1, conflict_budget=1)
        s.add_cnf(c.lits, c.offsets)
        r = s.solve(); t2 = time.perf_counter()
        s.close(); t3 = time.perf_counter()
        print(f"rect {size}: solve {t2-t0:.3f}s free {t3-t2:.3f}s", flush=True)
