"""Fixed cost of one solve() (new handle, add_cnf, solve, model, free) on easy SAT instances. GPU box only."""
import sys
import time

sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, WorldGrid  # noqa: E402
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402

for size, k in [(8, 20), (16, 40), (24, 24), (32, 120), (64, 200)]:
    g = WorldGrid.rect(size, size)
    e = Encoding.encode(PLATFORMS_DEFAULT, g)
    c = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    for rep in range(3):
        t0 = time.perf_counter()
        s = Mi355Sat()
        s.add_cnf(c.lits, c.offsets)
        t1 = time.perf_counter()
        r = s.solve()
        t2 = time.perf_counter()
        m = s.full_solution(c.n_vars)
        st = s.stats()
        s.close()
        t3 = time.perf_counter()
        print(f"rect {size} k={k} clauses={c.n_clauses}: new+add {t1-t0:.3f}s solve {t2-t1:.3f}s (kernel {st['kernel_seconds']:.3f}s, "
              f"{st['kernel_launches']} launches, {st['conflicts']} conflicts) model+free {t3-t2:.3f}s -> {r.name}", flush=True)
