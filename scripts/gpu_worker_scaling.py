"""Per-worker and aggregate rates of ONE search slice vs the number of concurrent workers (exchange and
ramp off): how much of a worker's speed is memory latency it would also see alone.  GPU box only."""
import sys

sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, WorldGrid, algorithmic_bytes  # noqa: E402
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402

for size, k in [(64, 46), (24, 8)]:
    g = WorldGrid.rect(size, size)
    e = Encoding.encode(PLATFORMS_DEFAULT, g)
    c = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    for W in (1, 16, 64, 256, 1024, 2048, 4096):
        s = Mi355Sat(workers=W, slice_ms=400, conflict_budget=1, share=-1, ramp=-1)
        s.add_cnf(c.lits, c.offsets)
        s.solve()
        st = s.stats()
        ks = st["kernel_seconds"]
        print(f"rect {size} k={k} W={W:5d}: {st['propagations']/ks:.3e} prop/s ({st['propagations']/ks/W:.3e} per worker), "
              f"{st['conflicts']/ks:.3e} confl/s ({st['conflicts']/ks/W:7.1f} per worker), {st['propagations']/max(1,st['bcp_steps']):.1f} prop/step, "
              f"{ks/max(1,st['bcp_steps'])*W*1e6:.2f} us/step/worker, alg {algorithmic_bytes(st)/ks/1e9:.1f} GB/s", flush=True)
        s.close()
