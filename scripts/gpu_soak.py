"""Soak: the bench workload (rect 64x64 sweep) for a given number of seconds with the default options;
fails loudly on any device-side error status (watch pool / learnt store exhaustion).  GPU box only."""
import sys
import time

sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, SolverResult, WorldGrid  # noqa: E402
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k_hi, k_lo = (51, 44) if size == 64 else (size * size // 40 + 4, size * size // 40 - 3)
g = WorldGrid.rect(size, size)
e = Encoding.encode(PLATFORMS_DEFAULT, g)
c = e.with_limits_into_cnf(PlatformLimits({(1, 1): k_hi}), sweep=True)
ks = list(range(k_hi, k_lo - 1, -1))
s = Mi355Sat(slice_ms=250)
s.add_cnf(c.lits, c.offsets)
s.sweep_begin([([-int(c.card_outputs[k])] if k < k_hi else []) for k in ks])
t0 = time.time()
last = 0
while time.time() - t0 < seconds:
    res, nd = s.sweep_step()          # raises SolverError on a device error status
    if time.time() - last > 10:
        st = s.stats()
        print(f"t={time.time()-t0:6.1f}s decided={nd} conflicts={st['conflicts']:.3e} props={st['propagations']:.3e} "
              f"learnts={st['learnts']:.3e} exported={st['shared_exported']} imported={st['shared_imported']}", flush=True)
        last = time.time()
    if nd == len(ks):
        break
s.sweep_end()
print("results", dict(zip(ks, [r.name for r in res])))
s.close()
print("SOAK OK")
