"""Where a rect M x M ladder's wall-clock goes: per solver instance the wall time inside solve / the sweep steps, the
kernel time and the simplification time (verbose lines of the library are passed through).  GPU box only."""
import sys
import time

sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, WorldGrid, solver_loop_sweep  # noqa: E402
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = WorldGrid.rect(m, m)
t = time.perf_counter()
e = Encoding.encode(PLATFORMS_DEFAULT, g)
print(f"encode {time.perf_counter() - t:.3f} s", flush=True)
for rep in range(2):
    made = []

    def mk():
        s = Mi355Sat(verbose=1 if rep == 1 else 0)
        made.append(s)
        return s

    t0 = time.perf_counter()
    hist = solver_loop_sweep(g, e, PlatformLimits({(1, 1): m}), out=lambda l: None, time_limit=100, make_solver=mk)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: total {dt:.3f} s; per record: " +
          ", ".join(f"k={h.get('k')} {h['result']} {h['seconds']:.3f}s kernel={h['stats'].get('kernel_seconds', 0):.3f}s" for h in hist), flush=True)
