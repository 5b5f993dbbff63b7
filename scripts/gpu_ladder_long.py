"""One long run of the product's batch loop on a harder rung (default: rect 28x28, CPU restatement 39 s). GPU box only."""
import sys
import time

sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, WorldGrid, solver_loop_sweep  # noqa: E402
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 28
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 280
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = WorldGrid.rect(m, m)
e = Encoding.encode(PLATFORMS_DEFAULT, g)
t0 = time.perf_counter()
hist = solver_loop_sweep(g, e, PlatformLimits({(1, 1): max(4, m * m // 24)}), out=lambda l: print(f"[{time.perf_counter()-t0:7.1f}s] {l}", flush=True),
                         time_limit=limit, make_solver=lambda: Mi355Sat(workers=workers, slice_ms=10))
print("history", [(h["k"], h["result"].name, h["count"]) for h in hist], f"total {time.perf_counter()-t0:.1f}s")
st = hist[-1]["stats"]
print({k: st[k] for k in ("conflicts", "propagations", "shared_exported", "shared_imported", "shared_imported_units", "kernel_seconds")})
