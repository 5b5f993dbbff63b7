"""Wall-clock of the refinement loop `-l1:M` on rect M x M: the product's batch loop on the GPU (3 runs) against
the CPU restatement's sequential loop on the GPU box's host (1 core).  GPU box only."""
import os
import sys
import time

sys.path.insert(0, ".")
from oracle import oracle as ora  # noqa: E402  (diagnostic script: the CPU side is the checker's solver)
from timberborn_support_solver_amd import (Encoding, Mi355Sat, PlatformLayout, PlatformLimits, SolverResult, WorldGrid,  # noqa: E402
                                           solver_loop_sweep)
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 20, 24, 26]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
limit = float(sys.argv[3]) if len(sys.argv) > 3 else 150
cpu_limit = float(sys.argv[4]) if len(sys.argv) > 4 else 240
_la = int(os.environ.get("LOOKAHEAD", "0"))     # 1: the last bounds two at a time (loop.py::solver_loop_pair); N >= 3: N at a time (solver_loop_fan)
lookahead = False if _la == 0 else (True if _la <= 2 else _la)
KW = dict(kv.split("=") for kv in os.environ.get("SOLVER_OPTS", "").split(",") if kv)
KW = {k: int(v) for k, v in KW.items()}
for m in sizes:
    g = WorldGrid.rect(m, m)
    e = Encoding.encode(PLATFORMS_DEFAULT, g)
    k, tc, kstar, confl, cpu_rungs = m, time.perf_counter(), None, 0, []
    while time.perf_counter() - tc < cpu_limit:
        ck = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
        o = ora.OracleSolver()
        o.add_cnf(ck.lits, ck.offsets)
        tr = time.perf_counter()
        r = o.solve(conflict_budget=20_000_000)
        confl += o.stats()["conflicts"]
        cpu_rungs.append((k, {10: "Sat", 20: "Unsat"}.get(r, "Interrupted"), round(time.perf_counter() - tr, 1)))
        if r == 20:
            kstar = k + 1
            break
        if r != 10:
            break
        k = PlatformLayout.from_assignment(o.model(ck.n_vars)[:e.n_vars], e).platform_count() - 1
    cpu_s = time.perf_counter() - tc
    print(f"rect {m}: CPU done in {cpu_s:.1f} s: {cpu_rungs}", flush=True)   # (keeps a long rung from looking hung)
    gpu = []
    for rep in range(reps):
        t0 = time.perf_counter()
        hist = solver_loop_sweep(g, e, PlatformLimits({(1, 1): m}), out=lambda l: None, time_limit=limit, lookahead=lookahead,
                                 make_solver=lambda half=False: Mi355Sat(**({"workers": 512} if half else {}), **KW))
        ok = hist[-1]["result"] == SolverResult.Unsat
        gpu.append((round(time.perf_counter() - t0, 2), [h["count"] for h in hist if h["count"]][-1] if ok else None,
                    [(h["k"], h["result"].name, h["count"], round(h["seconds"], 1)) for h in hist]))
        print(f"rect {m}: GPU run {rep} {gpu[-1]}", flush=True)
    print(f"rect {m} -l1:{m}: CPU {cpu_s:.2f} s (k*={kstar}, {confl} conflicts) | GPU batch loop {gpu}", flush=True)
