#!/usr/bin/env python3
"""The batch loop of solver_loop_sweep on rect M x M with a progress line every few seconds: which bounds are open,
the cut so far, conflicts.  usage: gpu_ladder_debug.py M LIMIT [k_hi]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from timberborn_support_solver_amd import PLATFORMS_DEFAULT, Encoding, Mi355Sat, PlatformLayout, PlatformLimits, SolverResult, WorldGrid
from timberborn_support_solver_amd.loop import frontier_weights
m, limit = int(sys.argv[1]), float(sys.argv[2])
g = WorldGrid.rect(m, m)
e = Encoding.encode(PLATFORMS_DEFAULT, g)
k0 = int(sys.argv[3]) if len(sys.argv) > 3 else m - 1
weighted = os.environ.get("WEIGHTS", "1") == "1"
cnf = e.with_limits_into_cnf(PlatformLimits({(1, 1): k0}), sweep=True)
ks = list(range(k0, -1, -1))
sets = [([-int(cnf.card_outputs[k])] if k < len(cnf.card_outputs) else []) for k in ks]
s = Mi355Sat()
s.add_cnf(cnf.lits, cnf.offsets); s.reserve(cnf.n_vars)
t0 = time.perf_counter()
s.sweep_begin(sets)
print(f"rect {m} batch k={k0}..0: vars {cnf.n_vars} clauses {cnf.n_clauses}; begin {time.perf_counter()-t0:.2f}s", flush=True)
best_c, unsat_k, looked, last = None, -1, set(), 0
while time.perf_counter() - t0 < limit:
    res, nd = s.sweep_step()
    for i, r in enumerate(res):
        if r == SolverResult.Unsat: unsat_k = max(unsat_k, ks[i])
        elif r == SolverResult.Sat and i not in looked:
            looked.add(i)
            c = PlatformLayout.from_assignment(s.sweep_solution_of(i, e.n_vars), e).platform_count()
            best_c = c if best_c is None else min(best_c, c)
            print(f"[{time.perf_counter()-t0:6.1f}s] SAT at k={ks[i]} with {c} platforms", flush=True)
    if best_c is not None and unsat_k + 1 >= best_c: break
    s.sweep_drop([i for i, k in enumerate(ks) if res[i] == SolverResult.Interrupted and ((best_c is not None and k >= best_c) or k < unsat_k)])
    if weighted: s.sweep_set_weights(frontier_weights(ks, res, best_c, unsat_k))
    if time.perf_counter() - last > 5:
        last = time.perf_counter()
        st = s.stats()
        open_ks = [k for k, r in zip(ks, res) if r == SolverResult.Interrupted and (best_c is None or k < best_c) and k > unsat_k]
        print(f"[{last-t0:6.1f}s] open {open_ks} best {best_c} unsat<= {unsat_k} conflicts {st['conflicts']:.3e} exp {st['shared_exported']} imp {st['shared_imported']:.3e} workers {st['workers']}", flush=True)
print(f"done in {time.perf_counter()-t0:.1f}s: best {best_c} unsat {unsat_k}", flush=True)
s.sweep_end(); s.close()
