#!/usr/bin/env python3
"""SURVEY 8d "reporting": every BASELINE config in both platform modes as one JSON artifact (profiles/rNN_configs.json).

Per config x {default platforms, 1x1 supports only}: verdict per k, k*, wall-clock, propagations, propagations/s,
conflicts, algorithmic bytes/s (12*n_deq + 9*n_watch + 5*n_cl_lit + 8*n_move + 13*n_enq over kernel time), the time
limit and whether it was hit.  GPU side: the product's loop (solver_loop_sweep, defaults); CPU side where SURVEY asks for
it (C1: the whole loop; C2: BCP) - the oracle's CDCL restatement, 1 core, labelled as such (it is not rustsat-glucose).

    python3 scripts/report_configs.py gpurun_out/r03_configs.json [--limit-c3 400]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from timberborn_support_solver_amd import (PLATFORMS_DEFAULT, Encoding, Mi355Sat, PlatformLayout, PlatformLimits,  # noqa: E402
                                           SolverResult, WorldGrid, algorithmic_bytes, solver_loop_sweep)

EX3 = [" XXXXXXXXX ", "XXXXXXXXXXX", "XXXXXXXXXXX", "XXXXXXXXXXX", "XXXXXXXXXXX", "XXXXXXXXXXX", " XXXXXXXXX "]


def rec_of(h):
    st = h.get("stats") or {}
    ks = float(st.get("kernel_seconds", 0.0)) or None
    return {"k": h["k"], "verdict": h["result"].name.upper() if hasattr(h["result"], "name") else str(h["result"]),
            "count": h["count"], "valid": h["valid"], "seconds": round(h["seconds"], 4),
            "propagations": int(st.get("propagations", 0)), "conflicts": int(st.get("conflicts", 0)),
            "kernel_seconds": ks, "propagations_per_s": (st.get("propagations", 0) / ks) if ks else None,
            "algorithmic_bytes_per_s": (algorithmic_bytes(st) / ks) if ks and st else None, "workers": int(st.get("workers", 0))}


def gpu_loop(grid, defs, k0, limit):
    enc = Encoding.encode(defs, grid)
    t0 = time.perf_counter()
    hist = solver_loop_sweep(grid, enc, PlatformLimits({(1, 1): k0}), out=lambda line: None, time_limit=limit)
    dt = time.perf_counter() - t0
    sat = [h for h in hist if h["result"] == SolverResult.Sat]
    proven = bool(sat) and hist[-1]["result"] == SolverResult.Unsat
    return {"iterations": [rec_of(h) for h in hist], "optimum_k": sat[-1]["count"] if proven else None,
            "best_count": sat[-1]["count"] if sat else None, "first_unsat_k": hist[-1]["k"] if proven else None,
            "wall_clock_s": round(dt, 3), "time_limit_s": limit, "hit_time_limit": hist[-1]["result"] == SolverResult.Interrupted,
            "all_layouts_valid": all(h["valid"] for h in sat)}


def cpu_loop(grid, defs, k0, limit):
    from oracle import oracle as ora
    enc = Encoding.encode(defs, grid)
    t0, k, its, kstar = time.perf_counter(), k0, [], None
    while time.perf_counter() - t0 < limit:
        ck = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
        o = ora.OracleSolver()
        o.add_cnf(ck.lits, ck.offsets)
        t = time.perf_counter()
        r = o.solve(conflict_budget=3_000_000)
        st = o.stats()
        cnt = PlatformLayout.from_assignment(o.model(ck.n_vars)[:enc.n_vars], enc).platform_count() if r == 10 else None
        its.append({"k": k, "verdict": {10: "SAT", 20: "UNSAT"}.get(r, "INTERRUPTED"), "count": cnt, "seconds": round(time.perf_counter() - t, 4),
                    "propagations": st["propagations"], "conflicts": st["conflicts"]})
        if r == 20:
            kstar = k + 1
        if r != 10 or cnt == 0:
            break
        k = cnt - 1
    return {"iterations": its, "optimum_k": kstar, "wall_clock_s": round(time.perf_counter() - t0, 3), "time_limit_s": limit,
            "kind": "port (oracle CDCL restatement, 1 core; not rustsat-glucose)"}


def bcp_compare(grid, defs, k0):
    """C2: single-instance BCP from the unit clauses and seeded decision scripts, GPU vs CPU, fixpoints compared."""
    import numpy as np
    from helpers import scripted_decisions
    from oracle import oracle as ora
    enc = Encoding.encode(defs, grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k0}))
    scripts = [[]] + [scripted_decisions(enc, grid, seed, 12, p_positive=0.15) for seed in range(1, 17)]
    s = Mi355Sat()
    s.add_cnf(cnf.lits, cnf.offsets)
    confl, vals, tl = s.propagate_batch(scripts, n_vars=cnf.n_vars, repeat=50)
    st = s.stats()
    s.close()
    t0 = time.perf_counter()
    same, deq = True, 0
    for _ in range(5):
        for i, dec in enumerate(scripts):
            c, v, n, d = ora.bcp(cnf.lits, cnf.offsets, cnf.n_vars, dec)
            deq += d
            same = same and c == confl[i] and (c or (n == tl[i] and np.array_equal(v, vals[i])))
    dtc = time.perf_counter() - t0
    return {"scripts": len(scripts), "fixpoints_bit_exact": bool(same), "gpu_propagations": int(st["propagations"]),
            "gpu_kernel_seconds": st["kernel_seconds"], "gpu_propagations_per_s": st["propagations"] / max(st["kernel_seconds"], 1e-9),
            "gpu_algorithmic_bytes_per_s": algorithmic_bytes(st) / max(st["kernel_seconds"], 1e-9),
            "cpu_propagations_per_s": deq / max(dtc, 1e-9), "cpu_kind": "oracle/check.c occurrence-list BCP, 1 core",
            "note": "17 instances of one small formula: a latency measurement of the BCP kernel, not a throughput one"}


def sweep64(defs, slices=20, warmup=5):
    """C5 on one GPU: the bench workload (at-most-k for k = 51..44 over one CNF, 4096 workers, 250 ms slices)."""
    grid = WorldGrid.rect(64, 64)
    enc = Encoding.encode(defs, grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 51}), sweep=True)
    ks = list(range(51, 43, -1))
    sets = [([-int(cnf.card_outputs[k])] if k < 51 else []) for k in ks]
    s = Mi355Sat(workers=4096, slice_ms=250, share=-1, ramp=-1)
    s.add_cnf(cnf.lits, cnf.offsets)
    s.reserve(cnf.n_vars)
    s.sweep_begin(sets)
    for _ in range(warmup):
        s.sweep_step()
    st0, t0 = s.stats(), time.perf_counter()
    for _ in range(slices):
        res, _ = s.sweep_step()
    dt = time.perf_counter() - t0
    st1 = s.stats()
    s.sweep_end()
    s.close()
    d = {k: st1[k] - st0[k] for k in ("propagations", "conflicts", "n_deq", "n_watch", "n_cl_lit", "n_move", "n_enq", "kernel_seconds")}
    return {"bounds": ks, "verdict_per_k": {str(k): r.name.upper() for k, r in zip(ks, res)}, "optimum_k": None,
            "note": "no bound is decided within the measured slices (area bound k* >= 43; out of reach for any side, SURVEY 6)",
            "slices": slices, "wall_clock_s": round(dt, 3), "propagations": int(d["propagations"]), "conflicts": int(d["conflicts"]),
            "propagations_per_s": d["propagations"] / dt, "algorithmic_bytes_per_s": algorithmic_bytes(d) / max(d["kernel_seconds"], 1e-9),
            "vars": int(cnf.n_vars), "clauses": int(cnf.n_clauses)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--limit-c3", type=float, default=400.0)
    ap.add_argument("--limit-1x1", type=float, default=60.0)
    ap.add_argument("--skip", default="", help="comma list of config names to skip")
    a = ap.parse_args()
    skip = set(x for x in a.skip.split(",") if x)
    out = {"generator": "scripts/report_configs.py", "configs": {}}
    modes = {"default": PLATFORMS_DEFAULT, "1x1": [(1, 1)]}

    def put(name, mode, key, fn):
        if name in skip:
            return
        t = time.perf_counter()
        out["configs"].setdefault(name, {}).setdefault(mode, {})[key] = fn()
        print(f"[report] {name} {mode} {key}: {time.perf_counter() - t:.1f} s", file=sys.stderr, flush=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)

    for mode, defs in modes.items():
        put("C1 rect 8 8 k0=20", mode, "gpu", lambda: gpu_loop(WorldGrid.rect(8, 8), defs, 20, 60))
        put("C1 rect 8 8 k0=20", mode, "cpu", lambda: cpu_loop(WorldGrid.rect(8, 8), defs, 20, 60))
        put("C2 rect 16 16 k0=40", mode, "bcp", lambda: bcp_compare(WorldGrid.rect(16, 16), defs, 40))
        put("C2 rect 16 16 k0=40", mode, "gpu", lambda: gpu_loop(WorldGrid.rect(16, 16), defs, 40, 90))
        put("C4 ex3 k0=20", mode, "gpu", lambda: gpu_loop(WorldGrid.from_rows(EX3), defs, 20, 60))
        put("C5 rect 64 64 k-sweep", mode, "gpu", lambda: sweep64(defs))
        put("C3 rect 32 32 k0=120", mode, "gpu",
            lambda: gpu_loop(WorldGrid.rect(32, 32), defs, 120, a.limit_c3 if mode == "default" else a.limit_1x1))
    print(json.dumps({k: {m: {kk: (vv.get("optimum_k"), vv.get("wall_clock_s")) for kk, vv in mv.items()} for m, mv in v.items()}
                      for k, v in out["configs"].items()}))


if __name__ == "__main__":
    main()
