"""A/B of the learnt-clause exchange on the wall-clock-to-first-UNSAT ladder (GPU box only):
the whole sweep k0..0 of rect m x m as one batch until the cut closes, exchange off / on."""
import argparse
import json
import sys
import time

sys.path.insert(0, ".")
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, SolverResult, WorldGrid  # noqa: E402
from timberborn_support_solver_amd.encoder import PLATFORMS_DEFAULT  # noqa: E402


def run(m, share, share_lbd, workers, slice_ms, limit, drop=True, rebalance=0, share_interval=0, cube_split=0, share_len=0):
    g = WorldGrid.rect(m, m)
    e = Encoding.encode(PLATFORMS_DEFAULT, g)
    k0 = max(4, m * m // 24)
    c = e.with_limits_into_cnf(PlatformLimits({(1, 1): k0}), sweep=True)
    ks = list(range(k0, -1, -1))
    sets = [([-int(c.card_outputs[k])] if k < k0 else []) for k in ks]
    sv = Mi355Sat(workers=max(len(ks), workers // len(ks) * len(ks)), slice_ms=slice_ms, share=share, share_lbd=share_lbd, rebalance=rebalance, share_interval=share_interval, cube_split=cube_split, share_len=share_len)
    sv.add_cnf(c.lits, c.offsets)
    t0 = time.perf_counter()
    sv.sweep_begin(sets)
    kstar = None
    while time.perf_counter() - t0 < limit:
        res, _ = sv.sweep_step()
        sat_k = min([k for k, r in zip(ks, res) if r == SolverResult.Sat], default=None)
        unsat_k = max([k for k, r in zip(ks, res) if r == SolverResult.Unsat], default=None)
        if sat_k is not None and unsat_k is not None and unsat_k + 1 >= sat_k:
            kstar = sat_k
            break
        if drop:   # implied answers: every k above a SAT one, every k below an UNSAT one
            sv.sweep_drop([i for i, k in enumerate(ks) if res[i] == SolverResult.Interrupted and
                           ((sat_k is not None and k > sat_k) or (unsat_k is not None and k < unsat_k))])
    dt = time.perf_counter() - t0
    sv.sweep_end()
    st = sv.stats()
    sv.close()
    return {"m": m, "cube_split": cube_split, "workers": workers, "rebalance": rebalance, "share": share, "interval": share_interval, "share_lbd": share_lbd, "share_len": share_len, "kstar": kstar, "seconds": round(dt, 3), "conflicts": st["conflicts"],
            "propagations": st["propagations"], "exported": st["shared_exported"], "imported": st["shared_imported"],
            "imported_units": st["shared_imported_units"], "kernel_s": round(st["kernel_seconds"], 3)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="16,20,24")
    ap.add_argument("--workers", type=int, default=4096)
    ap.add_argument("--slice-ms", type=int, default=10)
    ap.add_argument("--limit", type=float, default=60)
    ap.add_argument("--modes", default="-1:0,0:2,0:4", help="share:share_lbd[:rebalance[:workers[:interval[:cube_split]]]] ...")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU restatement's sequential decreasing-k loop")
    a = ap.parse_args()
    for m in [int(x) for x in a.sizes.split(",")]:
        if a.cpu:
            from oracle import oracle as ora
            from timberborn_support_solver_amd import PlatformLayout
            g = WorldGrid.rect(m, m)
            e = Encoding.encode(PLATFORMS_DEFAULT, g)
            k, tc, kstar, confl = max(4, m * m // 24), time.perf_counter(), None, 0
            while time.perf_counter() - tc < a.limit * 3:
                ck = e.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
                o = ora.OracleSolver()
                o.add_cnf(ck.lits, ck.offsets)
                r = o.solve(conflict_budget=20_000_000)
                confl += o.stats()["conflicts"]
                if r == 20:
                    kstar = k + 1
                    break
                if r != 10:
                    break
                k = PlatformLayout.from_assignment(o.model(ck.n_vars)[:e.n_vars], e).platform_count() - 1
            print(json.dumps({"m": m, "cpu_seconds": round(time.perf_counter() - tc, 3), "kstar": kstar, "conflicts": confl}), flush=True)
        for mode in a.modes.split(","):
            f = [int(x) for x in mode.split(":")] + [0, 0, 0, 0, 0]
            print(json.dumps(run(m, f[0], f[1], f[3] or a.workers, a.slice_ms, a.limit, rebalance=f[2], share_interval=f[4],
                                 cube_split=f[5], share_len=f[6])), flush=True)
