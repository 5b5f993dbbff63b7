#!/usr/bin/env python3
"""Ad-hoc exploration on the GPU box: time-to-verdict on mid-size rungs, GPU vs the CPU oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timberborn_support_solver_amd import *
from oracle import oracle as ora

def run(terrain, pset, k, workers, slice_conflicts=500, cpu=True, budget=0, **kw):
    w, h = (int(v) for v in terrain[4:].split("x"))
    grid = WorldGrid.rect(w, h)
    enc = Encoding.encode(PLATFORMS_DEFAULT if pset == "default" else [(1, 1)], grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    s = Mi355Sat(workers=workers, slice_conflicts=slice_conflicts, verbose=int(os.environ.get('VERBOSE', '0')), conflict_budget=budget, **kw)
    s.add_cnf(cnf.lits, cnf.offsets)
    t = time.time(); r = s.solve(); dt = time.time() - t
    st = s.stats()
    print(f"GPU {terrain} {pset} k={k} W={workers}: {r.name} wall={dt:.2f}s kernel={st['kernel_seconds']:.2f}s conflicts={st['conflicts']} "
          f"props={st['propagations']} props/s={st['propagations']/max(st['kernel_seconds'],1e-9):.3e} "
          f"alg GB/s={algorithmic_bytes(st)/max(st['kernel_seconds'],1e-9)/1e9:.1f} B/prop={algorithmic_bytes(st)/max(1,st['propagations']):.0f} launches={st['kernel_launches']} "
          f"lits/step={st['propagations']/max(1,st['bcp_steps']):.2f} requeued={st['bcp_requeued']}", flush=True)
    s.close()
    if cpu:
        o = ora.OracleSolver(); o.add_cnf(cnf.lits, cnf.offsets)
        t = time.time(); ro = o.solve(conflict_budget=400000); dto = time.time() - t
        so = o.stats()
        print(f"CPU {terrain} {pset} k={k}: {ro} wall={dto:.2f}s conflicts={so['conflicts']} props={so['propagations']} props/s={so['propagations']/max(dto,1e-9):.3e}", flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["a"]
    if "a" in which:
        run("rect16x16", "default", 3, 64)
        run("rect24x24", "default", 8, 256)
        run("rect24x24", "default", 9, 256)
        run("rect16x16", "1x1", 16, 256)
        run("rect32x32", "default", 24, 256)
        run("rect32x32", "default", 10, 256)
    if "b" in which:
        for W in (1, 64, 256, 1024, 2048):
            run("rect24x24", "default", 8, W, cpu=False)
    if "d" in which:
        for kw in (dict(lds_val=-1, max_groups=1), dict(lds_val=-1, max_groups=8), dict(lds_val=1, max_groups=1), dict(lds_val=1, max_groups=8)):
            print(kw, flush=True)
            run("rect24x24", "default", 8, 256, cpu=False, **kw)
            run("rect64x64", "default", 46, 1280, slice_conflicts=100, cpu=False, budget=1280 * 100, **kw)
    if "w" in which:   # wall-clock to verdict: cube splitting (default) vs plain portfolio
        for name, pset, k in [("rect24x24", "default", 8), ("rect24x24", "default", 9), ("rect16x16", "1x1", 14), ("rect32x32", "default", 12), ("rect32x32", "default", 13)]:
            run(name, pset, k, 3072, slice_conflicts=0, cpu=False)
        run("rect24x24", "default", 8, 3072, slice_conflicts=0, cpu=False, cube_split=-1)
    if "c" in which:
        for W in (256, 1024, 2048, 4096):
            run("rect64x64", "default", 46, W, slice_conflicts=100, cpu=False, budget=W * 100)
