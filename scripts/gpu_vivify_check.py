#!/usr/bin/env python3
"""Vivification on the GPU: the DRUP proof of a multi-worker run with it (checked by the oracle), then time to the verdict
with and without.  usage: gpu_vivify_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_grid, platform_defs
from oracle import oracle as ora
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, SolverResult
from timberborn_support_solver_amd.dimacs import read_drup
grid = make_grid("rect16x16"); enc = Encoding.encode(platform_defs("1x1"), grid)
cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 10}))
for viv in (16,):
    s = Mi355Sat(workers=16, slice_ms=5, vivify=viv, verbose=1)
    s.set_proof_path("/tmp/p.drup"); s.add_cnf(cnf.lits, cnf.offsets)
    t = time.time(); r = s.solve(); dt = time.time() - t
    print("proof run: vivify", viv, r.name, f"{dt:.1f}s conflicts={s.stats()['conflicts']}", flush=True)
    s.close()
    t = time.time()
    print("  RUP check:", ora.check_rup(cnf.lits, cnf.offsets, cnf.n_vars, read_drup("/tmp/p.drup")), f"{time.time()-t:.1f}s", flush=True)
