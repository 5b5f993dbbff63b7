"""EXPERIMENT: does cube-and-conquer pay on the support-grid CNFs?  CPU only (oracle CDCL as the conquer
solver).  usage: cc_cpu.py SIZE K DEPTH NCAND FRAC"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as ora
from timberborn_support_solver_amd import PLATFORMS_DEFAULT, Encoding, PlatformLimits, WorldGrid

n, k, depth, ncand, frac = (int(x) for x in sys.argv[1:6])
plat = PLATFORMS_DEFAULT if len(sys.argv) < 7 else [(1, 1)]
grid = WorldGrid.rect(n, n)
enc = Encoding.encode(plat, grid)
cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
print("vars", cnf.n_vars, "clauses", cnf.n_clauses, flush=True)
here = os.path.dirname(os.path.abspath(__file__))
path = f"/dev/shm/cc_{n}_{k}.bin"
with open(path, "wb") as f:
    np.array([cnf.n_vars, cnf.n_clauses], dtype=np.int64).tofile(f)
    np.asarray(cnf.offsets, dtype=np.uint64).tofile(f)
    np.asarray(cnf.lits, dtype=np.int32).tofile(f)
if os.environ.get("BASE", "1") == "1":
    o = ora.OracleSolver(); o.add_cnf(cnf.lits, cnf.offsets)
    t = time.perf_counter(); r = o.solve(conflict_budget=int(os.environ.get("BUDGET", "3000000"))); dt = time.perf_counter() - t
    print("baseline", r, o.stats()["conflicts"], "conflicts", f"{dt:.2f}s", flush=True)
t = time.perf_counter()
out = subprocess.run([os.path.join(here, "cubes"), path, str(depth), str(ncand), str(frac), os.environ.get("CRH", "0")], capture_output=True, text=True)
print(out.stderr.strip(), f"cube time {time.perf_counter() - t:.2f}s", flush=True)
cubes = [[int(x) for x in line.split()[:-1]] for line in out.stdout.splitlines()]
confl = []
t = time.perf_counter()
for mode in ("fresh", "shared"):
    o = None
    confl = []; res = {10: 0, 20: 0, 0: 0}
    t = time.perf_counter()
    for c in cubes:
        if mode == "fresh" or o is None:
            o = ora.OracleSolver(); o.add_cnf(cnf.lits, cnf.offsets); c0 = 0
        else:
            c0 = o.stats()["conflicts"]
        r = o.solve(c, conflict_budget=2000000)
        res[r] += 1
        confl.append(o.stats()["conflicts"] - c0)
    confl = np.array(confl)
    print(mode, "cubes", len(cubes), res, "total conflicts", confl.sum(), "max", confl.max() if len(confl) else 0, "median", np.median(confl) if len(confl) else 0,
          f"{time.perf_counter() - t:.2f}s", flush=True)
os.remove(path)
