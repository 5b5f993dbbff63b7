#!/opt/conda/bin/python3.9
"""Generates tests/golden/verdicts_hard.json (run in the BUILD container only; about 25 minutes on 6 cores).

The optimum rungs of the ladders that carry BASELINE configs[2] and the first-UNSAT half of the metric: rect 26x26
k = 10 / 11, rect 28x28 k = 11 / 12, rect 32x32 k = 14 / 15 with the default platform set.  Same method as
make_verdicts.py: PicoSAT 0.6.3 (pycosat; present only in the build container) decides the CNF of
oracle/encoder_oracle.py (+ into_cnf); one process per rung.  Kept in its own file because the rungs take
PicoSAT minutes (rect 32 k = 14 UNSAT 515 s, k = 15 SAT 1253 s) while verdicts.json regenerates in seconds.

    /opt/conda/bin/python3.9 tests/golden/make_verdicts_hard.py
"""
import json
import multiprocessing as mp
import os
import sys
import time

import pycosat

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoder_oracle as eo  # noqa: E402

RUNGS = [("rect32x32", 15), ("rect32x32", 14), ("rect28x28", 11), ("rect28x28", 12), ("rect26x26", 10), ("rect26x26", 11)]


def run(job):
    name, k = job
    w, h = name[4:].split("x")
    enc = eo.Encoding(eo.PLATFORMS_DEFAULT, eo.grid_rect(int(w), int(h)))
    clauses, n_vars, cards = enc.with_limits({(1, 1): k})
    cnf, n_vars, _ = eo.into_cnf(clauses, n_vars, cards)
    t = time.time()
    res = pycosat.solve(cnf, vars=n_vars)
    dt = time.time() - t
    verdict = "UNSAT" if res == "UNSAT" else "SAT"
    if verdict == "SAT":
        model = [0] * n_vars
        for l in res:
            model[abs(l) - 1] = 1 if l > 0 else -1
        count = sum(1 for (x, y, d), v in enc.plat_var.items() if d == (1, 1) and model[v - 1] > 0)
        assert count <= k
    rec = {"terrain": name, "platforms": "default", "k": k, "verdict": verdict, "n_vars": n_vars,
           "n_clauses": len(cnf), "picosat_seconds": round(dt, 3)}
    print(rec, flush=True)
    return rec


def main():
    with mp.Pool(min(6, len(RUNGS))) as pool:
        out = pool.map(run, RUNGS, chunksize=1)
    out.sort(key=lambda r: (int(r["terrain"][4:].split("x")[0]), r["k"]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "verdicts_hard.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_verdicts_hard.py", "solver": "PicoSAT 0.6.3 (pycosat)",
                   "verdicts": out}, f, indent=1)


if __name__ == "__main__":
    main()
