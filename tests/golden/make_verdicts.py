#!/opt/conda/bin/python3.9
"""Generates tests/golden/verdicts.json (run in the BUILD container only).

An independent complete solver — PicoSAT 0.6.3 through the `pycosat` binding that
happens to exist at /opt/conda/bin/python3.9 in this container — decides the CNFs
produced by oracle/encoder_oracle.py (+ into_cnf) for a ladder of (terrain,
platform set, k).  The verdicts pin the CPU CDCL restatement and the HIP solver
(SURVEY §8c (2)); PicoSAT is not on the GPU box, hence the committed fixture.

    /opt/conda/bin/python3.9 tests/golden/make_verdicts.py
"""
import json
import os
import sys
import time

import pycosat

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoder_oracle as eo  # noqa: E402

REF_TEST_DIR = "/root/reference/test"   # reference fixtures (inputs only, no expected outputs)

TERRAINS = {
    "ex1": [
        "XXXXX", "XXXXX", "X  XX", "X   X", "X    ", "XXX  "],
    "ex2": [
        "XX                   ", "XXXX                 ", "XXXXXX               ",
        "XXXXXXXXXXX          ", "XXXXXXXXXXXXXXXX     ", "XXXXXXXXXXXXXXXX     ",
        "XXXXXXXXXXXXXXXXX    ", "XXXXXXXXXXXXXXXXXX   ", "XXXXXXXX  XXXXXXXX   ",
        "XXXXXXXX  XXXXXXXXX  ", "XXXXXXXXXXXXXXXXXXX  ", "XXXXXXXXXXXXXXXXXXX  ",
        "XXXXXXXXXXXXXXXXXXXX ", "XXXXXXXXXXXXXXXXXXXX ", "XXXXXXXXXXX   XXXXXX ",
        "XXXXXXXXXXX   XXXXXXX"],
    "ex3": [
        " XXXXXXXXX ", "XXXXXXXXXXX", "XXXXXXXXXXX", "XXXXXXXXXXX", "XXXXXXXXXXX", "XXXXXXXXXXX",
        " XXXXXXXXX "],
}


def terrain(name):
    if name.startswith("rect"):
        w, h = name[4:].split("x")
        return eo.grid_rect(int(w), int(h))
    if os.path.isdir(REF_TEST_DIR):  # the committed rows above must equal the reference's files
        assert eo.grid_from_toml(os.path.join(REF_TEST_DIR, name + ".toml")) == eo.grid_from_rows(TERRAINS[name])
    return eo.grid_from_rows(TERRAINS[name])


# (terrain, platform set, list of k) — small enough for PicoSAT in seconds
LADDER = [
    ("ex1", "default", [0, 1, 2, 5]), ("ex1", "1x1", [1, 2, 3, 4, 8]),
    ("ex3", "default", [0, 1, 2, 20]), ("ex3", "1x1", [2, 3, 4, 5, 20]),
    ("rect8x8", "default", [0, 1, 2, 3, 20]), ("rect8x8", "1x1", [2, 3, 4, 5, 20]),
    ("rect16x16", "default", [2, 3, 4, 5, 40]), ("rect16x16", "1x1", [16, 18, 20, 40]),
    ("ex2", "default", [2, 3, 4, 5, 30]), ("ex2", "1x1", [13, 14, 15, 18, 30]),
    ("rect24x24", "default", [8, 9, 10, 12, 30]),
]


def main():
    out = []
    for name, pset, ks in LADDER:
        grid = terrain(name)
        defs = eo.PLATFORMS_DEFAULT if pset == "default" else [(1, 1)]
        enc = eo.Encoding(defs, grid)
        for k in ks:
            clauses, n_vars, cards = enc.with_limits({(1, 1): k})
            cnf, n_vars, _ = eo.into_cnf(clauses, n_vars, cards)
            t = time.time()
            res = pycosat.solve(cnf, vars=n_vars)
            dt = time.time() - t
            verdict = "UNSAT" if res == "UNSAT" else "SAT"
            count = None
            if verdict == "SAT":
                model = [0] * n_vars
                for l in res:
                    model[abs(l) - 1] = 1 if l > 0 else -1
                count = sum(1 for (x, y, d), v in enc.plat_var.items() if d == (1, 1) and model[v - 1] > 0)
                assert count <= k
            out.append({"terrain": name, "platforms": pset, "k": k, "verdict": verdict,
                        "n_vars": n_vars, "n_clauses": len(cnf), "picosat_seconds": round(dt, 3)})
            print(out[-1], flush=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "verdicts.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_verdicts.py", "solver": "PicoSAT 0.6.3 (pycosat)",
                   "terrains": TERRAINS, "verdicts": out}, f, indent=1)


if __name__ == "__main__":
    main()
