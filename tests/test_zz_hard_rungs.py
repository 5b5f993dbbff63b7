"""The optimum rungs of the ladders that carry BASELINE configs[2] ("rect 32 32 ... full CDCL to proven optimum") and the
first-UNSAT half of the metric, on the MI355X, against PicoSAT's verdicts (tests/golden/verdicts_hard.json, generator
tests/golden/make_verdicts_hard.py: rect 26 k = 10 / 11, rect 28 k = 11 / 12, rect 32 k = 14 / 15).

The file sorts last on purpose: these are the long ones (stated limits below; running into a limit fails the test), and
with `-x` a failure here cannot hide anything else.  CPU half: the fixture is well-formed and the oracle agrees with
PicoSAT on the one rung it decides in seconds.
"""
import threading

import pytest

from helpers import check_sat_answer, golden, make_grid, platform_defs
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, SolverResult

HARD = golden("verdicts_hard.json")["verdicts"]
# (terrain, k) -> stated time limit in seconds for the GPU.  Measured with the default fleet (1024 workers): round 2 rect 26
# k = 10 37-56 s, rect 28 k = 11 38-64 s, rect 32 k = 14 157-183 s; end of round 3 10-17 s, 10-17 s, 16-27 s (profiles/r03_*).
LIMITS = {("rect26x26", 10): 150, ("rect26x26", 11): 60, ("rect28x28", 11): 150, ("rect28x28", 12): 60, ("rect32x32", 14): 400}


def test_fixture_holds_the_three_optima():
    by = {(v["terrain"], v["k"]): v["verdict"] for v in HARD}
    for terrain, kstar in (("rect26x26", 11), ("rect28x28", 12), ("rect32x32", 15)):
        assert by[(terrain, kstar)] == "SAT" and by[(terrain, kstar - 1)] == "UNSAT"


def test_oracle_agrees_on_the_quick_rung():
    from oracle import oracle as ora
    v = next(v for v in HARD if (v["terrain"], v["k"]) == ("rect26x26", 11))
    grid = make_grid(v["terrain"])
    enc = Encoding.encode(platform_defs("default"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): v["k"]}))
    assert (cnf.n_vars, cnf.n_clauses) == (v["n_vars"], v["n_clauses"])      # the CNF PicoSAT decided
    o = ora.OracleSolver()
    o.add_cnf(cnf.lits, cnf.offsets)
    assert o.solve() == 10
    check_sat_answer(cnf, o.model(cnf.n_vars), enc, grid, v["k"])


@pytest.mark.gpu
@pytest.mark.parametrize("v", sorted((v for v in HARD if (v["terrain"], v["k"]) in LIMITS), key=lambda v: LIMITS[(v["terrain"], v["k"])]),
                         ids=lambda v: f"{v['terrain']}-k{v['k']}-{v['verdict']}")
def test_gpu_decides_the_optimum_rungs(v):
    grid = make_grid(v["terrain"])
    enc = Encoding.encode(platform_defs("default"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): v["k"]}))
    assert (cnf.n_vars, cnf.n_clauses) == (v["n_vars"], v["n_clauses"])
    s = Mi355Sat()
    s.add_cnf(cnf.lits, cnf.offsets)
    tm = threading.Timer(LIMITS[(v["terrain"], v["k"])], s.interrupter().interrupt)
    tm.start()
    try:
        r = s.solve()
    finally:
        tm.cancel()
    st = s.stats()
    print(f"{v['terrain']} k={v['k']}: {r.name} in {st['solve_seconds']:.1f} s, {st['conflicts']:.3e} conflicts, {st['workers']} workers")
    assert r.name.upper() == v["verdict"], (v, st["solve_seconds"])
    if r == SolverResult.Sat:
        check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, v["k"])
    s.close()
