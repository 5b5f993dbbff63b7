"""Pins the oracle itself: PicoSAT golden verdicts, self-certifying models, RUP-checked refutations."""
import numpy as np
import pytest

from helpers import VERDICTS, make_grid, platform_defs, scripted_decisions
from oracle import oracle as ora
from timberborn_support_solver_amd import Encoding, PlatformLimits

CASES = [v for v in VERDICTS["verdicts"] if v["picosat_seconds"] < 1.0]


@pytest.mark.parametrize("v", CASES, ids=lambda v: f"{v['terrain']}-{v['platforms']}-k{v['k']}")
def test_cdcl_restatement_agrees_with_picosat(v):
    grid = make_grid(v["terrain"])
    enc = Encoding.encode(platform_defs(v["platforms"]), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): v["k"]}))
    assert (cnf.n_vars, cnf.n_clauses) == (v["n_vars"], v["n_clauses"])   # same CNF PicoSAT decided
    s = ora.OracleSolver()
    s.add_cnf(cnf.lits, cnf.offsets)
    if v["n_clauses"] < 8000:
        s.enable_proof()
    r = s.solve()
    assert {10: "SAT", 20: "UNSAT"}[r] == v["verdict"]
    if r == 10:
        assert ora.check_model(cnf.lits, cnf.offsets, s.model(cnf.n_vars)) == -1
    elif v["n_clauses"] < 8000:
        assert ora.check_rup(cnf.lits, cnf.offsets, cnf.n_vars, s.proof()) == 1


def test_rup_checker_rejects_a_bogus_proof():
    lits, offs = ora.to_csr([[1, 2], [-1, 2], [1, -2], [-1, -2]])
    assert ora.check_rup(lits, offs, 2, np.array([2, 0, 0], dtype=np.int32)) == 1
    lits, offs = ora.to_csr([[1, 2], [-1, 2], [1, -2]])            # satisfiable
    assert ora.check_rup(lits, offs, 2, np.array([-2, 0, 0], dtype=np.int32)) == 0


def test_bcp_checker_small_cases():
    lits, offs = ora.to_csr([[1], [-1, 2], [-2, 3, 4], [-4]])
    c, vals, n, _ = ora.bcp(lits, offs, 5, [])
    assert c == 0 and vals.tolist() == [1, 1, 1, -1, 0] and n == 4
    c, *_ = ora.bcp(lits, offs, 5, [-3])
    assert c == 1
    c, vals, n, _ = ora.bcp(lits, offs, 5, [5, 3])                  # 3 already true: skipped
    assert c == 0 and vals.tolist() == [1, 1, 1, -1, 1]
    lits, offs = ora.to_csr([[1], [-1]])
    assert ora.bcp(lits, offs, 1, [])[0] == 1
    lits, offs = ora.to_csr([])
    assert ora.bcp(lits, offs, 3, [])[0] == 0


def test_budget_and_assumptions():
    grid = make_grid("rect16x16")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 14}))
    s = ora.OracleSolver()
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve(conflict_budget=200) == 0                        # interrupted by budget
    st = s.stats()
    assert st["conflicts"] >= 200 and st["n_terminated"] == 1 and st["propagations"] == st["n_deq"] > 0
