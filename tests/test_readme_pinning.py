"""Reference-held known answers: the four layouts printed in the reference's README
(/root/reference/README.md:46-116 -> tests/golden/readme_layouts.json: 18 / 17 / 16 / 15 supports of
size 1x1 on a 21x16 terrain) are the only solver outputs the reference publishes.  They pin

  * the encoder (product CNF and the oracle's literal restatement): each printed layout, stated as
    assumptions on the 1x1 placement variables, must be a model of the CNF with at-most-k for k = its
    own count  =>  SAT, and one support fewer than printed (k = count - 1) must contradict it;
  * the solver: `solve()` at k = 18, 17, 16, 15 on that terrain must answer SAT (README.md:46-116 shows
    the reference's solver doing exactly that) with a model that satisfies every clause and validates.

The CPU half (oracle solver on both CNFs) runs everywhere; the GPU half goes through the C ABI.
"""
import numpy as np
import pytest

from helpers import check_sat_answer, emu_lib, golden
from oracle import encoder_oracle as eo, oracle as ora
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, SolverResult, WorldGrid

README = golden("readme_layouts.json")["layouts"]


def layout_assumptions(lay, var_of):
    """Every tile's 1x1 placement variable, true exactly on the printed supports."""
    rows = lay["terrain_rows"]
    sup = {tuple(p) for p in lay["supports_xy"]}
    return [var_of(x, y) if (x, y) in sup else -var_of(x, y) for y in range(len(rows)) for x in range(len(rows[0]))]


@pytest.mark.parametrize("lay", README, ids=lambda l: f"{l['marked']}-supports")
def test_printed_layouts_are_models_of_product_and_oracle_cnf(lay):
    rows, k = lay["terrain_rows"], lay["marked"]
    grid = WorldGrid.from_rows(rows)
    enc = Encoding.encode([(1, 1)], grid)
    o = eo.Encoding([(1, 1)], eo.grid_from_rows(rows))
    for kk, want in [(k, 10), (k - 1, 20)]:
        # product CNF (C++ encoder + totalizer)
        cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): kk}))
        a = layout_assumptions(lay, lambda x, y: enc.platform_var(x, y, (1, 1)))
        s = ora.OracleSolver()
        s.add_cnf(cnf.lits, cnf.offsets)
        assert s.solve(a) == want
        if want == 10:
            lay_out = check_sat_answer(cnf, s.model(cnf.n_vars), enc, grid, kk)
            assert sorted((x, y) for x, y, *_ in lay_out.platforms()) == sorted(tuple(p) for p in lay["supports_xy"])
        # the kernel logic (wavefront emulator build of the product library) on the product CNF
        e = Mi355Sat(_lib_override=emu_lib(), workers=2, simp=-1)
        e.add_cnf(cnf.lits, cnf.offsets)
        assert e.solve_batch([a])[0].value == want
        e.close()
        # the oracle's literal restatement of encoder.rs
        cl, nv, cards = o.with_limits({(1, 1): kk})
        ocl, onv, _ = eo.into_cnf(cl, nv, cards)
        lits, offs = ora.to_csr(ocl)
        s2 = ora.OracleSolver()
        s2.add_cnf(lits, offs)
        s2.reserve(onv)
        assert s2.solve(layout_assumptions(lay, lambda x, y: o.plat_var[(x, y, (1, 1))])) == want


@pytest.mark.gpu
@pytest.mark.parametrize("lay", README, ids=lambda l: f"{l['marked']}-supports")
def test_gpu_accepts_printed_layouts_and_solves_the_readme_rungs(lay):
    rows, k = lay["terrain_rows"], lay["marked"]
    grid = WorldGrid.from_rows(rows)
    enc = Encoding.encode([(1, 1)], grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    a = layout_assumptions(lay, lambda x, y: enc.platform_var(x, y, (1, 1)))
    # (1) the printed layout as assumptions: SAT, and the model IS that layout
    s = Mi355Sat(workers=64)
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve_batch([a])[0] == SolverResult.Sat
    lay_out = check_sat_answer(cnf, s.solution_of(0, cnf.n_vars), enc, grid, k)
    assert sorted((x, y) for x, y, *_ in lay_out.platforms()) == sorted(tuple(p) for p in lay["supports_xy"])
    s.close()
    # (2) plain solve() at the README's bound: SAT (README.md:46-116), model self-certifying
    s = Mi355Sat()
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve() == SolverResult.Sat
    check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, k)
    s.close()
    # (3) one support fewer than printed contradicts the printed layout
    cnf1 = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k - 1}))
    s = Mi355Sat(workers=64)
    s.add_cnf(cnf1.lits, cnf1.offsets)
    assert s.solve_batch([a])[0] == SolverResult.Unsat
    s.close()
