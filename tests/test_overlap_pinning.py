"""Reference-held vectors for the DEFAULT platform set: the 40 placement pairs of the reference's own
`platform_overlap_yes` / `platform_overlap_no` unit tests (/root/reference/src/platform.rs:151-232 ->
tests/golden/overlap_pairs.json, generator committed).  They pin

  * the overlap geometry of both validators (product `PlatformLayout.validate` and oracle/layout_oracle.py,
    restating src/encoder/platform_layout.rs:85-149): two platforms at distinct anchors are flagged as overlapping
    exactly when the reference says `overlaps`;
  * the encoder's two overlap clause families (src/encoder.rs:559-571 corner-in-other, :576-596 row x column
    crossing) together with the DAG implications they rely on (:450-458): on a 16x16 all-terrain grid with the
    default platform set and no limit, assuming P_a(anchor a) AND P_b(anchor b)
      - is UNSAT - already by unit propagation - for every overlapping pair at distinct anchors,
      - is SAT with a validating layout that holds both platforms for every non-overlapping pair,
      - is SAT for the pairs at the SAME anchor (P_d(p) reads "a platform of at least d at p": nested sizes are one
        platform, src/encoder.rs:450-458 + platform_layout.rs:34-42), the layout holding the larger of the two;
    on the product CNF (C++ encoder), on the oracle's literal restatement of encoder.rs, through the product's
    kernels in the emulator build, and (-m gpu) on the MI355X through the C ABI.
"""
import numpy as np
import pytest

from helpers import check_sat_answer, emu_lib, golden
from oracle import encoder_oracle as eo, layout_oracle as lo, oracle as ora
from timberborn_support_solver_amd import (PLATFORMS_DEFAULT, Encoding, Mi355Sat, PlatformLayout, PlatformLimits,
                                           SolverResult, WorldGrid)

PAIRS = golden("overlap_pairs.json")["pairs"]
N = 16


def same_anchor(p):
    return p["a"][2:] == p["b"][2:]


def expected(p):
    return 20 if (p["overlap"] and not same_anchor(p)) else 10


def pid(p):
    a, b = p["a"], p["b"]
    return f"{a[0]}x{a[1]}@{a[2]},{a[3]}-{b[0]}x{b[1]}@{b[2]},{b[3]}-{'yes' if p['overlap'] else 'no'}"


def test_fixture_is_the_reference_test_matrix():
    assert len(PAIRS) == 40 and sum(p["overlap"] for p in PAIRS) == 22
    # the reference's predicate (platform.rs:85-98: inclusive corners) restated on the vectors themselves
    for p in PAIRS:
        (aw, ah, ax, ay), (bw, bh, bx, by) = p["a"], p["b"]
        geo = bx + bw - 1 >= ax and by + bh - 1 >= ay and bx <= ax + aw - 1 and by <= ay + ah - 1
        assert geo == p["overlap"], p


@pytest.mark.parametrize("p", [p for p in PAIRS if not same_anchor(p)], ids=pid)
def test_validators_flag_exactly_the_overlapping_pairs(p):
    grid = WorldGrid.rect(N, N)
    plats = [(p["a"][2], p["a"][3], p["a"][0], p["a"][1], 0), (p["b"][2], p["b"][3], p["b"][0], p["b"][1], 0)]
    res = PlatformLayout.from_platforms(plats).validate(grid)
    assert (res.n_overlapping_platforms > 0) == p["overlap"] and res.n_out_of_bounds_platforms == 0
    q = {(x, y): ((w, h), False) for x, y, w, h, _ in plats}
    _, overlapping, oob = lo.validate(q, eo.grid_rect(N, N))
    assert (len(overlapping) > 0) == p["overlap"] and not oob
    if p["overlap"]:
        assert res.n_overlapping_platforms == 2 and len(overlapping) == 2


@pytest.fixture(scope="module")
def formulas():
    grid = WorldGrid.rect(N, N)
    enc = Encoding.encode(PLATFORMS_DEFAULT, grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({}))
    o = eo.Encoding(list(PLATFORMS_DEFAULT), eo.grid_rect(N, N))
    cl, nv, cards = o.with_limits({})
    ocl, onv, _ = eo.into_cnf(cl, nv, cards)
    return grid, enc, cnf, o, ora.to_csr(ocl), onv


def assumptions(p, var_of):
    return [var_of(p["a"][2], p["a"][3], (p["a"][0], p["a"][1])), var_of(p["b"][2], p["b"][3], (p["b"][0], p["b"][1]))]


def check_layout_holds_the_pair(lay, p):
    have = {(x, y): (w, h) for x, y, w, h, _ in lay.platforms()}
    if same_anchor(p):
        big = max((p["a"][0], p["a"][1]), (p["b"][0], p["b"][1]))
        got = have[tuple(p["a"][2:])]
        assert got[0] >= big[0] and got[1] >= big[1]
    else:
        for w, h, x, y in (p["a"], p["b"]):
            assert have[(x, y)][0] >= w and have[(x, y)][1] >= h


@pytest.mark.parametrize("p", PAIRS, ids=pid)
def test_overlap_families_on_product_and_oracle_cnf(formulas, p):
    grid, enc, cnf, o, (olits, ooffs), onv = formulas
    want = expected(p)
    a = assumptions(p, enc.platform_var)
    s = ora.OracleSolver()
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve(a) == want
    if want == 10:
        check_layout_holds_the_pair(check_sat_answer(cnf, s.model(cnf.n_vars), enc, grid, N * N), p)
    # unit propagation alone decides the overlapping pairs (binary overlap clauses + DAG implications)
    c, _, _, _ = ora.bcp(cnf.lits, cnf.offsets, cnf.n_vars, a)
    assert bool(c) == (want == 20)
    # the oracle's literal restatement of encoder.rs
    s2 = ora.OracleSolver()
    s2.add_cnf(olits, ooffs)
    s2.reserve(onv)
    assert s2.solve(assumptions(p, lambda x, y, d: o.plat_var[(x, y, d)])) == want


def test_overlap_families_through_the_emulated_kernels(formulas):
    grid, enc, cnf, *_ = formulas
    sets = [assumptions(p, enc.platform_var) for p in PAIRS]
    e = Mi355Sat(_lib_override=emu_lib(), workers=len(sets), simp=-1)
    e.add_cnf(cnf.lits, cnf.offsets)
    confl, _, _ = e.propagate_batch(sets, n_vars=cnf.n_vars)
    assert [bool(c) for c in confl] == [expected(p) == 20 for p in PAIRS]
    e.close()


@pytest.mark.gpu
def test_gpu_overlap_families(formulas):
    grid, enc, cnf, *_ = formulas
    sets = [assumptions(p, enc.platform_var) for p in PAIRS]
    s = Mi355Sat(workers=4 * len(sets))
    s.add_cnf(cnf.lits, cnf.offsets)
    res = s.solve_batch(sets)
    assert [r.value for r in res] == [expected(p) for p in PAIRS]
    for i, p in enumerate(PAIRS):
        if res[i] == SolverResult.Sat:
            check_layout_holds_the_pair(check_sat_answer(cnf, s.solution_of(i, cnf.n_vars), enc, grid, N * N), p)
    s.close()
    s = Mi355Sat()
    s.add_cnf(cnf.lits, cnf.offsets)
    confl, vals, tl = s.propagate_batch(sets, n_vars=cnf.n_vars)
    for i, p in enumerate(PAIRS):
        c, v, n, _ = ora.bcp(cnf.lits, cnf.offsets, cnf.n_vars, sets[i])
        assert bool(confl[i]) == bool(c) == (expected(p) == 20)
        if not c:
            assert n == tl[i] and np.array_equal(v, vals[i])
    s.close()
