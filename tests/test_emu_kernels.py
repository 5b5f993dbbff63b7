"""Kernel LOGIC tests without a GPU: the same kernels.hip.h + mi355sat.hip compiled with g++
against tests/emu (a 64-lane fiber emulator; see tests/emu/hip_shim.h) and driven through the
same C ABI.  Parity with the oracle on small cases; the GPU tests are the authority for the
real machine."""
import threading
import time

import numpy as np
import pytest

from helpers import (assert_ring_records_are_implied, VERDICTS, check_sat_answer, emu_lib, long_list_formula, make_grid, platform_defs,
                     scripted_decisions)
from oracle import oracle as ora
from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits, SolverResult, solver_loop

SMALL = [v for v in VERDICTS["verdicts"] if v["n_clauses"] < 1300 or (v["terrain"], v["platforms"], v["k"]) in
         {("ex1", "default", 1), ("ex1", "default", 2), ("rect8x8", "default", 2)}]


def emu_solver(**kw):
    kw.setdefault("simp", -1)      # probing through the fiber emulator is slow: the simplification has its own tests below
    return Mi355Sat(_lib_override=emu_lib(), **kw)


@pytest.mark.parametrize("v", SMALL, ids=lambda v: f"{v['terrain']}-{v['platforms']}-k{v['k']}")
def test_emulated_solve_matches_golden_verdicts(v):
    grid = make_grid(v["terrain"])
    enc = Encoding.encode(platform_defs(v["platforms"]), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): v["k"]}))
    s = emu_solver(workers=2, slice_conflicts=500, reduce_first=40, reduce_inc=10)
    s.add_cnf(cnf.lits, cnf.offsets)
    r = s.solve()
    assert r.name.upper() == v["verdict"]
    if r == SolverResult.Sat:
        check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, v["k"])
    st = s.stats()
    assert st["propagations"] == st["n_deq"] and st["n_sat"] + st["n_unsat"] == 1
    s.close()


def test_emulated_bcp_fixpoints_are_bit_exact():
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("default"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 6}))
    scripts = [scripted_decisions(enc, grid, seed, 6) for seed in range(1, 9)] + [[], [enc.platform_var(2, 2, (5, 5))]]
    s = emu_solver()
    s.add_cnf(cnf.lits, cnf.offsets)
    confl, vals, tl = s.propagate_batch(scripts, n_vars=cnf.n_vars)
    n_fix = 0
    for i, dec in enumerate(scripts):
        c, v, n, _ = ora.bcp(cnf.lits, cnf.offsets, cnf.n_vars, dec)
        assert c == confl[i]
        if not c:
            n_fix += 1
            assert np.array_equal(v, vals[i]) and n == tl[i]
    assert n_fix >= 3
    s.close()


def test_emulated_sweep_under_assumptions_and_reduce_db():
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 8}), sweep=True)
    ks = [3, 4, 8]
    s = emu_solver(workers=3, slice_conflicts=100, reduce_first=25, reduce_inc=5)
    s.add_cnf(cnf.lits, cnf.offsets)
    res = s.solve_batch([[-int(cnf.card_outputs[k])] if k < 8 else [] for k in ks])
    assert [r.name for r in res] == ["Unsat", "Sat", "Sat"]
    for i, k in enumerate(ks):
        if res[i] == SolverResult.Sat:
            check_sat_answer(cnf, s.solution_of(i, cnf.n_vars), enc, grid, k)
    assert s.stats()["reduce_dbs"] > 0
    s.close()


def test_emulated_solver_loop_ex1_reaches_unsat_like_the_reference_loop():
    grid = make_grid("ex1")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    lines = []
    hist = solver_loop(grid, enc, PlatformLimits({(1, 1): 8}), make_solver=lambda: emu_solver(workers=1), out=lines.append)
    assert [h["result"].name for h in hist][-1] == "Unsat" and hist[-1]["k"] == 2          # k* = 3
    assert all(h["valid"] for h in hist[:-1])
    assert lines[-1] == "No solution found for the current constraints"
    assert lines[0].startswith("Solution found (")


def test_degenerate_inputs():
    s = emu_solver()
    assert s.solve() == SolverResult.Sat                       # empty formula
    s.close()
    s = emu_solver()
    s.add_clause([1, 2]); s.add_clause([-1]); s.add_clause([-2])
    assert s.solve() == SolverResult.Unsat                     # refuted by host-side unit propagation
    s.close()
    s = emu_solver()
    s.add_clause([])                                           # empty clause
    assert s.solve() == SolverResult.Unsat
    s.close()
    s = emu_solver()
    s.add_clause([1, -1, 2]); s.add_clause([3, 3, -4]); s.reserve(9)
    assert s.solve() == SolverResult.Sat
    m = s.full_solution(9)
    assert len(m) == 9 and (m[2] == 1 or m[3] == -1) and s.lit_val(3) in (3, -3) and s.lit_val(12) == 0
    s.close()


def test_interrupt_from_another_thread_and_conflict_budget():
    grid = make_grid("rect16x16")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 14}))   # hard UNSAT: never finishes in the emulator
    s = emu_solver(workers=1, slice_conflicts=5)
    s.add_cnf(cnf.lits, cnf.offsets)
    intr = s.interrupter()
    threading.Timer(1.0, intr.interrupt).start()
    t0 = time.time()
    assert s.solve() == SolverResult.Interrupted
    assert time.time() - t0 < 60 and s.stats()["n_terminated"] == 1
    # the interrupt was consumed by the solve it stopped: the same handle searches again ...
    threading.Timer(1.0, intr.interrupt).start()
    t0 = time.time()
    assert s.solve() == SolverResult.Interrupted and time.time() - t0 > 0.5
    # ... and one that arrives while nothing runs is not lost: it stops the next solve at once, only that one
    intr.interrupt()
    t0 = time.time()
    assert s.solve() == SolverResult.Interrupted and time.time() - t0 < 0.5
    assert s.stats()["n_terminated"] == 3
    s.close()
    intr.interrupt()   # after close(): a no-op, not a call into a freed handle
    s = emu_solver(workers=1, slice_conflicts=5, conflict_budget=10)
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve() == SolverResult.Interrupted
    s.close()


def test_emulated_cube_splitting_partitions_the_search_space():
    """Work stealing between short slices: an UNSAT verdict needs every split-off cube closed
    (closed = splits + 1), a SAT model found inside a cube is a model of the whole formula."""
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    for k, want in [(3, SolverResult.Unsat), (4, SolverResult.Sat)]:
        cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
        s = emu_solver(workers=8, slice_conflicts=8, cube_split=1)
        s.add_cnf(cnf.lits, cnf.offsets)
        assert s.solve() == want
        if want == SolverResult.Sat:
            check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, k)
        s.close()
    # portfolio mode (no splitting) still answers
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 3}))
    s = emu_solver(workers=3, slice_conflicts=40)
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve() == SolverResult.Unsat
    s.close()


def test_emulated_unsat_verdicts_carry_a_drup_proof(tmp_path):
    """Every worker logs the clauses it learns (the default configuration: several workers, clause exchange on); the
    host interleaves the logs slice by slice; the oracle's independent RUP checker must accept the proof."""
    from timberborn_support_solver_amd.dimacs import read_drup, read_dimacs, write_dimacs
    exchanged = 0
    for terrain, pset, k in [("ex1", "1x1", 2)]:
        grid = make_grid(terrain)
        enc = Encoding.encode(platform_defs(pset), grid)
        cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
        proof = str(tmp_path / "p.drup")
        s = emu_solver(workers=8, slice_conflicts=8)
        s.set_proof_path(proof)
        s.add_cnf(cnf.lits, cnf.offsets)
        assert s.solve() == SolverResult.Unsat
        st = s.stats()
        exchanged += st["shared_imported"] + st["shared_imported_units"]
        s.close()
        assert ora.check_rup(cnf.lits, cnf.offsets, cnf.n_vars, read_drup(proof)) == 1
        # DIMACS round trip of the same CNF
        path = str(tmp_path / "f.cnf")
        write_dimacs(path, cnf.lits, cnf.offsets, cnf.n_vars)
        l2, o2, nv2 = read_dimacs(path)
        assert nv2 == cnf.n_vars and np.array_equal(l2, cnf.lits) and np.array_equal(o2, cnf.offsets)
    # deletion lines: a worker that reduces its clause database logs "d ..." for the dropped clauses nobody else can hold
    # (never exchanged, never imported); the checker deletes them and must still accept the proof
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 3}))
    proof = str(tmp_path / "d.drup")
    s = emu_solver(workers=3, slice_conflicts=16, reduce_first=25, reduce_inc=10)
    s.set_proof_path(proof)
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve() == SolverResult.Unsat
    st = s.stats()
    assert st["reduce_dbs"] > 0
    exchanged += st["shared_imported"] + st["shared_imported_units"]
    s.close()
    assert exchanged > 0      # the proofs include derivations that used other workers' clauses
    n_del = sum(1 for line in open(proof) if line.startswith("d "))
    assert n_del > 0
    assert ora.check_rup(cnf.lits, cnf.offsets, cnf.n_vars, read_drup(proof)) == 1
    assert ora.check_rup(cnf.lits, cnf.offsets, cnf.n_vars, read_drup(proof, deletions=False)) == 1


def test_emulated_weight_loop_like_the_gui():
    """crates/gui/src/app.rs:235-245: tighten the total weight until Unsat; here with weights = platform area
    on ex1 the loop must end at the minimum total area of a valid layout, every step validated."""
    from timberborn_support_solver_amd import weight_loop
    grid = make_grid("ex1")
    defs = [(1, 1), (1, 2), (3, 3)]
    enc = Encoding.encode(defs, grid)
    weights = {(1, 1): 1, (1, 2): 1, (3, 3): 7}     # nested types add up: 1x2 = 1+1, 3x3 = 1+1+7 (platform_layout.rs:174-183)
    hist = weight_loop(grid, enc, PlatformLimits({}, weights, None), make_solver=lambda: emu_solver(workers=1), out=lambda s: None)
    assert hist[-1]["result"] == SolverResult.Unsat
    ws = [h["weight"] for h in hist[:-1]]
    assert all(h["valid"] for h in hist[:-1]) and ws == sorted(ws, reverse=True) and len(set(ws)) == len(ws)
    assert hist[-1]["weight_limit"] == ws[-1] - 1
    # the oracle agrees that one less is impossible
    cnf = enc.with_limits_into_cnf(PlatformLimits({}, weights, ws[-1] - 1))
    o = ora.OracleSolver()
    o.add_cnf(cnf.lits, cnf.offsets)
    assert o.solve() == 20


def test_emulated_clause_exchange_and_locality_order():
    """Workers pass short / low-LBD learnt clauses on between slices (units are attached at level 0);
    verdicts and models stay those of the formula, also with the device's own variable numbering."""
    grid = make_grid("ex1")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 2}))                  # k* = 3: UNSAT
    seen = 0
    for kw in (dict(), dict(var_order=1), dict(share=-1)):
        s = emu_solver(workers=8, slice_conflicts=8, **kw)
        s.add_cnf(cnf.lits, cnf.offsets)
        assert s.solve() == SolverResult.Unsat
        st = s.stats()
        if kw.get("share") == -1:
            assert st["shared_exported"] == 0 and st["shared_imported"] == 0
        else:
            assert st["shared_exported"] > 0
            seen += st["shared_imported"] + st["shared_imported_units"]
        s.close()
    assert seen > 0
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 4}))
    s = emu_solver(workers=8, slice_conflicts=8, var_order=1)
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve() == SolverResult.Sat
    check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, 4)
    s.close()


def test_emulated_exchange_ring_holds_only_consequences_of_the_formula():
    """A sweep over several bounds with the exchange on: what workers of one instance put into the ring is
    attached by workers of every other instance, so each record must follow from the formula alone - in
    particular it must not depend on any instance's assumption (the at-most-k bound)."""
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 8}), sweep=True)
    ks = [8, 5, 4, 3, 2]
    s = emu_solver(workers=10, slice_conflicts=16, share_lbd=6)
    s.add_cnf(cnf.lits, cnf.offsets)
    res = s.solve_batch([[-int(cnf.card_outputs[k])] if k < 8 else [] for k in ks])
    assert [r.name for r in res] == ["Sat", "Sat", "Sat", "Unsat", "Unsat"]
    assert assert_ring_records_are_implied(s, cnf) > 0
    s.close()


def test_emulated_locality_order_keeps_bcp_fixpoints_bit_exact():
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("default"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 6}))
    scripts = [scripted_decisions(enc, grid, seed, 6) for seed in range(1, 5)] + [[]]
    s = emu_solver(var_order=1)
    s.add_cnf(cnf.lits, cnf.offsets)
    confl, vals, tl = s.propagate_batch(scripts, n_vars=cnf.n_vars)
    for i, dec in enumerate(scripts):
        c, v, n, _ = ora.bcp(cnf.lits, cnf.offsets, cnf.n_vars, dec)
        assert c == confl[i]
        if not c:
            assert np.array_equal(v, vals[i]) and n == tl[i]
    s.close()


def test_emulated_sweep_moves_workers_to_open_instances_and_drops_implied_ones():
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 8}), sweep=True)
    ks = [8, 6, 5, 4, 3, 2]                                                    # k* = 4
    sets = [[-int(cnf.card_outputs[k])] if k < 8 else [] for k in ks]
    s = emu_solver(workers=6, slice_conflicts=20)
    s.add_cnf(cnf.lits, cnf.offsets)
    s.sweep_begin(sets)
    for _ in range(400):
        res, nd = s.sweep_step()
        sat_k = min([k for k, r in zip(ks, res) if r == SolverResult.Sat], default=None)
        unsat_k = max([k for k, r in zip(ks, res) if r == SolverResult.Unsat], default=None)
        if sat_k is not None and unsat_k is not None and unsat_k + 1 >= sat_k:
            break
        s.sweep_drop([i for i, k in enumerate(ks) if res[i] == SolverResult.Interrupted and
                      ((sat_k is not None and k > sat_k) or (unsat_k is not None and k < unsat_k))])
    assert (sat_k, unsat_k) == (4, 3)
    s.sweep_end()
    i4 = ks.index(4)
    check_sat_answer(cnf, s.solution_of(i4, cnf.n_vars), enc, grid, 4)
    for i, k in enumerate(ks):                          # whatever else was decided is consistent with k* = 4
        assert res[i] in (SolverResult.Interrupted, SolverResult.Sat if k >= 4 else SolverResult.Unsat)
    s.close()


def test_emulated_solver_loop_sweep_prints_the_reference_messages():
    from timberborn_support_solver_amd import solver_loop_sweep
    grid = make_grid("ex1")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    lines = []
    # specialize_after: None = the batch runs to the cut;  0 = it hands over to the sequential loop (a fresh CNF per
    # bound, as the reference poses them) as soon as at most two bounds are open
    for spec in (None, 0.0):
        lines = []
        hist = solver_loop_sweep(grid, enc, PlatformLimits({(1, 1): 8}), out=lines.append, specialize_after=spec,
                                 make_solver=lambda: emu_solver(workers=9, slice_conflicts=20))
        sat = [h for h in hist if h["result"] == SolverResult.Sat]
        assert hist[-1]["result"] == SolverResult.Unsat and hist[-1]["k"] == 2 and len(sat) >= 1           # k* = 3
        assert sat[-1]["count"] == 3 and all(h["valid"] for h in sat)
        assert [h["count"] for h in sat] == sorted([h["count"] for h in sat], reverse=True)
        assert "Solution found (3 platforms total)" in lines and lines[-1] == "No solution found for the current constraints"
    with pytest.raises(ValueError):
        solver_loop_sweep(grid, enc, PlatformLimits({(1, 1): 3}, weights={(1, 1): 2}, weight_limit=5))


def test_emulated_assignment_in_hbm_variant(tmp_path):
    """lds_val=-1 forces the variant the large instances run (the small test instances would otherwise stage the
    2-bit assignment in LDS): the assignment stays in HBM behind relaxed agent-scope atomics.  Verdicts, models,
    a checked DRUP proof and bit-exact BCP fixpoints through that path."""
    from timberborn_support_solver_amd.dimacs import read_drup
    for terrain, plats, k in [("ex1", "1x1", 3), ("rect8x8", "1x1", 3), ("rect8x8", "default", 2)]:
        v = [x for x in VERDICTS["verdicts"] if (x["terrain"], x["platforms"], x["k"]) == (terrain, plats, k)][0]
        grid = make_grid(terrain)
        enc = Encoding.encode(platform_defs(plats), grid)
        cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
        s = emu_solver(workers=4, slice_conflicts=50, lds_val=-1)
        s.add_cnf(cnf.lits, cnf.offsets)
        r = s.solve()
        assert r.name.upper() == v["verdict"]
        if r == SolverResult.Sat:
            check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, k)
        s.close()
    grid = make_grid("ex1")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 2}))
    proof = str(tmp_path / "p.drup")
    s = emu_solver(workers=1, slice_conflicts=50, lds_val=-1)
    s.set_proof_path(proof)
    s.add_cnf(cnf.lits, cnf.offsets)
    assert s.solve() == SolverResult.Unsat
    s.close()
    assert ora.check_rup(cnf.lits, cnf.offsets, cnf.n_vars, read_drup(proof)) == 1
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("default"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 6}))
    scripts = [scripted_decisions(enc, grid, seed, 6) for seed in range(1, 6)] + [[]]
    s = emu_solver(lds_val=-1)
    s.add_cnf(cnf.lits, cnf.offsets)
    confl, vals, tl = s.propagate_batch(scripts, n_vars=cnf.n_vars)
    for i, dec in enumerate(scripts):
        c, v, n, _ = ora.bcp(cnf.lits, cnf.offsets, cnf.n_vars, dec)
        assert c == confl[i]
        if not c:
            assert np.array_equal(v, vals[i]) and n == tl[i]
    s.close()


@pytest.mark.parametrize("terrain,pset,k,want", [("ex1", "1x1", 2, "Unsat"), ("ex1", "1x1", 3, "Sat"), ("ex1", "default", 1, "Sat")])
def test_emulated_simplification_keeps_verdicts_models_and_proofs(tmp_path, terrain, pset, k, want):
    """SURVEY 8 f3 (`simp::Glucose`, crates/repl/src/main.rs:17): equivalent-literal substitution, failed-literal
    probing (ms_probe_kernel) and subsumption (ms_subsume_kernel) before the search.  Verdicts are those without
    it; models come back in the caller's variables and satisfy the ORIGINAL clauses; the DRUP proof (simplification
    lemmas first, then the search's) passes the oracle's RUP checker against the original formula."""
    from timberborn_support_solver_amd.dimacs import read_drup
    grid = make_grid(terrain)
    enc = Encoding.encode(platform_defs(pset), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    proof = str(tmp_path / "p.drup")
    s = emu_solver(workers=2, simp=0)
    if want == "Unsat":
        s.set_proof_path(proof)
    s.add_cnf(cnf.lits, cnf.offsets)
    r = s.solve()
    assert r.name == want
    st = s.stats()
    assert st["simp_units"] > 0                               # probing found failed literals on every one of these
    if r == SolverResult.Sat:
        check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, k)
    else:
        assert ora.check_rup(cnf.lits, cnf.offsets, cnf.n_vars, read_drup(proof)) == 1
    s.close()


def test_emulated_simplification_substitutes_equivalent_literals_and_maps_models_back():
    """A chain of equivalences (x1 = x2 = ~x3 as binary implication cycles) plus clauses over them: the substituted
    variables must come back with their representative's value, assumptions on substituted variables must work,
    and a formula that forces x and ~x into one component is UNSAT without search."""
    cl = [[-1, 2], [-2, 1], [-2, -3], [3, 2], [1, 3, 4, 5], [-4, -5, 6, 7], [-1, 4, 6, 7], [-6, -7, 8, 9]]
    s = emu_solver(workers=1, simp=0)
    for c in cl:
        s.add_clause(c)
    assert s.solve() == SolverResult.Sat
    m = s.full_solution(9)
    assert ora.check_model(*ora.to_csr(cl), m) == -1 and m[0] == m[1] == -m[2]
    assert s.stats()["simp_equivalences"] >= 2
    s.close()
    s = emu_solver(workers=2, simp=0)
    for c in cl:
        s.add_clause(c)
    res = s.solve_batch([[3], [-2, 3], [2, -4, -5]])            # x3 forces ~x1, ~x2;  ~x2 & x3 consistent;  x2 -> x1 -> needs 4..7
    assert [r.name for r in res] == ["Sat", "Sat", "Sat"]
    for i, a in enumerate([[3], [-2, 3], [2, -4, -5]]):
        m = s.solution_of(i, 9)
        assert ora.check_model(*ora.to_csr(cl + [[l] for l in a]), m) == -1
    s.close()
    s = emu_solver(workers=1, simp=0)
    for c in cl + [[1, 3, 9], [-9, 1], [-1, -3, 9], [2, -9, 3], [-2, 3]]:   # x2 -> x3 closes the cycle x2 -> x3 -> ~x2
        s.add_clause(c)
    o = ora.OracleSolver()
    lits, offs = ora.to_csr(cl + [[1, 3, 9], [-9, 1], [-1, -3, 9], [2, -9, 3], [-2, 3]])
    o.add_cnf(lits, offs)
    assert s.solve().value == o.solve()
    s.close()


def test_emulated_subsumption_kernel_removes_and_strengthens():
    """(a | b) subsumes (a | b | c); (a | ~c) strengthens (a | b | c | d) ... to (a | b | d) only via (a|~c)+(a|b|c|d):
    checked through the counters and by equivalence of the answers under every assumption of the inputs."""
    import itertools
    cl = [[1, 2], [1, 2, 3], [1, 2, 3, 4], [1, -3], [-1, 5, 6], [-1, 5, 6, 7], [2, 3, 4, 5], [-2, 3, 4, 5], [5, 6, 7, 8]]
    s = emu_solver(workers=2, simp=0)
    for c in cl:
        s.add_clause(c)
    sets = [[v if b else -v for v, b in zip(range(1, 5), bits)] for bits in itertools.product([0, 1], repeat=4)]
    res = s.solve_batch(sets)
    assert s.stats()["simp_clauses_removed"] >= 3
    for a, r in zip(sets, res):
        o = ora.OracleSolver()
        lits, offs = ora.to_csr(cl)
        o.add_cnf(lits, offs)
        assert r.value == o.solve(a), a
    s.close()


def test_emulated_variable_elimination_rebuilds_models_and_keeps_assumptions():
    """Bounded variable elimination (`SimpSolver::eliminate` of the reference's backend, opts.simp = 2) on small
    hand-made formulas: x3 and x6 occur only in a few clauses and are resolved away; the model that comes back
    gives them values satisfying the ORIGINAL clauses; a variable named in an assumption is not eliminated, so every
    assumption set gets the oracle's verdict."""
    import itertools
    cl = [[1, 3], [-3, 2, 4], [-3, 5], [3, -4, -5], [6, 1, 2], [-6, -1], [-6, -2, 5], [4, 5, 7, 8], [-7, -8, 1], [7, -5, 2]]
    s = emu_solver(workers=1, simp=2)
    for c in cl:
        s.add_clause(c)
    assert s.solve() == SolverResult.Sat
    assert s.stats()["simp_eliminated"] >= 2
    assert ora.check_model(*ora.to_csr(cl), s.full_solution(8)) == -1
    s.close()
    sets = [[v if b else -v for v, b in zip((3, 6, 1), bits)] for bits in itertools.product([0, 1], repeat=3)]
    s = emu_solver(workers=2, simp=2)
    for c in cl:
        s.add_clause(c)
    res = s.solve_batch(sets)
    for i, (a, r) in enumerate(zip(sets, res)):
        o = ora.OracleSolver()
        o.add_cnf(*ora.to_csr(cl))
        assert r.value == o.solve(a), a
        if r == SolverResult.Sat:
            assert ora.check_model(*ora.to_csr(cl + [[l] for l in a]), s.solution_of(i, 8)) == -1
    s.close()


@pytest.mark.parametrize("terrain,pset,k,want", [("ex1", "1x1", 3, "Sat"), ("ex1", "default", 1, "Sat"), ("ex3", "1x1", 3, "Unsat")])
def test_emulated_variable_elimination_on_the_encoder_s_formulas(tmp_path, terrain, pset, k, want):
    """The same on the encoder's CNFs: verdicts as without it, layouts valid against the original clauses and the
    terrain, and the DRUP proof - the resolvents are lemmas of it - accepted by the oracle's RUP checker."""
    from timberborn_support_solver_amd.dimacs import read_drup
    grid = make_grid(terrain)
    enc = Encoding.encode(platform_defs(pset), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    proof = str(tmp_path / "p.drup")
    s = emu_solver(workers=2, simp=2)
    if want == "Unsat":
        s.set_proof_path(proof)
    s.add_cnf(cnf.lits, cnf.offsets)
    r = s.solve()
    assert r.name == want
    assert s.stats()["simp_eliminated"] > 0
    if r == SolverResult.Sat:
        check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, k)
    else:
        assert ora.check_rup(cnf.lits, cnf.offsets, cnf.n_vars, read_drup(proof)) == 1
    s.close()


def test_emulated_variable_elimination_keeps_a_sweep_s_bounds():
    """The bounds of a sweep are assumptions on the totalizer's outputs: those variables survive the elimination and
    every bound has its golden verdict, with models valid in the caller's variables."""
    grid = make_grid("ex1")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 6}), sweep=True)
    ks = [6, 4, 3, 2, 1]                                                       # k* = 3
    sets = [[-int(cnf.card_outputs[k])] if k < 6 else [] for k in ks]
    s = emu_solver(workers=5, simp=2)
    s.add_cnf(cnf.lits, cnf.offsets)
    res = s.solve_batch(sets)
    assert s.stats()["simp_eliminated"] > 0
    for i, (k, r) in enumerate(zip(ks, res)):
        assert r.name == ("Sat" if k >= 3 else "Unsat"), k
        if r == SolverResult.Sat:
            check_sat_answer(cnf, s.solution_of(i, cnf.n_vars), enc, grid, k)
    s.close()


def test_emulated_sweep_priorities_and_reopening_keep_the_answers():
    """mi355sat_sweep_set_weights / mi355sat_sweep_reopen are scheduling only: whatever the caller's priorities and
    however often bounds are withdrawn and taken up again, every decided bound has its golden verdict."""
    from timberborn_support_solver_amd.solver import SolverError
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 8}), sweep=True)
    ks = [8, 6, 5, 4, 3, 2]                                                    # k* = 4
    sets = [[-int(cnf.card_outputs[k])] if k < 8 else [] for k in ks]
    s = emu_solver(workers=12, slice_conflicts=15)
    s.add_cnf(cnf.lits, cnf.offsets)
    s.sweep_begin(sets)
    with pytest.raises(SolverError):
        s.sweep_set_weights([1.0, 2.0])                                        # one weight per instance
    with pytest.raises(SolverError):
        s.sweep_set_weights([1.0] * 5 + [-1.0])
    s.sweep_drop([0, 1, 2])                                                    # start with the low bounds only
    s.sweep_set_weights([0.02, 0.02, 0.02, 1.0, 0.02, 1.0])
    res = None
    for step in range(600):
        res, nd = s.sweep_step()
        if step == 3:
            s.sweep_reopen([0, 1, 2, 5])                                       # (5 was never withdrawn: ignored)
            s.sweep_set_weights([1.0, 0.02, 0.02, 0.02, 0.02, 1.0])
        if all(r != SolverResult.Interrupted for r in res):
            break
    s.sweep_end()
    assert [r.name for r in res] == ["Sat", "Sat", "Sat", "Sat", "Unsat", "Unsat"]
    for i, k in enumerate(ks):
        if res[i] == SolverResult.Sat:
            check_sat_answer(cnf, s.solution_of(i, cnf.n_vars), enc, grid, k)
    s.close()


def test_emulated_reopened_bound_without_weights_gets_a_worker():
    """A bound withdrawn while every worker went to the other open bound, then taken up again WITHOUT weights: the
    rebalancing must hand it a worker from the bound that has the most (round 2 left it with none and the sweep never
    ended)."""
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 8}), sweep=True)
    ks = [3, 2]                                                                # both refuted (k* = 4)
    s = emu_solver(workers=4, slice_conflicts=10)
    s.add_cnf(cnf.lits, cnf.offsets)
    s.sweep_begin([[-int(cnf.card_outputs[k])] for k in ks])
    s.sweep_drop([1])
    s.sweep_step()                                                             # the workers of bound 2 move to bound 3
    s.sweep_reopen([1])
    res = None
    for _ in range(400):
        res, _ = s.sweep_step()
        if all(r != SolverResult.Interrupted for r in res):
            break
    s.sweep_end()
    assert [r.name for r in res] == ["Unsat", "Unsat"]
    s.close()


@pytest.mark.parametrize("one_per_simd", [2, 4], ids=["two-waves-build", "full-fleet-build"])
def test_emulated_called_builds_of_the_search_kernel(one_per_simd):
    """The search kernel exists in three builds (waves per SIMD 1 / 2 / 4).  Small fleets run the inlined one; this forces
    the two CALLED builds - the per-conflict code a function that works on register copies (2) or on the caller's context
    (4) - through the same verdict / model / exchange checks."""
    for terrain, pset, k, want in [("ex1", "1x1", 2, "Unsat"), ("ex1", "1x1", 3, "Sat"), ("rect8x8", "1x1", 3, "Unsat")]:
        grid = make_grid(terrain)
        enc = Encoding.encode(platform_defs(pset), grid)
        cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
        s = emu_solver(workers=4, slice_conflicts=10, one_per_simd=one_per_simd)
        s.add_cnf(cnf.lits, cnf.offsets)
        assert s.solve().name == want
        if want == "Sat":
            check_sat_answer(cnf, s.full_solution(cnf.n_vars), enc, grid, k)
        s.close()


def test_emulated_deterministic_mode_repeats_itself():
    """opts.deterministic: conflict-bounded slices, ordered collection of the exchanged clauses, no ramp-up, nobody leaves a
    slice early - two runs with the same seed report the same counters (and the golden verdict)."""
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 3}))
    runs = []
    for _ in range(2):
        s = Mi355Sat(_lib_override=emu_lib(), workers=4, slice_conflicts=25, deterministic=1, seed=7, simp=-1)
        s.add_cnf(cnf.lits, cnf.offsets)
        assert s.solve() == SolverResult.Unsat
        st = s.stats()
        runs.append((st["conflicts"], st["propagations"], st["decisions"], st["shared_exported"], st["shared_imported"]))
        s.close()
    assert runs[0] == runs[1] and runs[0][0] > 0 and runs[0][3] > 0


@pytest.mark.parametrize("terrain,pset,k0,kstar", [("ex1", "1x1", 6, 3), ("rect8x8", "1x1", 8, 4)])
def test_emulated_lookahead_loop_is_the_sequential_loop(terrain, pset, k0, kstar):
    """solver_loop_pair: the bound the reference would pose next and the one below it run side by side (two handles, two
    host threads); what comes out is the reference's loop - its messages, valid layouts, the golden optimum and the
    refuted bound."""
    from timberborn_support_solver_amd import solver_loop_pair
    grid = make_grid(terrain)
    enc = Encoding.encode(platform_defs(pset), grid)
    lines = []
    hist = solver_loop_pair(grid, enc, PlatformLimits({(1, 1): k0}), out=lines.append,
                            make_solver=lambda: emu_solver(workers=3, slice_conflicts=20))
    assert hist[-1]["result"] == SolverResult.Unsat and hist[-1]["k"] == kstar - 1
    sat = [h for h in hist if h["result"] == SolverResult.Sat]
    assert sat and sat[-1]["count"] == kstar and all(h["valid"] and h["count"] <= h["k"] for h in sat)
    assert lines[-1] == "No solution found for the current constraints"
    assert f"Solution found ({kstar} platforms total)" in lines and "Solution validation FAILED" not in lines


@pytest.mark.parametrize("terrain,pset,k0,kstar,width", [("ex1", "1x1", 6, 3, 3), ("rect8x8", "1x1", 8, 4, 4), ("rect8x8", "default", 5, 2, 4)])
def test_emulated_fan_loop_is_the_sequential_loop(terrain, pset, k0, kstar, width):
    """solver_loop_fan: `width` bounds side by side (one handle, one host thread each); what comes out is the reference's
    loop - its messages, valid layouts with count <= the bound posed, the golden optimum and the refuted bound - and no
    solver is left running."""
    from timberborn_support_solver_amd import solver_loop_fan
    grid = make_grid(terrain)
    enc = Encoding.encode(platform_defs(pset), grid)
    lines, made = [], []

    def mk():
        made.append(emu_solver(workers=2, slice_conflicts=20))
        return made[-1]

    hist = solver_loop_fan(grid, enc, PlatformLimits({(1, 1): k0}), out=lines.append, make_solver=mk, width=width)
    assert hist[-1]["result"] == SolverResult.Unsat and hist[-1]["k"] == kstar - 1
    sat = [h for h in hist if h["result"] == SolverResult.Sat]
    assert sat and sat[-1]["count"] == kstar and all(h["valid"] and h["count"] <= h["k"] for h in sat)
    assert [h["k"] for h in hist] == sorted((h["k"] for h in hist), reverse=True)
    assert lines[-1] == "No solution found for the current constraints"
    assert f"Solution found ({kstar} platforms total)" in lines and "Solution validation FAILED" not in lines
    assert len(made) >= width and all(m._h is None or not m._h for m in made)      # every handle closed


@pytest.mark.parametrize("lds_val", [0, -1], ids=["assignment-in-lds", "assignment-in-slab"])
def test_emulated_bcp_fixpoints_long_clauses_and_long_watch_lists(lds_val):
    """propagate() against the oracle's occurrence-list BCP on formulas whose steps take every side path: several
    clause tails per step, watch lists of 44 long clauses (remainder spread flat over the wave), moved watches, two
    groups meeting in one clause.  Fixpoints are unique, so they must agree literal for literal (and so must "conflict")."""
    n_fix = n_conf = 0
    for seed in (11, 12, 13):
        lits, offsets, n_vars, n_hubs, rng = long_list_formula(seed)
        scripts = []
        for _ in range(10):
            dec = [int(h + 1) for h in rng.permutation(n_hubs)[: int(rng.integers(1, n_hubs + 1))]]
            dec += [int(v + 1) * (1 if rng.random() < 0.5 else -1) for v in rng.choice(np.arange(n_hubs, n_vars), size=int(rng.integers(4, 110)), replace=False)]
            scripts.append([int(x) for x in rng.permutation(dec)])
        s = emu_solver(lds_val=lds_val, simp=-1)
        s.add_cnf(lits, offsets)
        confl, vals, tl = s.propagate_batch(scripts, n_vars=n_vars)
        for i, dec in enumerate(scripts):
            c, v, n, _ = ora.bcp(lits, offsets, n_vars, dec)
            assert c == confl[i], (seed, i)
            if c:
                n_conf += 1
            else:
                n_fix += 1
                assert np.array_equal(v, vals[i]) and n == tl[i], (seed, i)
        st = s.stats()
        assert st["n_move"] > 0 and st["n_cl_lit"] > 8 * st["n_move"]       # watches moved, tails scanned
        s.close()
    assert n_fix >= 5 and n_conf >= 3


@pytest.mark.parametrize("lds_val", [0, -1], ids=["assignment-in-lds", "assignment-in-slab"])
def test_emulated_search_keeps_long_watch_lists_intact(lds_val):
    """The same side paths under SEARCH: lists are revisited after backjumps and rewritten in place by the flat remainder,
    learnt clauses of tens of literals join them.  Verdicts against the oracle, models against every clause.  (What this
    cannot see: ONE watcher of a clause lost - the other watch still finds the conflict, only later; tried by mutation.)
    A random 3-SAT core (120 variables, 3.0 clauses per variable on top of the chains) keeps the conflicts coming; the
    hubs (40 clauses of 4..6 literals each, tight enough to propagate) and the long clauses sit on its variables."""
    n_sat = n_unsat = 0
    for seed in (21, 22, 25, 26):
        # (hub clauses of 4..6 literals: tight enough that a watcher lost from a hub's list costs a propagation)
        lits, offsets, n_vars, n_hubs, rng = long_list_formula(seed, n_vars=240, n_long=90, per_hub=40, hub_len=(3, 6))
        core = np.arange(n_hubs, n_hubs + 120)
        extra = []
        for _ in range(int(3.0 * 120)):
            vs = rng.choice(core, size=3, replace=False)
            extra.append([int(v + 1) * (1 if rng.random() < 0.5 else -1) for v in vs])
        for h in range(n_hubs):                       # the hubs are implied now and then: core literal -> hub
            for _ in range(3):
                extra.append([-int(rng.choice(core) + 1), h + 1])
        lits = np.concatenate([lits, np.array([l for c in extra for l in c], dtype=np.int32)])
        offsets = np.concatenate([offsets, offsets[-1] + np.cumsum([len(c) for c in extra]).astype(np.uint64)])
        o = ora.OracleSolver()
        o.add_cnf(lits, offsets)
        want = o.solve()
        s = emu_solver(workers=2, slice_conflicts=400, reduce_first=60, reduce_inc=20, lds_val=lds_val, simp=-1)
        s.add_cnf(lits, offsets)
        r = s.solve()
        assert (r == SolverResult.Sat) == (want == 10) and r in (SolverResult.Sat, SolverResult.Unsat), (seed, r, want)
        if r == SolverResult.Sat:
            n_sat += 1
            assert ora.check_model(lits, offsets, s.full_solution(n_vars)) == -1
        else:
            n_unsat += 1
        st = s.stats()
        assert st["conflicts"] >= 20 and st["n_move"] > 1000, (seed, st["conflicts"], st["n_move"])
        s.close()
    assert n_sat >= 1 and n_unsat >= 1, (n_sat, n_unsat)
