"""Shared helpers for the test-suite (test infrastructure)."""
import ctypes
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


VERDICTS = golden("verdicts.json")


def terrain_rows(name):
    if name.startswith("rect"):
        w, h = (int(v) for v in name[4:].split("x"))
        return ["X" * w] * h
    return VERDICTS["terrains"][name]


def make_grid(name):
    from timberborn_support_solver_amd import WorldGrid
    return WorldGrid.from_rows(terrain_rows(name))


def platform_defs(pset):
    from timberborn_support_solver_amd import PLATFORMS_DEFAULT
    return PLATFORMS_DEFAULT if pset == "default" else [(1, 1)]


_emu = None


def emu_lib():
    """The wavefront-emulator build of the solver library (tests/emu): CPU-side logic tests only."""
    global _emu
    if _emu is None:
        import subprocess
        d = os.path.join(ROOT, "tests", "emu")
        if os.environ.get("TBS_EMU_LIB"):     # e.g. the sanitizer build of tests/emu/asan.mk
            _emu = ctypes.CDLL(os.environ["TBS_EMU_LIB"])
            return _emu
        subprocess.check_call(["make", "-C", d, "libmi355sat_emu.so"], stdout=subprocess.DEVNULL)
        _emu = ctypes.CDLL(os.path.join(d, "libmi355sat_emu.so"))
    return _emu


def splitmix64(seed):
    z = seed & (2 ** 64 - 1)
    while True:
        z = (z + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        x = z
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        yield x ^ (x >> 31)


def scripted_decisions(enc, grid, seed, n_decisions, p_positive=0.35):
    """Seeded decision scripts over platform variables: mostly negative
    literals (a positive 5x5 forces a lot), so that many scripts reach a fixpoint."""
    g = splitmix64(seed)
    dims = enc.platform_dims()
    dec = []
    for _ in range(n_decisions):
        r = next(g)
        x, y = (r >> 8) % grid.width, (r >> 24) % grid.height
        d = dims[(r >> 40) % len(dims)]
        v = enc.platform_var(int(x), int(y), d)
        pos = ((r >> 52) % 1000) < int(1000 * p_positive)
        dec.append(v if pos else -v)
    return dec


_oracle_enc = {}


def oracle_encoding(enc, grid):
    """The oracle's literal restatement of the encoder for the same terrain / platform set (cached: it is
    pure Python).  Its variable numbering equals the product's (tests/test_encoder.py proves the CNFs
    bit-exact), so product models can be read through it."""
    from oracle import encoder_oracle as eo
    key = (tuple(grid.rows()), tuple(enc.defs))
    if key not in _oracle_enc:
        _oracle_enc[key] = eo.Encoding(list(enc.defs), eo.grid_from_rows(grid.rows()))
    return _oracle_enc[key]


def check_sat_answer(cnf, model, enc, grid, k):
    """A SAT answer is right iff the model satisfies every clause, the layout validates and has <= k
    platforms.  Layout and validity are derived twice: by the product (libtbs_host.so) and by the
    oracle's restatement of platform_layout.rs (oracle/layout_oracle.py) - on the GPU box the product's
    validator must not be its own judge."""
    from oracle import layout_oracle as lo, oracle as ora
    from timberborn_support_solver_amd import PlatformLayout
    assert ora.check_model(cnf.lits, cnf.offsets, model) == -1
    lay = PlatformLayout.from_assignment(model[:enc.n_vars], enc)
    assert lay.validate(grid).is_valid()
    assert lay.platform_count() <= k
    o = oracle_encoding(enc, grid)
    q = lo.from_assignment(np.asarray(model[:enc.n_vars]).tolist(), o)
    assert lo.is_valid(lo.validate(q, o.grid)) and lo.platform_count(q) <= k
    assert sorted(lay.platforms()) == sorted((x, y, d[0], d[1], int(r)) for (x, y), (d, r) in q.items())
    return lay


def assert_ring_records_are_implied(solver, cnf, max_records=None):
    """Exchange soundness: every clause in the learnt-clause exchange ring must follow from the caller's
    formula ALONE (workers attach ring records under any assumption set and declare UNSAT when one is
    falsified at level 0).  The oracle solver refutes formula AND NOT(clause) for each record."""
    from oracle import oracle as ora
    recs = solver.debug_share_ring()
    o = ora.OracleSolver()
    o.add_cnf(cnf.lits, cnf.offsets)
    o.reserve(cnf.n_vars)
    seen = set()
    for c in recs[:max_records]:
        assert 1 <= len(c) <= 31 and all(l != 0 and abs(l) <= cnf.n_vars for l in c), c
        key = tuple(sorted(c))
        if key in seen:
            continue
        seen.add(key)
        assert o.solve([-l for l in c]) == 20, ("exchange ring holds a clause the formula does not imply", c)
    return len(recs)


def long_list_formula(seed, n_vars=360, n_long=140, n_hubs=6, per_hub=44, hub_len=(9, 40)):
    """A formula cut for the BCP step's side paths: clauses of 10..48 literals (tails to scan, several per step), a few
    hub literals watched by 44 long clauses each (watch lists far longer than a lane group: the flat remainder, its
    in-place compaction and the re-queueing when two groups meet in one clause), binary and ternary chains between."""
    rng = np.random.default_rng(seed)
    clauses = []

    def rand_clause(k, first=None):
        vs = rng.choice(np.arange(n_hubs, n_vars), size=k, replace=False)
        c = [int(v + 1) * (1 if rng.random() < 0.5 else -1) for v in vs]
        return ([first] + c) if first is not None else c

    for _ in range(n_long):
        clauses.append(rand_clause(int(rng.integers(10, 49))))
    for h in range(n_hubs):                          # -(h+1) in a watched position of every one of its clauses
        for _ in range(per_hub):
            c = rand_clause(int(rng.integers(hub_len[0], hub_len[1])), first=-(h + 1))
            if rng.random() < 0.5:
                c[0], c[1] = c[1], c[0]
            clauses.append(c)
    for _ in range(n_vars):                          # implication chains
        a, b, c = (int(x) for x in rng.choice(np.arange(n_hubs, n_vars), size=3, replace=False))
        sa, sb, sc = (1 if rng.random() < 0.5 else -1 for _ in range(3))
        clauses.append([sa * (a + 1), sb * (b + 1)] if rng.random() < 0.35 else [sa * (a + 1), sb * (b + 1), sc * (c + 1)])
    lits = np.array([l for c in clauses for l in c], dtype=np.int32)
    offsets = np.cumsum([0] + [len(c) for c in clauses]).astype(np.uint64)
    return lits, offsets, n_vars, n_hubs, rng
