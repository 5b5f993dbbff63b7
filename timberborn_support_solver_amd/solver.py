"""Python view of the solver boundary (include/mi355sat.h).

`Mi355Sat` mirrors the part of rustsat's `Solve + Interrupt + SolveStats` that
the reference exercises:

    GlucoseSimp::default()            crates/repl/src/main.rs:295      -> Mi355Sat()
    solver.add_cnf(cnf)               crates/repl/src/solver_runner.rs:12
    solver.interrupter()              crates/repl/src/solver_runner.rs:13
    interrupter.interrupt()           crates/repl/src/main.rs:316
    solver.solve() -> SolverResult    crates/repl/src/solver_runner.rs:16
    solver.full_solution()            crates/repl/src/main.rs:329
    solver.stats()                    crates/repl/src/main.rs:363

plus the two batched entry points the sharded sweep and the BCP configuration
use (solve_batch / propagate_batch).  Errors come back as `SolverError`, the
analogue of the reference's `anyhow::Result`.
"""
import ctypes
import enum
import threading

import numpy as np

from . import _lib


class SolverResult(enum.Enum):  # rustsat::solvers::SolverResult
    Sat = 10
    Unsat = 20
    Interrupted = 0


class SolverError(RuntimeError):
    pass


class Mi355SatOpts(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("workers", ctypes.c_int32), ("conflict_budget", ctypes.c_int64),
                ("slice_conflicts", ctypes.c_int32), ("seed", ctypes.c_uint64), ("verbose", ctypes.c_int32),
                ("reduce_first", ctypes.c_int32), ("reduce_inc", ctypes.c_int32), ("lds_val", ctypes.c_int32),
                ("max_groups", ctypes.c_int32), ("slice_ms", ctypes.c_int32), ("cube_split", ctypes.c_int32), ("share", ctypes.c_int32), ("share_lbd", ctypes.c_int32), ("share_len", ctypes.c_int32),
                ("share_interval", ctypes.c_int32), ("var_order", ctypes.c_int32), ("ramp", ctypes.c_int32), ("one_per_simd", ctypes.c_int32), ("simp", ctypes.c_int32), ("phase_mix", ctypes.c_int32), ("rephase", ctypes.c_int32), ("restart_k_pct", ctypes.c_int32), ("restart_k2_pct", ctypes.c_int32), ("import_pct", ctypes.c_int32), ("vivify", ctypes.c_int32), ("rebalance", ctypes.c_int32), ("deterministic", ctypes.c_int32)]


class Mi355SatStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("propagations", "decisions", "conflicts", "restarts", "learnts", "learnt_literals",
                 "reduce_dbs", "n_clauses", "max_var")] + \
               [("avg_clause_len", ctypes.c_double), ("solve_seconds", ctypes.c_double),
                ("kernel_seconds", ctypes.c_double), ("kernel_launches", ctypes.c_uint64)] + \
               [(n, ctypes.c_uint64) for n in
                ("n_deq", "n_watch", "n_cl_lit", "n_move", "n_enq", "n_sat", "n_unsat", "n_terminated",
                 "bcp_steps", "bcp_requeued")] + \
               [("shared_exported", ctypes.c_uint64), ("shared_imported", ctypes.c_uint64), ("shared_imported_units", ctypes.c_uint64),
                ("simp_units", ctypes.c_uint64), ("simp_equivalences", ctypes.c_uint64), ("simp_clauses_removed", ctypes.c_uint64),
                ("workers", ctypes.c_uint64), ("simp_eliminated", ctypes.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def algorithmic_bytes(stats):
    """12*n_deq + 9*n_watch + 5*n_cl_lit + 8*n_move + 13*n_enq."""
    return (12 * stats["n_deq"] + 9 * stats["n_watch"] + 5 * stats["n_cl_lit"] + 8 * stats["n_move"]
            + 13 * stats["n_enq"])


def _bind(L):
    vp = ctypes.c_void_p
    L.mi355sat_abi_sizes.restype = ctypes.c_uint64
    L.mi355sat_abi_sizes.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    st_size = ctypes.c_uint64(0)
    if (L.mi355sat_abi_sizes(ctypes.byref(st_size)), st_size.value) != (ctypes.sizeof(Mi355SatOpts), ctypes.sizeof(Mi355SatStats)):
        raise RuntimeError("libmi355sat was built from another include/mi355sat.h than this binding mirrors (struct sizes differ)")
    L.mi355sat_new.restype = vp
    L.mi355sat_new.argtypes = [ctypes.POINTER(Mi355SatOpts)]
    L.mi355sat_free.argtypes = [vp]
    L.mi355sat_signature.restype = ctypes.c_char_p
    L.mi355sat_last_error.restype = ctypes.c_char_p
    L.mi355sat_last_error.argtypes = [vp]
    L.mi355sat_add_cnf.argtypes = [vp, vp, vp, ctypes.c_uint64]
    L.mi355sat_add.argtypes = [vp, ctypes.c_int32]
    L.mi355sat_reserve.argtypes = [vp, ctypes.c_uint64]
    L.mi355sat_solve.argtypes = [vp]
    L.mi355sat_solve_batch.argtypes = [vp, vp, vp, ctypes.c_uint64, vp, ctypes.c_int]
    L.mi355sat_sweep_begin.argtypes = [vp, vp, vp, ctypes.c_uint64]
    L.mi355sat_sweep_step.argtypes = [vp, vp, vp]
    L.mi355sat_sweep_end.argtypes = [vp]
    L.mi355sat_sweep_drop.argtypes = [vp, vp, ctypes.c_uint64]
    L.mi355sat_sweep_reopen.argtypes = [vp, vp, ctypes.c_uint64]
    L.mi355sat_sweep_set_weights.argtypes = [vp, vp, ctypes.c_uint64]
    L.mi355sat_sweep_model_of.argtypes = [vp, ctypes.c_uint64, vp, ctypes.c_uint64]
    L.mi355sat_propagate_batch.argtypes = [vp, vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp, ctypes.c_int32]
    L.mi355sat_val.argtypes = [vp, ctypes.c_int32]
    L.mi355sat_val.restype = ctypes.c_int32
    L.mi355sat_model.argtypes = [vp, vp, ctypes.c_uint64]
    L.mi355sat_model_of.argtypes = [vp, ctypes.c_uint64, vp, ctypes.c_uint64]
    L.mi355sat_interrupt.argtypes = [vp]
    L.mi355sat_stats.argtypes = [vp, ctypes.POINTER(Mi355SatStats)]
    L.mi355sat_set_proof_path.argtypes = [vp, ctypes.c_char_p]
    L.mi355sat_debug_share_ring.argtypes = [vp, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    L.mi355sat_share_export.argtypes = [vp, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.mi355sat_share_import.argtypes = [vp, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    return L


_bound = {}


def _p(a):
    return a.ctypes.data if a is not None and a.size else None


class Interrupter:
    """`S::Interrupter`: Send + 'static, callable from another thread while solve() runs."""

    def __init__(self, lib, handle_ref, lock):
        self._lib, self._ref, self._lock = lib, handle_ref, lock

    def interrupt(self):
        with self._lock:   # close() clears the reference under the same lock before it frees the handle
            h = self._ref[0]
            if h:
                self._lib.mi355sat_interrupt(h)


class Mi355Sat:
    def __init__(self, device=-1, workers=0, conflict_budget=0, slice_conflicts=0, seed=0, verbose=0,
                 reduce_first=0, reduce_inc=0, lds_val=0, max_groups=0, slice_ms=0, cube_split=0, share=0, share_lbd=0, share_len=0, share_interval=0, rebalance=0, var_order=0, ramp=0, one_per_simd=0, simp=0, phase_mix=0, rephase=0, restart_k_pct=0, restart_k2_pct=0, import_pct=0, vivify=0, deterministic=0, _lib_override=None):
        # _lib_override: test hook (the wavefront-emulator build under tests/emu); the product
        # always binds the HIP library and fails loudly without it.
        raw = _lib_override if _lib_override is not None else _lib.solver_lib()
        if id(raw) not in _bound:
            _bound[id(raw)] = _bind(raw)
        self._L = _bound[id(raw)]
        opts = Mi355SatOpts(device=device, workers=workers, conflict_budget=conflict_budget,
                            slice_conflicts=slice_conflicts, seed=seed, verbose=verbose,
                            reduce_first=reduce_first, reduce_inc=reduce_inc, lds_val=lds_val, max_groups=max_groups, slice_ms=slice_ms, cube_split=cube_split, share=share, share_lbd=share_lbd, share_len=share_len, share_interval=share_interval, rebalance=rebalance, var_order=var_order, ramp=ramp, one_per_simd=one_per_simd, simp=simp, phase_mix=phase_mix, rephase=rephase, restart_k_pct=restart_k_pct, restart_k2_pct=restart_k2_pct, import_pct=import_pct, vivify=vivify, deterministic=deterministic)
        self._h = self._L.mi355sat_new(ctypes.byref(opts))
        if not self._h:
            raise SolverError("mi355sat_new failed: " + (self._L.mi355sat_last_error(None) or b"").decode())
        self._ref = [self._h]
        self._lock = threading.Lock()
        self._n_vars = 0

    def close(self):
        if getattr(self, "_h", None):
            with self._lock:   # no interrupter may still be inside mi355sat_interrupt(), none may enter afterwards
                self._ref[0] = None
            self._L.mi355sat_free(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc < 0:
            raise SolverError(f"{what} failed ({rc}): " + (self._L.mi355sat_last_error(self._h) or b"").decode())
        return rc

    @staticmethod
    def signature():
        return _bind(_lib.solver_lib()).mi355sat_signature().decode()

    # ---- Solve
    def add_cnf(self, lits, offsets):
        lits = np.ascontiguousarray(lits, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self._L.mi355sat_add_cnf(self._h, _p(lits), _p(offsets), len(offsets) - 1), "add_cnf")
        if lits.size:
            self._n_vars = max(self._n_vars, int(np.abs(lits).max()))

    def add_clause(self, clause):
        for l in clause:
            self._check(self._L.mi355sat_add(self._h, int(l)), "add")
            self._n_vars = max(self._n_vars, abs(int(l)))
        self._check(self._L.mi355sat_add(self._h, 0), "add")

    def reserve(self, n_vars):
        self._check(self._L.mi355sat_reserve(self._h, n_vars), "reserve")
        self._n_vars = max(self._n_vars, n_vars)

    def interrupter(self):
        return Interrupter(self._L, self._ref, self._lock)

    def solve(self):
        return SolverResult(self._check(self._L.mi355sat_solve(self._h), "solve"))

    def solve_batch(self, assumption_sets, stop_at_first=False):
        """assumption_sets: list of lists of DIMACS literals.  Returns [SolverResult]."""
        offs = np.zeros(len(assumption_sets) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(a) for a in assumption_sets])
        flat = np.asarray([l for a in assumption_sets for l in a], dtype=np.int32)
        res = np.zeros(len(assumption_sets), dtype=np.int32)
        self._check(self._L.mi355sat_solve_batch(self._h, _p(flat), _p(offs), len(assumption_sets), _p(res),
                                                 1 if stop_at_first else 0), "solve_batch")
        return [SolverResult(int(r)) for r in res]

    # stepwise form of solve_batch (one kernel slice per step); bench.py times these
    def sweep_begin(self, assumption_sets):
        offs = np.zeros(len(assumption_sets) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(a) for a in assumption_sets])
        flat = np.asarray([l for a in assumption_sets for l in a], dtype=np.int32)
        self._sweep_n = len(assumption_sets)
        self._check(self._L.mi355sat_sweep_begin(self._h, _p(flat), _p(offs), self._sweep_n), "sweep_begin")

    def sweep_step(self):
        """Returns ([SolverResult per instance], n_decided)."""
        res = np.zeros(self._sweep_n, dtype=np.int32)
        nd = ctypes.c_uint64(0)
        self._check(self._L.mi355sat_sweep_step(self._h, _p(res), ctypes.byref(nd)), "sweep_step")
        return [SolverResult(int(r)) for r in res], nd.value

    def sweep_drop(self, instances):
        """Withdraw instances (their answer is implied); their workers join the open ones."""
        idx = np.asarray(list(instances), dtype=np.uint64)
        if len(idx):
            self._check(self._L.mi355sat_sweep_drop(self._h, _p(idx), len(idx)), "sweep_drop")

    def sweep_set_weights(self, weights):
        """Share of the fleet per instance (only the open ones count); takes effect at the next step."""
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self._check(self._L.mi355sat_sweep_set_weights(self._h, _p(w), len(w)), "sweep_set_weights")

    def sweep_reopen(self, instances):
        """Take withdrawn, still undecided instances up again (idle workers move to them)."""
        idx = np.asarray(list(instances), dtype=np.uint64)
        if len(idx):
            self._check(self._L.mi355sat_sweep_reopen(self._h, _p(idx), len(idx)), "sweep_reopen")

    def sweep_solution_of(self, instance, n_vars=None):
        """Model of an instance that already reported Sat, while the sweep is running."""
        n_vars = self._n_vars if n_vars is None else n_vars
        out = np.zeros(n_vars, dtype=np.int8)
        self._check(self._L.mi355sat_sweep_model_of(self._h, instance, _p(out), n_vars), "sweep_solution_of")
        return out

    def sweep_end(self):
        self._check(self._L.mi355sat_sweep_end(self._h), "sweep_end")

    def propagate_batch(self, decision_sets, n_vars=None, repeat=1, want_values=True):
        """Scripted BCP: returns (conflict int32[n], values int8[n, n_vars] or None, trail_len int32[n])."""
        n = len(decision_sets)
        n_vars = self._n_vars if n_vars is None else n_vars
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(a) for a in decision_sets])
        flat = np.asarray([l for a in decision_sets for l in a], dtype=np.int32)
        confl = np.zeros(n, dtype=np.int32)
        tl = np.zeros(n, dtype=np.int32)
        vals = np.zeros((n, n_vars), dtype=np.int8) if want_values else None
        self._check(self._L.mi355sat_propagate_batch(self._h, _p(flat), _p(offs), n, _p(vals) if want_values else None,
                                                     n_vars, _p(confl), _p(tl), repeat), "propagate_batch")
        return confl, vals, tl

    def debug_share_ring(self):
        """Test hook: the clauses in the learnt-clause exchange ring, as lists of DIMACS literals."""
        n = ctypes.c_uint64(0)
        cap = 1 << 16
        while True:
            buf = np.zeros(cap, dtype=np.int32)
            rc = self._L.mi355sat_debug_share_ring(self._h, _p(buf), cap, ctypes.byref(n))
            if rc == -4:
                cap *= 4
                continue
            self._check(rc, "debug_share_ring")
            break
        out, cur = [], []
        for l in buf.tolist():
            if len(out) == n.value:
                break
            if l == 0:
                out.append(cur)
                cur = []
            else:
                cur.append(l)
        return out

    def share_export(self, max_words=1 << 20):
        """The clauses this handle's workers passed on since the last call, for handles on OTHER GPUs that search the same
        formula: int32 array of [lbd, DIMACS literals ..., 0] records (caller's variables); what does not fit max_words waits."""
        buf = np.zeros(max_words, dtype=np.int32)
        nw, nr = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._L.mi355sat_share_export(self._h, _p(buf), max_words, ctypes.byref(nw), ctypes.byref(nr)), "share_export")
        return buf[:nw.value].copy(), nr.value

    def share_import(self, records):
        """Append records exported by another handle (share_export's format) to this handle's exchange ring; returns how
        many were taken."""
        records = np.ascontiguousarray(records, dtype=np.int32)
        nr = ctypes.c_uint64(0)
        self._check(self._L.mi355sat_share_import(self._h, _p(records), len(records), ctypes.byref(nr)), "share_import")
        return nr.value

    def set_proof_path(self, path):
        """DRUP proof of the next plain solve(), in its default configuration: all workers, clause exchange on (one log per
        worker, drained after every slice)."""
        self._check(self._L.mi355sat_set_proof_path(self._h, path.encode() if path else None), "set_proof_path")

    def lit_val(self, lit):
        return self._L.mi355sat_val(self._h, int(lit))

    def full_solution(self, n_vars=None):
        """Assignment for vars 1..n_vars as int8: 1 true, -1 false, 0 unknown."""
        n_vars = self._n_vars if n_vars is None else n_vars
        out = np.zeros(n_vars, dtype=np.int8)
        self._check(self._L.mi355sat_model(self._h, _p(out), n_vars), "full_solution")
        return out

    def solution_of(self, instance, n_vars=None):
        n_vars = self._n_vars if n_vars is None else n_vars
        out = np.zeros(n_vars, dtype=np.int8)
        self._check(self._L.mi355sat_model_of(self._h, instance, _p(out), n_vars), "solution_of")
        return out

    # ---- SolveStats
    def stats(self):
        st = Mi355SatStats()
        self._check(self._L.mi355sat_stats(self._h, ctypes.byref(st)), "stats")
        return st.as_dict()

    def max_var(self):
        return self._n_vars
