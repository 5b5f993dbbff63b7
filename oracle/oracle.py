"""ORACLE (test infrastructure, never shipped, never on the product path).

ctypes front-end of oracle/liboracle.so (cdcl.c + check.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("cdcl.c", "check.c", "oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OraStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("propagations", "decisions", "conflicts", "restarts", "learnts", "learnt_literals",
                 "reduce_dbs", "n_clauses", "max_var")] + \
               [("avg_clause_len", ctypes.c_double), ("solve_seconds", ctypes.c_double)] + \
               [(n, ctypes.c_uint64) for n in
                ("n_deq", "n_watch", "n_cl_lit", "n_move", "n_enq", "n_sat", "n_unsat", "n_terminated")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32p, u64p, i8p = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
        L.ora_new.restype = vp
        L.ora_free.argtypes = [vp]
        L.ora_reserve.argtypes = [vp, ctypes.c_uint64]
        L.ora_add_cnf.argtypes = [vp, i32p, u64p, ctypes.c_uint64]
        L.ora_solve.argtypes = [vp, i32p, ctypes.c_int32, ctypes.c_int64]
        L.ora_model.argtypes = [vp, i8p, ctypes.c_uint64]
        L.ora_stats.argtypes = [vp, ctypes.POINTER(OraStats)]
        L.ora_interrupt.argtypes = [vp]
        L.ora_enable_proof.argtypes = [vp]
        L.ora_proof_len.argtypes = [vp]
        L.ora_proof_len.restype = ctypes.c_int64
        L.ora_proof.argtypes = [vp]
        L.ora_proof.restype = ctypes.POINTER(ctypes.c_int32)
        L.ora_check_model.argtypes = [i32p, u64p, ctypes.c_uint64, i8p, ctypes.c_uint64]
        L.ora_check_model.restype = ctypes.c_int64
        L.ora_bcp.argtypes = [i32p, u64p, ctypes.c_uint64, ctypes.c_uint64, i32p, ctypes.c_uint64, i8p,
                              ctypes.POINTER(ctypes.c_int32), u64p]
        L.ora_check_rup.argtypes = [i32p, u64p, ctypes.c_uint64, ctypes.c_uint64, i32p, ctypes.c_int64]
        _lib = L
    return _lib


def to_csr(clauses):
    """list of lists of DIMACS literals -> (lits int32, offsets uint64)"""
    offsets = np.zeros(len(clauses) + 1, dtype=np.uint64)
    if clauses:
        offsets[1:] = np.cumsum([len(c) for c in clauses], dtype=np.uint64)
    lits = np.fromiter((l for c in clauses for l in c), dtype=np.int32, count=int(offsets[-1]))
    return lits, offsets


def _p(a):
    return a.ctypes.data if a is not None and a.size else None


class OracleSolver:
    """Single-thread CPU CDCL restatement ('port' baseline)."""

    def __init__(self):
        self._L = lib()
        self._h = self._L.ora_new()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ora_free(self._h)
            self._h = None

    def reserve(self, n_vars):
        self._L.ora_reserve(self._h, n_vars)

    def add_cnf(self, lits, offsets):
        lits = np.ascontiguousarray(lits, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._L.ora_add_cnf(self._h, _p(lits), _p(offsets), len(offsets) - 1)

    def enable_proof(self):
        self._L.ora_enable_proof(self._h)

    def solve(self, assumptions=(), conflict_budget=0):
        a = np.asarray(list(assumptions), dtype=np.int32)
        return self._L.ora_solve(self._h, _p(a), len(a), conflict_budget)

    def model(self, n_vars):
        out = np.zeros(n_vars, dtype=np.int8)
        if self._L.ora_model(self._h, _p(out), n_vars) != 0:
            raise RuntimeError("no model")
        return out

    def stats(self):
        st = OraStats()
        self._L.ora_stats(self._h, ctypes.byref(st))
        return st.as_dict()

    def proof(self):
        n = self._L.ora_proof_len(self._h)
        if n == 0:
            return np.zeros(0, dtype=np.int32)
        return np.ctypeslib.as_array(self._L.ora_proof(self._h), shape=(n,)).copy()


def check_model(lits, offsets, model):
    lits = np.ascontiguousarray(lits, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    model = np.ascontiguousarray(model, dtype=np.int8)
    return lib().ora_check_model(_p(lits), _p(offsets), len(offsets) - 1, _p(model), len(model))


def bcp(lits, offsets, n_vars, decisions):
    """returns (conflict:int, values:int8[n_vars], trail_len, dequeued)"""
    lits = np.ascontiguousarray(lits, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    dec = np.asarray(list(decisions), dtype=np.int32)
    vals = np.zeros(n_vars, dtype=np.int8)
    tl = ctypes.c_int32(0)
    cnt = np.zeros(4, dtype=np.uint64)
    c = lib().ora_bcp(_p(lits), _p(offsets), len(offsets) - 1, n_vars, _p(dec), len(dec), _p(vals),
                      ctypes.byref(tl), _p(cnt))
    return c, vals, tl.value, int(cnt[0])


def check_rup(lits, offsets, n_vars, proof):
    lits = np.ascontiguousarray(lits, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    proof = np.ascontiguousarray(proof, dtype=np.int32)
    return lib().ora_check_rup(_p(lits), _p(offsets), len(offsets) - 1, n_vars, _p(proof), len(proof))
