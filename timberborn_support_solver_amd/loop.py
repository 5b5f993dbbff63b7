"""The outer decreasing-k refinement loop, a Python mirror of `solver_loop`
(crates/repl/src/main.rs:280-366) and `run_solver` (crates/repl/src/solver_runner.rs:8-20).

Per iteration, as in the reference: with_limits -> into_cnf -> a FRESH solver ->
add_cnf -> solve -> (Sat) full_solution -> PlatformLayout::from_assignment ->
`k := platform_count - 1`; stop on Unsat / Interrupted / a layout with no
platforms.  The observable messages are the reference's (main.rs:332,336,342,348).
"""
import time

from .encoder import PlatformLayout, PlatformLimits
from .solver import Mi355Sat, SolverResult


def run_solver(solver, cnf):
    """run_solver<S>: add the CNF, hand back (result_thunk, interrupter).  The reference moves
    the solver to a blocking thread; here the caller decides where to call the thunk."""
    solver.add_cnf(cnf.lits, cnf.offsets)
    solver.reserve(cnf.n_vars)
    interrupter = solver.interrupter()
    return (lambda: (solver.solve(), solver)), interrupter


def solver_loop(grid, encoding, limits, make_solver=None, out=print, on_interrupter=None, max_iterations=None):
    """Returns a list of per-iteration records:
    {k, result, count, valid, seconds, stats}.  `make_solver` builds the backend
    (default Mi355Sat()), mirroring `GlucoseSimp::default()` at main.rs:295."""
    make_solver = make_solver or (lambda: Mi355Sat())
    limits = PlatformLimits(dict(limits.card_limits))
    history = []
    while max_iterations is None or len(history) < max_iterations:
        cnf = encoding.with_limits_into_cnf(limits)
        solver = make_solver()
        thunk, interrupter = run_solver(solver, cnf)
        if on_interrupter:
            on_interrupter(interrupter)
        t0 = time.perf_counter()
        result, solver = thunk()
        dt = time.perf_counter() - t0
        rec = {"k": limits.card_limits.get((1, 1)), "result": result, "count": None, "valid": None,
               "seconds": dt, "stats": solver.stats()}
        history.append(rec)
        if result == SolverResult.Unsat:
            out("No solution found for the current constraints")
            solver.close()
            return history
        if result == SolverResult.Interrupted:
            out("Solver interrupted")
            solver.close()
            return history
        layout = PlatformLayout.from_assignment(solver.full_solution(encoding.n_vars), encoding)
        solver.close()
        count = layout.platform_count()
        rec["count"] = count
        rec["layout"] = layout
        if count == 0:
            out("Found a solution with no platforms - aborting")
            return history
        limits.card_limits[(1, 1)] = count - 1
        out(f"Solution found ({count} platforms total)")
        for (w, h), n in sorted(layout.platform_stats().items()):
            out(f"{w}x{h}: {n}")
        validation = layout.validate(grid)
        rec["valid"] = validation.is_valid()
        out("Solution validation OK" if rec["valid"] else "Solution validation FAILED")
    return history


def weight_loop(grid, encoding, limits, make_solver=None, out=print, max_iterations=None):
    """The GUI's weight-minimising loop (crates/gui/src/app.rs:148-175, 235-245): solve, take the layout
    (after run_trivial_optimization), set weight_limit = total_weight - 1 and solve again until Unsat /
    Interrupted or the weight cannot drop further.  Returns the per-iteration records."""
    make_solver = make_solver or (lambda: Mi355Sat())
    limits = PlatformLimits(dict(limits.card_limits), dict(limits.weights), limits.weight_limit)
    history = []
    while max_iterations is None or len(history) < max_iterations:
        cnf = encoding.with_limits_into_cnf(limits)
        solver = make_solver()
        thunk, _ = run_solver(solver, cnf)
        result, solver = thunk()
        rec = {"weight_limit": limits.weight_limit, "result": result, "weight": None, "valid": None, "stats": solver.stats()}
        history.append(rec)
        if result != SolverResult.Sat:
            out("No solution found for the current constraints" if result == SolverResult.Unsat else "Solver interrupted")
            solver.close()
            return history
        layout = PlatformLayout.from_assignment(solver.full_solution(encoding.n_vars), encoding)
        solver.close()
        layout.run_trivial_optimization(grid)
        weight = layout.total_weight(limits.weights)
        rec.update(weight=weight, valid=layout.validate(grid).is_valid(), layout=layout)
        out(f"Got a solution with weight {weight}")
        if weight - 1 <= 0:
            return history
        limits.weight_limit = weight - 1
    return history
