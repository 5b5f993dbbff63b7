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


def solver_loop_pair(grid, encoding, limits, make_solver=None, out=print, on_interrupter=None):
    """The reference's sequential loop (main.rs:290-346) with ONE bound of lookahead: while the bound k the reference
    would pose next is being solved, the bound k - 1 runs beside it in a solver of its own (its own CNF, its own stream;
    `make_solver(half=True)` gives each half of the fleet - on rect 28 / 32 half the default fleet decides a bound as
    fast as the whole, profiles/r03_m_knob_sweep1.log).  Near the optimum the loop's last two bounds are the expensive
    ones (the last model, then the refutation): this way they overlap.  What the caller sees is the reference's loop:
    the same messages in the same order, records only for the bounds the sequential loop would have posed; a
    speculative solve whose answer is implied by the other's is interrupted and dropped.
      k SAT with count c  -> next bound c - 1: if that is k - 1 the running partner is promoted, else both restart
      k UNSAT             -> done (the partner, a lower bound, is implied UNSAT)
      k - 1 SAT (first)   -> k is implied SAT: its solve is dropped, the partner's model is the iteration's
      k - 1 UNSAT (first) -> k decides alone: SAT means k is the optimum, and the loop ends with k - 1's refutation"""
    import threading
    k = limits.card_limits[(1, 1)]
    history = []

    def make(cnf):
        if make_solver is None:      # half the default fleet each (1024 workers above 20 000 clauses, mi355sat.h)
            return Mi355Sat(workers=512) if cnf.n_clauses > 20000 else Mi355Sat()
        try:
            return make_solver(half=True)
        except TypeError:            # a factory without the keyword: whatever it makes, twice
            return make_solver()

    def start(bound):
        cnf = encoding.with_limits_into_cnf(PlatformLimits({(1, 1): bound}))
        solver = make(cnf)
        thunk, interrupter = run_solver(solver, cnf)
        if on_interrupter:
            on_interrupter(interrupter)
        job = {"k": bound, "solver": solver, "intr": interrupter, "done": threading.Event(), "t0": time.perf_counter()}

        def work():
            try:
                job["result"] = thunk()[0]
            except Exception as e:        # surfaces in the caller's thread
                job["error"] = e
            job["seconds"] = time.perf_counter() - job["t0"]
            job["done"].set()

        job["thread"] = threading.Thread(target=work, daemon=True)
        job["thread"].start()
        return job

    def drop(job):
        if job is None:
            return
        job["intr"].interrupt()
        job["thread"].join()
        job["solver"].close()

    def finish(job):
        job["thread"].join()
        if "error" in job:
            raise job["error"]
        return job["result"]

    a, b = start(k), (start(k - 1) if k >= 1 else None)
    while True:
        # wait for whichever decides something
        while not a["done"].is_set() and not (b is not None and b["done"].is_set()):
            a["done"].wait(0.005)
        if not a["done"].is_set():      # the partner (k - 1) answered first
            rb = finish(b)
            if rb == SolverResult.Sat:   # then k is satisfiable too: the partner's model is this iteration's
                drop(a)
                a, b = b, None
            elif rb == SolverResult.Unsat:      # k decides alone; remember the refutation
                finish(a)
                b["solver"].close()
                b = {"k": b["k"], "refuted": True, "seconds": b["seconds"]}
            else:                        # interrupted from outside
                finish(a)
                b["solver"].close()
                b = None
        ra = finish(a)
        rec = {"k": a["k"], "result": ra, "count": None, "valid": None, "seconds": a["seconds"], "stats": a["solver"].stats()}
        history.append(rec)
        if ra != SolverResult.Sat:
            out("No solution found for the current constraints" if ra == SolverResult.Unsat else "Solver interrupted")
            a["solver"].close()
            if b is not None and not b.get("refuted"):
                drop(b)
            return history
        layout = PlatformLayout.from_assignment(a["solver"].full_solution(encoding.n_vars), encoding)
        a["solver"].close()
        count = layout.platform_count()
        rec["count"], rec["layout"] = count, layout
        if count == 0:
            out("Found a solution with no platforms - aborting")
            if b is not None and not b.get("refuted"):
                drop(b)
            return history
        out(f"Solution found ({count} platforms total)")
        for (w, h), n in sorted(layout.platform_stats().items()):
            out(f"{w}x{h}: {n}")
        rec["valid"] = layout.validate(grid).is_valid()
        out("Solution validation OK" if rec["valid"] else "Solution validation FAILED")
        nxt = count - 1
        if b is not None and b.get("refuted") and b["k"] >= nxt:      # the next bound is refuted already
            history.append({"k": nxt, "result": SolverResult.Unsat, "count": None, "valid": None, "seconds": b["seconds"], "stats": {}})
            out("No solution found for the current constraints")
            return history
        if b is not None and not b.get("refuted") and b["k"] == nxt:
            a, b = b, (start(nxt - 1) if nxt >= 1 else None)          # the partner is the next bound: promote it
        else:
            if b is not None and not b.get("refuted"):
                drop(b)
            a, b = start(nxt), (start(nxt - 1) if nxt >= 1 else None)


def solver_loop_fan(grid, encoding, limits, make_solver=None, out=print, on_interrupter=None, width=4):
    """The reference's sequential loop (main.rs:290-346) with `width - 1` bounds of lookahead: the bound k the reference
    would pose next and the bounds k - 1 ... k - width + 1 run side by side, each in a solver of its own (its own CNF
    as the reference makes it, its own stream and host thread, 1 / width of the default fleet: a quarter of it decides a
    hard bound at two thirds of the whole fleet's speed, profiles/r03_m_knob_sweep1.log).  Near the optimum a ladder
    spends as long on its last few models as on the refutation (rect 32x32: 2 + 2 + 5 + 10 s of models, 13 s of
    refutation): this way they overlap.  What the caller sees is the reference's loop - its messages in its order, one
    record per improvement and the refuting bound:
      any bound b SAT with count c   -> every bound >= c is answered: their solves are dropped, the model is the
                                        iteration's (for the bound the reference would have posed), next bound c - 1
      any bound b UNSAT              -> every bound <= b is refuted: their solves are dropped; if b is the bound the
                                        reference would pose now the loop ends, else it ends as soon as a model's
                                        count - 1 reaches a refuted bound"""
    import threading
    front = limits.card_limits[(1, 1)]
    history, jobs, refuted = [], {}, {"k": -1, "seconds": 0.0}

    def make(cnf):
        if make_solver is None:
            return Mi355Sat(workers=max(256, 1024 // width)) if cnf.n_clauses > 20000 else Mi355Sat()
        try:
            return make_solver(fraction=width)
        except TypeError:            # a factory without the keyword: whatever it makes, `width` times
            return make_solver()

    def start(bound):
        cnf = encoding.with_limits_into_cnf(PlatformLimits({(1, 1): bound}))
        solver = make(cnf)
        thunk, interrupter = run_solver(solver, cnf)
        if on_interrupter:
            on_interrupter(interrupter)
        job = {"k": bound, "solver": solver, "intr": interrupter, "done": threading.Event(), "t0": time.perf_counter()}

        def work():
            try:
                job["result"] = thunk()[0]
            except Exception as e:        # surfaces in the caller's thread
                job["error"] = e
            job["seconds"] = time.perf_counter() - job["t0"]
            job["done"].set()

        job["thread"] = threading.Thread(target=work, daemon=True)
        job["thread"].start()
        jobs[bound] = job

    def drop(bound):
        job = jobs.pop(bound)
        job["intr"].interrupt()
        job["thread"].join()
        job["solver"].close()

    def drop_all():
        for b in list(jobs):
            drop(b)

    def fill():
        for b in range(front, max(front - width, refuted["k"]), -1):
            if b >= 0 and b not in jobs:
                start(b)

    fill()
    while True:
        done = [b for b, j in jobs.items() if j["done"].is_set()]
        if not done:
            next(iter(jobs.values()))["done"].wait(0.005)
            continue
        b = min(done)                    # (the lowest answered bound says the most)
        job = jobs.pop(b)
        job["thread"].join()
        if "error" in job:
            drop_all()
            job["solver"].close()
            raise job["error"]
        r = job["result"]
        if r == SolverResult.Unsat:
            job["solver"].close()
            if b > refuted["k"]:
                refuted["k"], refuted["seconds"] = b, job["seconds"]
            for x in [x for x in jobs if x <= b]:
                drop(x)
            if b >= front:
                history.append({"k": front, "result": r, "count": None, "valid": None, "seconds": job["seconds"], "stats": {}})
                out("No solution found for the current constraints")
                drop_all()
                return history
            continue
        if r != SolverResult.Sat:        # interrupted from outside: the loop ends like the reference's
            history.append({"k": front, "result": r, "count": None, "valid": None, "seconds": job["seconds"], "stats": job["solver"].stats()})
            out("Solver interrupted")
            job["solver"].close()
            drop_all()
            return history
        rec = {"k": front, "result": r, "count": None, "valid": None, "seconds": job["seconds"], "stats": job["solver"].stats()}
        history.append(rec)
        layout = PlatformLayout.from_assignment(job["solver"].full_solution(encoding.n_vars), encoding)
        job["solver"].close()
        count = layout.platform_count()
        rec["count"], rec["layout"] = count, layout
        for x in [x for x in jobs if x >= count]:      # answered by this model
            drop(x)
        if count == 0:
            out("Found a solution with no platforms - aborting")
            drop_all()
            return history
        out(f"Solution found ({count} platforms total)")
        for (w, h), n in sorted(layout.platform_stats().items()):
            out(f"{w}x{h}: {n}")
        rec["valid"] = layout.validate(grid).is_valid()
        out("Solution validation OK" if rec["valid"] else "Solution validation FAILED")
        front = count - 1
        if front <= refuted["k"]:        # the next bound is refuted already
            history.append({"k": front, "result": SolverResult.Unsat, "count": None, "valid": None, "seconds": refuted["seconds"], "stats": {}})
            out("No solution found for the current constraints")
            drop_all()
            return history
        fill()


def solver_loop_sweep(grid, encoding, limits, make_solver=None, out=print, on_interrupter=None, time_limit=None,
                      specialize_after=2.0, lookahead=False):
    """The same refinement as ONE batch: every bound k0, k0-1, ..., 0 is an assumption set over one
    CNF built for k0 (`with_limits_into_cnf(sweep=True)`), all solved concurrently on the device.  A SAT model
    with c platforms answers every bound >= c, an UNSAT bound every bound below it; those instances are
    withdrawn (`sweep_drop`) and their workers join the open ones.  Done when max UNSAT k + 1 == min count.
    Prints what the reference loop prints for the iterations it would still have to make (the best layout,
    then the refuting bound) and returns records shaped like solver_loop's.  Only the `-l1:K` form.

    The batch is the fast way DOWN (a dozen easy bounds decided in a second or two), not the fast way to the last
    refutation: a bound posed as an assumption over the totalizer of a looser one is refuted much more slowly than
    the same bound as the reference poses it, with its own CNF (measured on rect 26x26: k = 10 not refuted in 160 s
    and 1.8e8 conflicts inside the batch, 60-70 s and 5-7e7 conflicts on its own: at level 0 the bound's unit cuts
    the totalizer down, as an assumption it adds a literal and a level to every learnt clause).  So the batch runs
    for `specialize_after` seconds - on rect 26x26 that takes the count from 25 to 11 - then ends, and the
    reference's own sequential loop (a fresh solver and a fresh CNF per bound, main.rs:292-295) finishes from the
    best count.  (Waiting until at most two bounds were open cost rect 32x32 166 s in the batch.)"""
    if set(limits.card_limits) != {(1, 1)} or limits.weights or limits.weight_limit is not None:
        raise ValueError("solver_loop_sweep handles a single 1x1 cardinality limit; use solver_loop / weight_loop")
    # first iteration exactly as the reference makes it (the start bound is loose: its totalizer would only
    # burden the batch); the batch then covers count-1 .. 0
    first = solver_loop(grid, encoding, limits, make_solver=make_solver, out=out, on_interrupter=on_interrupter,
                        max_iterations=1)
    if first[-1]["result"] != SolverResult.Sat or not first[-1]["count"]:
        return first
    t0 = time.perf_counter()
    below = _sweep_below(grid, encoding, first[-1]["count"] - 1, make_solver, out, on_interrupter, time_limit, specialize_after)
    if not below or below[-1]["result"] != "specialize":
        return first + below
    below.pop()
    best = (below[-1] if below else first[-1])["count"]
    hist = first + below
    if time_limit is not None and time.perf_counter() - t0 > time_limit:
        out("Solver interrupted")
        return hist + [{"k": best - 1, "result": SolverResult.Interrupted, "count": None, "valid": None, "seconds": 0.0, "stats": {}}]
    timer = None
    interrupters = []
    expired = []

    def note(i):
        # (the limit may run out between two bounds, when the previous solver is closed and the next one not yet
        # registered: a solver registered after the deadline is interrupted at once instead of running unlimited)
        interrupters.append(i)
        if expired:
            i.interrupt()
        if on_interrupter:
            on_interrupter(i)

    def on_deadline():
        expired.append(1)
        for i in interrupters[-max(2, int(lookahead)):]:      # (with lookahead several solvers run at a time)
            i.interrupt()

    if time_limit is not None:   # the rest of the time budget holds for the sequential part as a whole
        import threading
        timer = threading.Timer(max(0.0, time_limit - (time.perf_counter() - t0)), on_deadline)
        timer.start()
    try:
        if lookahead is not True and int(lookahead) > 2:      # lookahead = N: N bounds at a time
            rest = solver_loop_fan(grid, encoding, PlatformLimits({(1, 1): best - 1}), make_solver=make_solver, out=out, on_interrupter=note,
                                   width=int(lookahead))
        elif lookahead:
            rest = solver_loop_pair(grid, encoding, PlatformLimits({(1, 1): best - 1}), make_solver=make_solver, out=out, on_interrupter=note)
        else:
            rest = solver_loop(grid, encoding, PlatformLimits({(1, 1): best - 1}), make_solver=make_solver, out=out, on_interrupter=note)
    finally:
        if timer:
            timer.cancel()
    return hist + rest


def frontier_weights(ks, res, best_c, unsat_k, rest=0.02):
    """The refinement ends when max UNSAT k + 1 == min SAT count, so two open bounds decide it: the highest (a
    model there lowers the ceiling - the reference's own next iteration, main.rs:346) and the lowest (a refutation
    there raises the floor).  They share the fleet; the bounds in between keep a few workers each (their learnt
    clauses travel through the exchange either way)."""
    open_ks = [k for k, r in zip(ks, res) if r == SolverResult.Interrupted and (best_c is None or k < best_c) and k > unsat_k]
    hi, lo = (max(open_ks), min(open_ks)) if open_ks else (None, None)
    return [1.0 if k in (hi, lo) else rest for k in ks]


def _sweep_below(grid, encoding, k0, make_solver, out, on_interrupter, time_limit, specialize_after=None, stall=0.4):
    cnf = encoding.with_limits_into_cnf(PlatformLimits({(1, 1): k0}), sweep=True)
    ks = list(range(k0, -1, -1))
    sets = [([-int(cnf.card_outputs[k])] if k < len(cnf.card_outputs) else []) for k in ks]
    solver = (make_solver or (lambda: Mi355Sat()))()
    solver.add_cnf(cnf.lits, cnf.offsets)
    solver.reserve(cnf.n_vars)
    flag = []
    inner = solver.interrupter()

    class _SweepInterrupter:   # the batch is driven from here, so the loop has to see the interrupt too
        def interrupt(self):
            flag.append(1)
            inner.interrupt()

    if on_interrupter:
        on_interrupter(_SweepInterrupter())
    t0 = time.perf_counter()
    solver.sweep_begin(sets)
    best_c, best_i, unsat_k, looked = None, None, -1, set()
    interrupted = specialize = False
    t_progress = t0
    while True:
        res, _ = solver.sweep_step()
        for i, r in enumerate(res):
            if r == SolverResult.Unsat:
                unsat_k = max(unsat_k, ks[i])
            elif r == SolverResult.Sat and i not in looked:
                looked.add(i)
                c = PlatformLayout.from_assignment(solver.sweep_solution_of(i, encoding.n_vars), encoding).platform_count()
                if best_c is None or c < best_c:
                    best_c, best_i = c, i
                    t_progress = time.perf_counter()
        if best_c is not None and (unsat_k + 1 >= best_c or best_c == 0):
            break
        if all(r != SolverResult.Interrupted for r in res):
            break
        if flag or (time_limit is not None and time.perf_counter() - t0 > time_limit):
            interrupted = True
            break
        # the batch is the way DOWN: it ends once it has stopped finding better layouts (no new best count for
        # `stall` seconds), at the latest after three times `specialize_after`; `specialize_after` itself is how long
        # it runs at least when it is still walking down
        now = time.perf_counter()
        if specialize_after is not None and ((now - t0 > specialize_after and now - t_progress > stall) or now - t0 > 3 * specialize_after
                                             or (now - t0 > stall and now - t_progress > 2 * stall)):
            specialize = True
            break
        solver.sweep_drop([i for i, k in enumerate(ks) if res[i] == SolverResult.Interrupted and
                           ((best_c is not None and k >= best_c) or k < unsat_k)])
        solver.sweep_set_weights(frontier_weights(ks, res, best_c, unsat_k))
    dt = time.perf_counter() - t0
    layout = None
    if best_i is not None:
        layout = PlatformLayout.from_assignment(solver.sweep_solution_of(best_i, encoding.n_vars), encoding)
    solver.sweep_end()
    stats = solver.stats()
    solver.close()
    history = []
    if layout is not None:
        rec = {"k": k0, "result": SolverResult.Sat, "count": best_c, "valid": layout.validate(grid).is_valid(),
               "seconds": dt, "stats": stats, "layout": layout}
        history.append(rec)
        if best_c == 0:
            out("Found a solution with no platforms - aborting")
            return history
        out(f"Solution found ({best_c} platforms total)")
        for (w, h), n in sorted(layout.platform_stats().items()):
            out(f"{w}x{h}: {n}")
        out("Solution validation OK" if rec["valid"] else "Solution validation FAILED")
    if specialize:      # the caller finishes with the sequential loop from the best count (or from k0 if none yet)
        history.append({"result": "specialize"})
    elif interrupted:
        history.append({"k": (best_c - 1) if best_c else k0, "result": SolverResult.Interrupted, "count": None, "valid": None,
                        "seconds": dt, "stats": stats})
        out("Solver interrupted")
    elif best_c is None or unsat_k + 1 >= best_c:
        history.append({"k": unsat_k if best_c is None else best_c - 1, "result": SolverResult.Unsat, "count": None,
                        "valid": None, "seconds": dt, "stats": stats})
        out("No solution found for the current constraints")
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
