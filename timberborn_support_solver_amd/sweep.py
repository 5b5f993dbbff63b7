"""Sharding of the decreasing-k sweep over GPUs (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference solves one fresh CNF per k, sequentially (solver_loop, crates/repl/src/main.rs:290-346);
the k's are independent and at-most-k is monotone, so: SAT with count c => every k >= c is SAT; UNSAT at
k => every k' <= k is UNSAT.  Ranks take k = k_hi - rank - i*world (speculative descending sweep) and
exchange only the cut: one all-reduce of (min SAT count, max UNSAT k) and a broadcast of the winning
model.  There is no data-path collective: payloads are a few bytes plus one model.

`solver_loop_sweep_sharded` is the loop itself: every rank uploads the same CNF (one totalizer for the
start bound, each lower bound an assumption), keeps its own shard of bounds open, and after every
`exchange_every` slices all ranks agree on the cut; bounds the cut implies are withdrawn everywhere, and a
rank whose shard is decided takes up the bounds still open on other ranks (its workers join those searches
with their own decision orders)."""
import time

import numpy as np
import torch
import torch.distributed as dist

from .encoder import PlatformLayout, PlatformLimits
from .solver import Mi355Sat, SolverResult

_BIG = 1 << 40


def shard_bounds(k_hi, k_lo, rank, world):
    return list(range(k_hi - rank, k_lo - 1, -world))


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def exchange_cut(local, model_of, n_model, device):
    """local: {k: ("sat", count) | ("unsat", None) | ("open", None)} for this rank's k's.
    Returns {"min_sat", "max_unsat", "model" (tensor of the best SAT model, from its owner), "done"}."""
    sat = [c for k, (r, c) in local.items() if r == "sat"]
    unsat = [k for k, (r, _) in local.items() if r == "unsat"]
    my_min = min(sat, default=_BIG)
    t = torch.tensor([-my_min, max(unsat, default=-1)], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    min_sat, max_unsat = -int(t[0]), int(t[1])
    model = torch.zeros(n_model, dtype=torch.float32, device=device)
    if min_sat < _BIG:
        rank = dist.get_rank() if dist.is_initialized() else 0
        owner = torch.tensor([rank if my_min == min_sat else -1], dtype=torch.int64, device=device)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(owner, op=dist.ReduceOp.MAX)
        if int(owner[0]) == rank:
            k_best = min(k for k, (r, c) in local.items() if r == "sat" and c == min_sat)
            m = model_of(k_best)
            model.copy_(torch.as_tensor(m, dtype=torch.float32))
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(model, src=int(owner[0]))
    return {"min_sat": None if min_sat >= _BIG else min_sat, "max_unsat": None if max_unsat < 0 else max_unsat,
            "model": model, "done": min_sat < _BIG and max_unsat + 1 >= min_sat}


def exchange_ring(solver, device, max_words=1 << 18):
    """Cross-GPU clause exchange: every rank hands out the records its workers passed on since the last round
    (`Mi355Sat.share_export`: [lbd, literals, 0] in the caller's variables) and attaches everybody else's
    (`share_import`) - one all-reduce for the longest buffer, one all-gather of the padded buffers (a few hundred KB per
    round over xGMI).  All ranks must search the same formula.  Returns (records sent, records taken)."""
    rank, world = _world()
    if world == 1:
        return 0, 0
    buf, n_sent = solver.share_export(max_words)
    longest = torch.tensor([len(buf)], dtype=torch.int64, device=device)
    dist.all_reduce(longest, op=dist.ReduceOp.MAX)
    longest = int(longest[0])
    if longest == 0:
        return 0, 0
    mine = torch.zeros(longest + 1, dtype=torch.int32, device=device)
    mine[0] = len(buf)
    if len(buf):
        mine[1:1 + len(buf)] = torch.as_tensor(buf, dtype=torch.int32)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    taken = 0
    for r, t in enumerate(parts):
        if r != rank and int(t[0]) > 0:
            taken += solver.share_import(t[1:1 + int(t[0])].cpu().numpy())
    return n_sent, taken


def _bcast_first(rec, n_model, device):
    """Rank 0 made the first (loose) iteration as the reference does; everybody needs its count and model."""
    rank, world = _world()
    head = torch.tensor([rec["result"].value if rank == 0 else 0, (rec["count"] or 0) if rank == 0 else 0],
                        dtype=torch.int64, device=device)
    model = torch.zeros(n_model, dtype=torch.float32, device=device)
    if rank == 0 and rec.get("model") is not None:
        model.copy_(torch.as_tensor(np.asarray(rec["model"], dtype=np.float32)))
    if world > 1:
        dist.broadcast(head, src=0)
        dist.broadcast(model, src=0)
    return SolverResult(int(head[0])), int(head[1]), model


def solver_loop_sweep_sharded(grid, encoding, limits, make_solver=None, out=print, time_limit=None, exchange_every=1,
                              device="cpu", stats_out=None, specialize_after=2.0, ring=True):
    """The decreasing-k refinement (crates/repl/src/main.rs:280-366, `-l1:K` form) sharded over the ranks of
    the default process group.  Every rank returns the same history (records shaped like
    solver_loop's: the start bound, the best layout, the refuted bound); rank 0 prints the reference's
    messages.  `device` is where the two tiny collectives live ("cuda" under nccl/RCCL, "cpu" under gloo).
    stats_out (dict) receives this rank's solver counters and the time to the cut.

    As in loop.py::solver_loop_sweep the batch is only the way DOWN: once it has run `specialize_after` seconds it
    ends and the ranks pose the next bound (best count - 1) the way the reference
    does, with its own CNF - every rank the same bound with its own seed (replicas: a single refutation does not
    shard), polling each other every slice; the first verdict is everybody's."""
    if set(limits.card_limits) != {(1, 1)} or limits.weights or limits.weight_limit is not None:
        raise ValueError("solver_loop_sweep_sharded handles a single 1x1 cardinality limit")
    rank, world = _world()
    say = out if rank == 0 else (lambda line: None)
    make_solver = make_solver or (lambda: Mi355Sat())
    n_model = encoding.n_vars
    t_start = time.perf_counter()
    # ---- first iteration exactly as the reference makes it (loose start bound), on rank 0
    first = {"k": limits.card_limits[(1, 1)], "result": SolverResult.Interrupted, "count": None, "valid": None}
    if rank == 0:
        cnf0 = encoding.with_limits_into_cnf(limits)
        s0 = make_solver()
        s0.add_cnf(cnf0.lits, cnf0.offsets)
        s0.reserve(cnf0.n_vars)
        first["result"] = s0.solve()
        if first["result"] == SolverResult.Sat:
            first["model"] = s0.full_solution(n_model)
            first["count"] = PlatformLayout.from_assignment(first["model"], encoding).platform_count()
        s0.close()
    res0, count0, model0 = _bcast_first(first, n_model, device)
    first["result"], first["count"] = res0, (count0 if res0 == SolverResult.Sat else None)
    first["model"] = model0
    history = [first]
    if res0 != SolverResult.Sat:
        say("No solution found for the current constraints" if res0 == SolverResult.Unsat else "Solver interrupted")
        return history
    best_c, best_model = count0, model0.clone()
    lay0 = PlatformLayout.from_assignment(best_model.cpu().numpy().astype(np.int8), encoding)
    first["valid"] = lay0.validate(grid).is_valid()
    _say_layout(say, lay0, count0, first["valid"])
    if count0 == 0:
        say("Found a solution with no platforms - aborting")
        return history
    # ---- the batch: bounds count0-1 .. 0, this rank's shard open, the rest withdrawn until needed
    k_hi = count0 - 1
    cnf = encoding.with_limits_into_cnf(PlatformLimits({(1, 1): k_hi}), sweep=True)
    ks = list(range(k_hi, -1, -1))
    idx = {k: i for i, k in enumerate(ks)}
    sets = [([-int(cnf.card_outputs[k])] if k < len(cnf.card_outputs) else []) for k in ks]
    mine = set(shard_bounds(k_hi, 0, rank, world))
    solver = make_solver()
    solver.add_cnf(cnf.lits, cnf.offsets)
    solver.reserve(cnf.n_vars)
    solver.sweep_begin(sets)
    active = set(mine)                      # bounds this rank has open
    solver.sweep_drop([idx[k] for k in ks if k not in active])
    local, looked, unsat_k = {k: ("open", None) for k in ks}, set(), -1
    t0 = time.perf_counter()
    step, interrupted, cut, specialize = 0, False, None, False
    ring_stats = [0, 0]
    while True:
        res, _ = solver.sweep_step()
        step += 1
        for i, r in enumerate(res):
            k = ks[i]
            if r == SolverResult.Unsat:
                local[k] = ("unsat", None)
            elif r == SolverResult.Sat and i not in looked:
                looked.add(i)
                c = PlatformLayout.from_assignment(solver.sweep_solution_of(i, n_model), encoding).platform_count()
                local[k] = ("sat", c)
        if step % exchange_every:
            continue
        cut = exchange_cut(local, lambda k: solver.sweep_solution_of(idx[k], n_model), n_model, device)
        if ring:       # the ranks share one formula (one totalizer, the bounds are assumptions): their learnt clauses too
            sent, taken = exchange_ring(solver, device)
            ring_stats[0] += sent
            ring_stats[1] += taken
        if cut["min_sat"] is not None and cut["min_sat"] < best_c:
            best_c, best_model = cut["min_sat"], cut["model"].clone()
        if cut["max_unsat"] is not None:
            unsat_k = max(unsat_k, cut["max_unsat"])
        if unsat_k + 1 >= best_c or best_c == 0:
            break
        # the time limit is part of the agreement too: all ranks leave in the same round
        late = torch.tensor([1 if (time_limit is not None and time.perf_counter() - t0 > time_limit) else 0],
                            dtype=torch.int64, device=device)
        if world > 1:
            dist.all_reduce(late, op=dist.ReduceOp.MAX)
        if int(late[0]):
            interrupted = True
            break
        # what the global cut leaves open: unsat_k < k < best_c.  Withdraw the rest; a rank whose shard has
        # nothing open any more takes up everything that is still open anywhere.
        open_ks = [k for k in ks if unsat_k < k < best_c]
        if specialize_after is not None:      # (the cut is agreed, the clocks are not: agree on leaving too)
            go = torch.tensor([1 if time.perf_counter() - t0 > specialize_after else 0], dtype=torch.int64, device=device)
            if world > 1:
                dist.all_reduce(go, op=dist.ReduceOp.MAX)
            if int(go[0]):
                specialize = True
                break
        want = [k for k in open_ks if k in mine and local[k][0] == "open"] or [k for k in open_ks if local[k][0] == "open"]
        want = set(want)
        solver.sweep_drop([idx[k] for k in active - want if local[k][0] == "open"])
        solver.sweep_reopen([idx[k] for k in want - active])
        active = want
        if active:   # within what this rank has open: the two bounds that decide the loop share its workers
            hi, lo = max(active), min(active)
            solver.sweep_set_weights([1.0 if k in (hi, lo) else 0.02 for k in ks])
    dt = time.perf_counter() - t0
    solver.sweep_end()
    stats = solver.stats()
    solver.close()
    if stats_out is not None:
        stats_out.update(stats=stats, seconds_to_cut=dt, seconds_total=time.perf_counter() - t_start, steps=step,
                         rank=rank, world=world, ring_exported=ring_stats[0], ring_imported=ring_stats[1])
    if best_c < count0:
        lay = PlatformLayout.from_assignment(best_model.cpu().numpy().astype(np.int8), encoding)
        rec = {"k": k_hi, "result": SolverResult.Sat, "count": best_c, "valid": lay.validate(grid).is_valid(),
               "seconds": dt, "layout": lay, "model": best_model}
        history.append(rec)
        _say_layout(say, lay, best_c, rec["valid"])
        if best_c == 0:
            say("Found a solution with no platforms - aborting")
            return history
    if specialize:
        left = None if time_limit is None else max(0.0, time_limit - dt)
        return history + _replica_tail(grid, encoding, best_c, make_solver, say, left, device, n_model, stats_out, ring)
    if interrupted:
        history.append({"k": best_c - 1, "result": SolverResult.Interrupted, "count": None, "valid": None, "seconds": dt})
        say("Solver interrupted")
    else:
        history.append({"k": best_c - 1, "result": SolverResult.Unsat, "count": None, "valid": None, "seconds": dt})
        say("No solution found for the current constraints")
    return history


def _replica_tail(grid, encoding, best_c, make_solver, say, time_limit, device, n_model, stats_out, ring=True):
    """The reference's sequential loop from `best_c` on, every rank a replica with its own seed: each bound gets its own
    CNF (main.rs:292-293) and is stepped slice by slice; after every slice one all-reduce tells whether any rank has
    the verdict, a model comes from its owner - and (ring) the ranks hand each other the clauses their workers passed
    on in that slice, so that the replicas share what they learn instead of only racing (on one GPU the exchange is
    what makes 1024 workers more than a portfolio; a plain portfolio gains nothing from more workers)."""
    rank, world = _world()
    history, t0 = [], time.perf_counter()
    while True:
        k = best_c - 1
        cnf = encoding.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
        solver = make_solver()
        solver.add_cnf(cnf.lits, cnf.offsets)
        solver.reserve(cnf.n_vars)
        solver.sweep_begin([[]])
        tk = time.perf_counter()
        verdict = 0
        while True:
            res, _ = solver.sweep_step()
            late = 1 if (time_limit is not None and time.perf_counter() - t0 > time_limit) else 0
            t = torch.tensor([res[0].value, late], dtype=torch.int64, device=device)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            verdict = int(t[0])
            if verdict or int(t[1]):
                break
            if ring:
                sent, taken = exchange_ring(solver, device)
                if stats_out is not None:
                    stats_out["ring_exported"] = stats_out.get("ring_exported", 0) + sent
                    stats_out["ring_imported"] = stats_out.get("ring_imported", 0) + taken
        model = torch.zeros(n_model, dtype=torch.float32, device=device)
        if verdict == SolverResult.Sat.value:
            owner = torch.tensor([rank if res[0] == SolverResult.Sat else -1], dtype=torch.int64, device=device)
            if world > 1:
                dist.all_reduce(owner, op=dist.ReduceOp.MAX)
            if int(owner[0]) == rank:
                model.copy_(torch.as_tensor(np.asarray(solver.sweep_solution_of(0, n_model), dtype=np.float32)))
            if world > 1:
                dist.broadcast(model, src=int(owner[0]))
        solver.sweep_end()
        if stats_out is not None:
            stats_out.setdefault("tail_conflicts", 0)
            stats_out["tail_conflicts"] += solver.stats()["conflicts"]
        solver.close()
        dt = time.perf_counter() - tk
        if verdict == SolverResult.Unsat.value:
            history.append({"k": k, "result": SolverResult.Unsat, "count": None, "valid": None, "seconds": dt})
            say("No solution found for the current constraints")
            return history
        if verdict == 0:
            history.append({"k": k, "result": SolverResult.Interrupted, "count": None, "valid": None, "seconds": dt})
            say("Solver interrupted")
            return history
        lay = PlatformLayout.from_assignment(model.cpu().numpy().astype(np.int8), encoding)
        best_c = lay.platform_count()
        rec = {"k": k, "result": SolverResult.Sat, "count": best_c, "valid": lay.validate(grid).is_valid(), "seconds": dt,
               "layout": lay, "model": model}
        history.append(rec)
        if best_c == 0:
            say("Found a solution with no platforms - aborting")
            return history
        _say_layout(say, lay, best_c, rec["valid"])


def _say_layout(say, layout, count, valid):
    say(f"Solution found ({count} platforms total)")
    for (w, h), n in sorted(layout.platform_stats().items()):
        say(f"{w}x{h}: {n}")
    say("Solution validation OK" if valid else "Solution validation FAILED")
