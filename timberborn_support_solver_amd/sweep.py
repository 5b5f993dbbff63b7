"""Sharding of the decreasing-k sweep over GPUs (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference solves one fresh CNF per k, sequentially (solver_loop, crates/repl/src/main.rs:290-346);
the k's are independent and at-most-k is monotone, so: SAT with count c => every k >= c is SAT; UNSAT at
k => every k' <= k is UNSAT.  Ranks take k = k_hi - rank - i*world (speculative descending sweep) and
exchange only the cut: one all-reduce of (min SAT count, max UNSAT k) and a broadcast of the winning
model.  There is no data-path collective: payloads are a few bytes plus one model."""
import torch
import torch.distributed as dist

_BIG = 1 << 40


def shard_bounds(k_hi, k_lo, rank, world):
    return list(range(k_hi - rank, k_lo - 1, -world))


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
