"""The N>1 path on CPU: world_size-2 gloo run of the cut exchange the sharded sweep uses
(min SAT k / max UNSAT k all-reduce; model broadcast from the winner)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from timberborn_support_solver_amd.sweep import exchange_cut, shard_bounds
    ks = shard_bounds(k_hi=20, k_lo=9, rank=rank, world=world)
    # pretend verdicts: k >= 14 SAT (count = k), k <= 13 UNSAT
    local = {k: ("sat", k) if k >= 14 else ("unsat", None) for k in ks}
    model = torch.full((8,), float(rank + 1)) if 14 in local else None
    cut = exchange_cut(local, model_of=lambda k: model, n_model=8, device="cpu")
    q.put((rank, ks, cut["min_sat"], cut["max_unsat"], cut["model"].tolist(), cut["done"]))
    dist.destroy_process_group()


def test_two_rank_cut_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ks0, s0, u0, m0, d0), (r1, ks1, s1, u1, m1, d1) = out
    assert ks0 == [20, 18, 16, 14, 12, 10] and ks1 == [19, 17, 15, 13, 11, 9]      # k_hi - rank - i*world
    assert s0 == s1 == 14 and u0 == u1 == 13 and d0 and d1                          # cut closed: 13 + 1 == 14
    assert m0 == m1 == [1.0] * 8                                                   # broadcast from the rank owning k=14


def _sharded_worker(rank, world, port, q, terrain, pset, k0, spec=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import check_sat_answer, emu_lib, make_grid, platform_defs
    from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits
    from timberborn_support_solver_amd.sweep import solver_loop_sweep_sharded
    grid = make_grid(terrain)
    enc = Encoding.encode(platform_defs(pset), grid)
    lines, st = [], {}
    # real solves: the product's kernels in the wavefront-emulator build, a few workers per rank
    hist = solver_loop_sweep_sharded(grid, enc, PlatformLimits({(1, 1): k0}), out=lines.append, time_limit=600,
                                     make_solver=lambda: Mi355Sat(_lib_override=emu_lib(), workers=6, slice_conflicts=30, simp=-1,
                                                                  seed=100 + rank),
                                     stats_out=st, specialize_after=spec)
    best = [h for h in hist if h["result"].name == "Sat"][-1]
    model = best["model"]
    model = model.tolist() if hasattr(model, "tolist") else list(model)
    q.put((rank, [(h["k"], h["result"].name, h["count"], h["valid"]) for h in hist], [int(x) for x in model], lines,
           st.get("stats", {}).get("conflicts", 0)))
    dist.destroy_process_group()


@pytest.mark.parametrize("terrain,pset,k0,kstar,spec", [("rect8x8", "1x1", 8, 4, None), ("ex1", "1x1", 10, 3, None),
                                                        ("ex1", "1x1", 10, 3, 0.0)])
def test_two_rank_sharded_sweep_with_real_solves_gloo(terrain, pset, k0, kstar, spec):
    """SURVEY 8e end to end on CPU: two ranks over gloo, each driving the product's search kernels (emulator
    build) on its shard of bounds k = k_hi - rank - i*world; the cut all-reduce + model broadcast make both
    ranks return the same history, and the optimum is the golden one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    # spec = 0.0: the batch hands over to the replica tail (every rank the same bound with its own CNF and seed)
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, terrain, pset, k0, spec)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, h0, m0, lines0, c0), (_, h1, m1, lines1, c1) = out
    assert h0 == h1 and m0 == m1                                    # both ranks agree on cut and model
    assert h0[-1] == (kstar - 1, "Unsat", None, None)
    sat = [h for h in h0 if h[1] == "Sat"]
    assert sat[-1][2] == kstar and all(h[3] for h in sat)
    assert lines0[-1] == "No solution found for the current constraints" and lines1 == []   # rank 0 speaks
    assert f"Solution found ({kstar} platforms total)" in lines0
    assert c0 > 0 and c1 > 0                                        # both ranks really searched (the batch part)
    # the agreed model is a model of the CNF at its own count (checked with the oracle)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from helpers import check_sat_answer, make_grid, platform_defs
    from timberborn_support_solver_amd import Encoding, PlatformLayout, PlatformLimits
    grid = make_grid(terrain)
    enc = Encoding.encode(platform_defs(pset), grid)
    lay = PlatformLayout.from_assignment(np.asarray(m0, dtype=np.int8), enc)
    assert lay.platform_count() == kstar and lay.validate(grid).is_valid()


def _ring_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import assert_ring_records_are_implied, emu_lib, make_grid, platform_defs
    from timberborn_support_solver_amd import Encoding, Mi355Sat, PlatformLimits
    from timberborn_support_solver_amd.sweep import exchange_ring
    grid = make_grid("rect8x8")
    enc = Encoding.encode(platform_defs("1x1"), grid)
    # the formula (totalizer for k = 6) is satisfiable, so "implied by the formula" says something; the bound the replicas
    # work on (k = 3, an assumption) is not: k* = 4 (golden)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): 6}), sweep=True)
    s = Mi355Sat(_lib_override=emu_lib(), workers=6, slice_conflicts=10, simp=-1, seed=500 + rank, share_lbd=8)
    s.add_cnf(cnf.lits, cnf.offsets)
    s.reserve(cnf.n_vars)
    s.sweep_begin([[-int(cnf.card_outputs[3])]])
    sent = taken = rounds = 0
    for _ in range(40):
        res, _ = s.sweep_step()
        a, b = exchange_ring(s, "cpu")
        sent, taken, rounds = sent + a, taken + b, rounds + 1
        st = s.stats()                                                  # (workers attach foreign records a slice after they arrive)
        got = st["shared_imported"] + st["shared_imported_units"] > 0
        done = torch.tensor([1 if (res[0].value or (sent and taken and got and rounds >= 3)) else 0], dtype=torch.int64)
        dist.all_reduce(done, op=dist.ReduceOp.MIN)                    # leave together
        if int(done[0]):
            break
    n_ring = assert_ring_records_are_implied(s, cnf)                   # own AND foreign records follow from the formula
    s.sweep_end()
    st = s.stats()
    s.close()
    q.put((rank, res[0].name, sent, taken, n_ring, st["shared_imported"] + st["shared_imported_units"]))
    dist.destroy_process_group()


def test_two_rank_clause_exchange_between_handles_gloo():
    """The cross-GPU ring (mi355sat_share_export / _import under sweep.exchange_ring): two replicas of one refutation
    (rect 8x8, 1x1 supports, the bound k = 3), each a handle of its own with its own seed, hand each other what their
    workers passed on; every record in either ring - foreign ones included - is a consequence of the formula (oracle),
    both ranks took records of the other and their workers attached some.  (The verdict through the replica tail with the
    ring on is the `spec = 0.0` case of the sharded-sweep test above.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, verdict, sent, taken, n_ring, imported in out:
        assert verdict in ("Interrupted", "Unsat")
        assert sent > 0 and taken > 0, (rank, sent, taken)
        assert n_ring >= taken and imported > 0
    assert out[0][2] >= out[1][3] and out[1][2] >= out[0][3]          # nobody took more than the other sent
