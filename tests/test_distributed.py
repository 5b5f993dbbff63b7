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
