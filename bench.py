#!/usr/bin/env python3
"""bench.py — literal-propagations/s of the MI355X-native SAT solve loop on the
BASELINE.json headline workload (rect 64x64, decreasing-k sweep at the
first-UNSAT bound, default platform set).

One *step* = one slice of the search kernel over this GPU's batch of (k, seed)
instances: every worker (one wavefront) advances its own CDCL search for
`--slice-ms` milliseconds of device time (time-bounded so that all workers stop
together; a conflict-bounded slice is dominated by its slowest worker).  Inputs (clause database, worker slabs) are resident in HBM
before the timed region.  Multi-GPU: one process per GPU; the (k, seed) instances
are independent, so ranks shard the seeds with no data-path collective and only
exchange the SAT/UNSAT cut (min SAT count, max UNSAT k: one tiny all-reduce per
step over RCCL) — weak scaling.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes
(12*n_deq + 9*n_watch + 5*n_cl_lit + 8*n_move + 13*n_enq) divided by
the search kernel's device time measured with HIP events on its own stream.
`cpu_baseline` = the oracle's single-thread CDCL restatement ("port": the real
rustsat-glucose cannot be built here) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _cpu_leg(path, assumps, budget):
    """One process of the all-cores CPU baseline (bench.py --cpu-leg ...): the oracle's CDCL on one at-most-k instance
    for a conflict budget; prints "propagations conflicts"."""
    import numpy as np
    from oracle import oracle as ora
    z = np.load(path)
    o = ora.OracleSolver()
    o.add_cnf(z["lits"], z["offs"])
    o.solve([int(x) for x in assumps.split(",") if x], conflict_budget=int(budget))
    st = o.stats()
    print(st["propagations"], st["conflicts"])


def measured_copy_gbs(torch, device_index):
    """Streaming ceiling measured on this box: device-to-device copy of 2 GiB (read + write = 4 GiB of traffic)."""
    n = 1 << 29
    a = torch.empty(n, dtype=torch.int32, device=f"cuda:{device_index}")
    b = torch.empty_like(a)
    a.fill_(1)
    best = 0.0
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    torch.cuda.empty_cache()
    return best


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1) and relay rank 0's line.  This parent never touches HIP or torch.cuda."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def _build_identity():
    """What a profile must match to be evidence about THIS build: the commit and the kernel sources' hash."""
    import hashlib
    import subprocess
    h = hashlib.sha256()
    for f in ("timberborn_support_solver_amd/csrc/device/kernels.hip.h", "timberborn_support_solver_amd/csrc/device/layout.h",
              "timberborn_support_solver_amd/csrc/mi355sat.hip"):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:
        head = ""
    return {"git_head": head or None, "kernel_source_sha16": h.hexdigest()[:16]}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-leg":
        return _cpu_leg(sys.argv[2], sys.argv[3], sys.argv[4])
    if len(sys.argv) >= 2 and sys.argv[1] == "--build-identity":
        print(json.dumps(_build_identity()))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=64, help="rect SIZE x SIZE")
    ap.add_argument("--k-lo", type=int, default=44)
    ap.add_argument("--k-hi", type=int, default=51)
    ap.add_argument("--workers", type=int, default=4096, help="wavefront workers per GPU (16 per CU)")
    ap.add_argument("--slice-ms", type=int, default=250, help="device time of one step (one kernel launch)")
    ap.add_argument("--cpu-conflicts", type=int, default=200000, help="conflict budget of the CPU baseline sample (~10-15 s)")
    ap.add_argument("--first-unsat-sizes", default="24,26,28", help="rect sizes of the wall-clock-to-first-UNSAT ladders, each run by "
                    "the GPU loop and by the CPU restatement's loop in this same run ('' or 0 = skip); first_unsat_wall_clock_s is "
                    "the largest size's")
    ap.add_argument("--first-unsat-size", type=int, default=None, help="(old form) one size; 0 = skip")
    ap.add_argument("--first-unsat-limit", type=float, default=150.0, help="time limit per ladder and side, seconds")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="processes of the all-cores CPU leg (0 = the cores this process may use, at most 16)")
    ap.add_argument("--var-order", type=int, default=0, help="0 caller's numbering (default), 1 locality order (A/B)")
    ap.add_argument("--share", type=int, default=-1, help="learnt-clause exchange in the THROUGHPUT sweep: -1 off (default: the "
                    "exchange makes the trajectory, and with it the rate, depend on slice timing), 0 on.  The first-UNSAT line always "
                    "runs the product default (exchange on).")
    ap.add_argument("--platforms", default="default", choices=["default", "1x1"])
    args = ap.parse_args()
    if args.first_unsat_size is not None:
        args.first_unsat_sizes = str(args.first_unsat_size)
    fu_sizes = [int(x) for x in args.first_unsat_sizes.split(",") if x.strip() and int(x) > 0]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the solver has no CPU path)")
    # BENCH_BACKEND=gloo + BENCH_DEVICE=0 rehearse the N>1 path with several ranks on ONE GPU
    # (RCCL refuses two ranks per device); the driver's runs use nccl (= RCCL) with one rank per GPU.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    device_index = int(os.environ.get("BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)

    from timberborn_support_solver_amd import (PLATFORMS_DEFAULT, Encoding, Mi355Sat, PlatformLimits, SolverResult,
                                               WorldGrid, algorithmic_bytes)

    n = args.size
    grid = WorldGrid.rect(n, n)
    defs = PLATFORMS_DEFAULT if args.platforms == "default" else [(1, 1)]
    enc = Encoding.encode(defs, grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): args.k_hi}), sweep=True)
    outs = cnf.card_outputs
    ks = list(range(args.k_hi, args.k_lo - 1, -1))  # descending sweep
    # bound k < k_hi is the assumption "NOT at-least-(k+1)"; k_hi itself is already a unit in the CNF
    assumption_sets = [([-int(outs[k])] if k < args.k_hi else []) for k in ks]
    workers = max(len(ks), args.workers // len(ks) * len(ks))

    lib_override = None
    if os.environ.get("BENCH_LIB"):   # A/B of diagnostic builds of the same library (e.g. -DMS_SPECULATE=0)
        import ctypes
        lib_override = ctypes.CDLL(os.path.abspath(os.environ["BENCH_LIB"]))
    solver = Mi355Sat(device=device_index, workers=workers, slice_ms=args.slice_ms, seed=1000 + rank, _lib_override=lib_override,
                      var_order=args.var_order, share=args.share,
                      ramp=-1)   # throughput of the whole fleet is what is measured: no ramp-up phase
    solver.add_cnf(cnf.lits, cnf.offsets)
    solver.reserve(cnf.n_vars)
    solver.sweep_begin(assumption_sets)  # upload + replicate: everything resident in HBM from here on

    def exchange_cut(results):
        """The only cross-GPU traffic of the path: (min SAT k, max UNSAT k) over all ranks."""
        sat_k = min([k for k, r in zip(ks, results) if r == SolverResult.Sat], default=1 << 30)
        unsat_k = max([k for k, r in zip(ks, results) if r == SolverResult.Unsat], default=-1)
        if dist is not None:
            t = torch.tensor([-sat_k, unsat_k], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sat_k, unsat_k = -int(t[0]), int(t[1])
        return sat_k, unsat_k

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res, _ = solver.sweep_step()
        exchange_cut(res)
    st0 = solver.stats()
    barrier()
    t0 = time.perf_counter()
    decided = 0
    marks = [(t0, st0["propagations"])]          # per-step stamps: the spread of the rate inside the timed region
    for _ in range(args.steps):
        res, decided = solver.sweep_step()
        cut = exchange_cut(res)
        marks.append((time.perf_counter(), solver.stats()["propagations"]))
    barrier()
    dt = time.perf_counter() - t0
    st1 = solver.stats()
    n_win = 3 if args.steps >= 3 else 1
    edges = [round(i * args.steps / n_win) for i in range(n_win + 1)]
    win = [(marks[b][1] - marks[a][1]) / max(marks[b][0] - marks[a][0], 1e-9) for a, b in zip(edges[:-1], edges[1:])]
    per_step = [round((marks[i + 1][1] - marks[i][1]) / max(marks[i + 1][0] - marks[i][0], 1e-9) / 1e9, 3) for i in range(args.steps)]

    d = {k: st1[k] - st0[k] for k in ("propagations", "conflicts", "decisions", "n_deq", "n_watch", "n_cl_lit",
                                      "n_move", "n_enq", "kernel_seconds", "kernel_launches")}
    props = d["propagations"]
    alg_bytes = algorithmic_bytes(d)
    tot = [float(props), dt, float(d["conflicts"])]
    if dist is not None:
        t = torch.tensor([float(props), float(d["conflicts"])], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tm = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tot = [float(t[0]), float(tm[0]), float(t[1])]
    total_props, max_dt, total_confl = tot

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"timed region done: {total_props / max_dt:.3e} propagations/s")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:   # reported baseline: rank 0 at N=1 only
        import numpy as np
        from oracle import oracle as ora
        np_lits, np_offs = np.asarray(cnf.lits, dtype=np.int32), np.asarray(cnf.offsets, dtype=np.uint64)
        o = ora.OracleSolver()
        o.add_cnf(cnf.lits, cnf.offsets)
        k_cpu = ks[-1]
        tc = time.perf_counter()
        o.solve(assumption_sets[-1], conflict_budget=args.cpu_conflicts)
        dtc = time.perf_counter() - tc
        so = o.stats()
        cpu = {"value": so["propagations"] / max(dtc, 1e-9), "unit": "propagations/s", "cores": 1, "kind": "port",
               "sample": f"rect {n}x{n} {args.platforms}, at-most-{k_cpu}, first {so['conflicts']} conflicts "
                         f"({dtc:.1f} s) of the oracle's single-thread CDCL restatement (not rustsat-glucose)",
               "conflicts_per_s": so["conflicts"] / max(dtc, 1e-9)}
        note(f"cpu baseline, 1 core: {cpu['value']:.3e} propagations/s in {dtc:.1f} s")
        # all host cores, one independent (k) instance per core, same conflict budget each
        # (child processes of their own, started with subprocess: this process has initialised HIP, whose runtime threads do
        # not survive a fork)
        import subprocess
        import tempfile
        # the host cores this process may use: its affinity mask, at most 16 (a one-GPU box's CPU share; os.cpu_count()
        # is the whole node's)
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncores = min(avail, args.cpu_cores if args.cpu_cores > 0 else 16)
        note(f"cpu baseline, all cores: starting {ncores} processes")
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "cnf.npz")
            np.savez(path, lits=np_lits, offs=np_offs)
            tc = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-leg", path,
                                       ",".join(str(int(l)) for l in assumption_sets[i % len(ks)]), str(args.cpu_conflicts)],
                                      stdout=subprocess.PIPE, text=True) for i in range(ncores)]
            rs = [tuple(int(x) for x in p.communicate()[0].split()) for p in procs]
            dta = time.perf_counter() - tc
        cpu["all_cores"] = {"value": sum(r[0] for r in rs) / max(dta, 1e-9), "unit": "propagations/s", "cores": ncores,
                            "conflicts_per_s": sum(r[1] for r in rs) / max(dta, 1e-9),
                            "sample": f"{ncores} processes, one at-most-k instance each (k cycling {ks[0]}..{ks[-1]}), "
                                      f"{args.cpu_conflicts} conflicts each, {dta:.1f} s wall"}

    if cpu and "all_cores" in cpu:
        note(f"cpu baseline, {cpu['all_cores']['cores']} cores: {cpu['all_cores']['value']:.3e} propagations/s")
    solver.sweep_end()
    solver.close()

    # The same sweep in the PRODUCT'S default configuration (learnt-clause exchange on): not `value` - with the exchange
    # on, what a worker does in slice n depends on what was collected after slice n - 1, i.e. on slice timing - but
    # reported beside it so that the headline is not mistaken for the default configuration.
    exchange_window = None
    if rank == 0 and world == 1 and args.share < 0 and not args.no_cpu:
        s2 = Mi355Sat(device=device_index, workers=workers, slice_ms=args.slice_ms, seed=1000, _lib_override=lib_override,
                      var_order=args.var_order, share=0, ramp=-1)
        s2.add_cnf(cnf.lits, cnf.offsets)
        s2.reserve(cnf.n_vars)
        s2.sweep_begin(assumption_sets)
        for _ in range(min(3, args.warmup)):
            s2.sweep_step()
        a0, ta = s2.stats(), time.perf_counter()
        n2 = max(1, min(8, args.steps))
        for _ in range(n2):
            s2.sweep_step()
        torch.cuda.synchronize()
        dta = time.perf_counter() - ta
        a1 = s2.stats()
        exchange_window = {"slices": n2, "propagations_per_s": (a1["propagations"] - a0["propagations"]) / dta,
                           "conflicts_per_s": (a1["conflicts"] - a0["conflicts"]) / dta,
                           "clauses_attached_from_the_exchange": int(a1["shared_imported"] - a0["shared_imported"]),
                           "note": "product default (exchange on), same workload and fleet; not the headline value"}
        s2.sweep_end()
        s2.close()
        note(f"exchange on: {exchange_window['propagations_per_s']:.3e} propagations/s, {exchange_window['conflicts_per_s']:.3e} conflicts/s")

    # ---- wall-clock to first UNSAT on the largest rung that finishes in bench time (64x64 does not:
    # SURVEY §6).  GPU: the product's own loop, solver_loop_sweep (first bound alone as the reference makes it,
    # then every lower bound as one batch until max UNSAT k + 1 == min count).  CPU: the reference's
    # sequential loop (k := count - 1) on the oracle.
    first_unsat = None
    sharded = None
    if world > 1 and fu_sizes:
        # strong scaling of ONE ladder over the ranks: the bounds k = k_hi - rank - i*world are sharded, the cut
        # (min SAT count, max UNSAT k) and the best model are the only things exchanged; in the replica
        # tail (the last bounds, every rank the same bound) the ranks also pass on the clauses they learn
        from timberborn_support_solver_amd.sweep import solver_loop_sweep_sharded
        m = 26 if 26 in fu_sizes else fu_sizes[0]     # a ladder whose last bound takes tens of seconds: the replica tail and the ring matter
        g2 = WorldGrid.rect(m, m)
        e2 = Encoding.encode(defs, g2)
        st_sh = {}
        barrier()
        tg = time.perf_counter()
        hist = solver_loop_sweep_sharded(g2, e2, PlatformLimits({(1, 1): max(4, m * m // 24)}), out=lambda line: None,
                                         time_limit=args.first_unsat_limit,
                                         make_solver=lambda: Mi355Sat(device=device_index, seed=1000 + rank),
                                         device=coll_dev, stats_out=st_sh)
        barrier()
        gpu_s = time.perf_counter() - tg
        sat = [h for h in hist if h["result"] == SolverResult.Sat]
        ok = bool(sat) and hist[-1]["result"] == SolverResult.Unsat and all(h["valid"] for h in sat)
        sharded = {"instance": f"rect {m} {m} {args.platforms}, solver_loop_sweep_sharded over {world} ranks",
                   "optimum_k": sat[-1]["count"] if ok else None, "gpu_seconds": gpu_s if ok else None,
                   "seconds_to_cut_after_first_bound": st_sh.get("seconds_to_cut"),
                   "ring_records_from_other_ranks": st_sh.get("ring_imported")}
    if rank == 0 and world == 1 and fu_sizes and not args.no_cpu:
        from oracle import oracle as ora
        from timberborn_support_solver_amd import PlatformLayout, solver_loop_sweep
        first_unsat = []
        for m in fu_sizes:
            g2 = WorldGrid.rect(m, m)
            e2 = Encoding.encode(defs, g2)
            k0 = max(4, m * m // 24)
            tg = time.perf_counter()
            hist = solver_loop_sweep(g2, e2, PlatformLimits({(1, 1): k0}), out=lambda line: None, time_limit=args.first_unsat_limit,
                                     make_solver=lambda: Mi355Sat(device=device_index))
            gpu_s = time.perf_counter() - tg
            sat = [h for h in hist if h["result"] == SolverResult.Sat]
            kstar = sat[-1]["count"] if sat and hist[-1]["result"] == SolverResult.Unsat and all(h["valid"] for h in sat) else None
            gpu_confl = sum(int(h["stats"].get("conflicts", 0)) for h in hist if h.get("stats"))
            # CPU: decreasing-k loop, fresh solver per k (crates/repl/src/main.rs:290-346)
            tc = time.perf_counter()
            k, cpu_kstar, cpu_confl = k0, None, 0
            while time.perf_counter() - tc < args.first_unsat_limit:
                ck = e2.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
                o = ora.OracleSolver()
                o.add_cnf(ck.lits, ck.offsets)
                r = o.solve(conflict_budget=5_000_000)
                cpu_confl += int(o.stats()["conflicts"])
                if r == 20:
                    cpu_kstar = k + 1
                    break
                if r != 10:
                    break
                cnt = PlatformLayout.from_assignment(o.model(ck.n_vars)[:e2.n_vars], e2).platform_count()
                k = cnt - 1
            cpu_s = time.perf_counter() - tc
            note(f"first UNSAT rect {m}: gpu {gpu_s:.2f} s (k* {kstar}, {gpu_confl:.3g} conflicts), cpu {cpu_s:.2f} s (k* {cpu_kstar}, {cpu_confl:.3g} conflicts)")
            first_unsat.append({
                "instance": f"rect {m} {m} {args.platforms}, -l1:{k0} to the proven optimum", "optimum_k": kstar,
                "gpu_seconds": gpu_s if kstar is not None else None, "gpu_conflicts": gpu_confl,
                "gpu_loop": "solver_loop_sweep (first bound alone, the lower bounds as one batch on the way down, the last bounds with their own CNF), product defaults",
                "cpu_seconds": cpu_s if cpu_kstar is not None else None, "cpu_optimum_k": cpu_kstar, "cpu_conflicts": cpu_confl,
                "cpu_kind": "port (oracle CDCL restatement, 1 core, the reference's sequential loop)",
                "gpu_over_cpu": (gpu_s / cpu_s) if (kstar is not None and cpu_kstar is not None) else None,
                "time_limit_s": args.first_unsat_limit})
    if rank == 0:
        kern_s = d["kernel_seconds"]
        launches = max(1, d["kernel_launches"])
        achieved = alg_bytes / max(kern_s, 1e-9) / 1e9
        peak_measured = measured_copy_gbs(torch, device_index)
        # HBM bytes per launch of the dominant kernel from the PMC passes (they cannot run inside this process:
        # separate `rocprofv3 --pmc` runs of this same command; the committed summary is attached when this run is
        # the configuration it was taken on)
        traffic, traffic_src = None, None
        import glob
        tfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_traffic.json")))
        default_cfg = (n == 64 and args.k_lo == 44 and args.k_hi == 51 and workers == 4096 and args.slice_ms == 250 and
                       args.platforms == "default" and args.share == -1 and args.var_order == 0)
        ident = _build_identity()
        traffic_note = "no PMC summary under profiles/ was taken on this build (kernel sources differ): traffic left null"
        for tf in reversed(tfiles):     # attach a PMC summary only if it was measured on THESE kernel sources
            tj = json.load(open(tf))
            if default_cfg and tj.get("kernel_source_sha16") == ident["kernel_source_sha16"]:
                traffic, traffic_src = tj["hbm_bytes_per_launch"], "profiles/" + os.path.basename(tf)
                traffic_note = f"PMC passes of this command on the same kernel sources (git {tj.get('git_head')}), kernel {tj.get('kernel')}"
                break
        out = {
            "metric": "literal-propagations/sec + wall-clock to first UNSAT, 64x64 rect",
            "value": total_props / max_dt,
            "unit": "propagations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": max_dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"rect {n} {n} k-sweep at the first-UNSAT bound: at-most-k for k={args.k_hi}..{args.k_lo}, "
                                   f"{args.platforms} platforms, one totalizer CNF shared by all k",
                       "vars": int(cnf.n_vars), "clauses": int(cnf.n_clauses), "literals": int(len(cnf.lits)),
                       "workers_per_gpu": int(st1["workers"]), "instances_per_gpu": len(ks), "slice_ms": args.slice_ms,
                       "exchange": ("off" if args.share < 0 else "on") + " in the throughput sweep (value); on in the first-UNSAT line",
                       "parallelism": f"{world} GPU(s) x {workers} wavefront workers; value: seeds sharded over ranks (weak); "
                                      f"first_unsat_sharded: bounds k = k_hi - rank - i*{world} of one ladder sharded (strong)"},
            "windows": {"n": len(win), "values": win, "median": sorted(win)[len(win) // 2],
                        "spread": (max(win) - min(win)) / max(sorted(win)[len(win) // 2], 1e-9),
                        "per_step": per_step,
                       "note": "rank 0's rate in consecutive thirds of the timed region; per_step: the rate of every slice"},
            "conflicts_per_s": total_confl / max_dt,
            "exchange_on_window": exchange_window,
            "decided_instances": int(decided),
            "first_unsat_wall_clock_s": first_unsat[-1]["gpu_seconds"] if first_unsat else None,
            "first_unsat": first_unsat,
            "first_unsat_sharded": sharded,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "peak_measured": peak_measured,
                         "peak_measured_note": "device-to-device copy of 2 GiB on this box (read + write bytes)",
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
                         "build": ident,
                         "kernel": "ms_search_kernel", "kernel_ms_avg": kern_s / launches * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes / launches,
                         "bytes_per_propagation": alg_bytes / max(1, props)},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
