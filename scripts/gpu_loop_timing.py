"""Wall-clock of the drop-in refinement loop (tbs_cli = csrc/host/solver_loop.cpp over the C ABI):
sequential (the reference's loop shape) vs --sweep (all bounds as one batch).  GPU box only."""
import os
import subprocess
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cli = os.path.join(root, "timberborn_support_solver_amd", "tbs_cli")
size = sys.argv[1] if len(sys.argv) > 1 else "24"
for rep in range(3):
    for mode in ([], ["--sweep"], ["--sweep", "--workers", "1024"], ["--sweep", "--workers", "4096"]):
        t0 = time.perf_counter()
        try:
            out = subprocess.run([cli, "rect", size, size, f"-l1:{size}"] + mode, capture_output=True, text=True, timeout=150).stdout
        except subprocess.TimeoutExpired:
            out = ""
        dt = time.perf_counter() - t0
        found = [l for l in out.splitlines() if l.startswith("Solution found")]
        print(f"rect {size} {' '.join(mode) or 'sequential'}: {dt:.2f} s, finished={'No solution found' in out}, last={found[-1] if found else None}", flush=True)
