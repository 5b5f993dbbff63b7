"""the Rust `Solve` shim (rust/mi355sat) cannot be compiled here (no cargo/rustc), so its call and
threading contract is replayed against the C ABI by a C program (tests/abi_threads.c, gcc -pthread):
`new` on thread A, `add` per literal, `solve` on thread B (crates/repl/src/solver_runner.rs:12-17), `interrupt`
from thread C while B solves (crates/repl/src/main.rs:298-323), then `val` x V, `stats`, `free` on A."""
import os
import subprocess

import numpy as np
import pytest

from helpers import ROOT, make_grid, platform_defs
from timberborn_support_solver_amd import Encoding, PlatformLimits

PKG = os.path.join(ROOT, "timberborn_support_solver_amd")


def build_harness(tmp_path):
    exe = str(tmp_path / "abi_threads")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "abi_threads.c"),
                           "-L" + PKG, "-lmi355sat", "-Wl,-rpath," + PKG])
    return exe


def write_cnf(path, cnf):
    with open(path, "wb") as f:
        np.array([cnf.n_vars, cnf.n_clauses], dtype=np.int64).tofile(f)
        np.asarray(cnf.offsets, dtype=np.uint64).tofile(f)
        np.asarray(cnf.lits, dtype=np.int32).tofile(f)


def test_harness_builds_against_the_header_and_library(tmp_path):
    """No GPU needed: the C replay compiles against include/mi355sat.h with -Werror and links every symbol it uses."""
    exe = build_harness(tmp_path)
    out = subprocess.run([exe], capture_output=True)
    assert out.returncode == 2          # usage error, reached main(): the dynamic linker resolved the library


@pytest.mark.gpu
@pytest.mark.parametrize("terrain,pset,k,expect,intr_ms", [("rect16x16", "default", 4, 10, -1), ("rect16x16", "default", 3, 20, -1),
                                                           ("ex3", "1x1", 4, 10, -1), ("rect32x32", "default", 14, 0, 700),
                                                           ("rect16x16", "default", 4, 10, 60000)])
def test_rust_shim_call_sequence_replayed_in_c(tmp_path, terrain, pset, k, expect, intr_ms):
    """SAT (model read one literal at a time and checked clause by clause inside the C program), UNSAT, an
    interrupt from a third thread that stops a search no solver here finishes (-> 0), and an interrupter thread
    that outlives an easy solve (it fires after the verdict; nothing may crash)."""
    exe = build_harness(tmp_path)
    grid = make_grid(terrain)
    enc = Encoding.encode(platform_defs(pset), grid)
    cnf = enc.with_limits_into_cnf(PlatformLimits({(1, 1): k}))
    path = str(tmp_path / "cnf.bin")
    write_cnf(path, cnf)
    if intr_ms == 60000:
        intr_ms = 1500      # fires after the easy solve has returned, before free
    out = subprocess.run([exe, path, str(expect), str(intr_ms)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert f"result {expect}" in out.stdout and f"n_clauses={cnf.n_clauses}" in out.stdout
