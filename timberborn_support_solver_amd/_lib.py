"""Loads the two in-tree shared libraries.  There is no fallback: if
libmi355sat.so (the HIP solver) is missing or cannot be loaded, importing the
solver fails loudly with instructions to build it."""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
SOLVER_LIB = os.path.join(_PKG, "libmi355sat.so")
HOST_LIB = os.path.join(_PKG, "libtbs_host.so")


class NativeLibraryMissing(ImportError):
    pass


def build(targets=("../libmi355sat.so", "../libtbs_host.so"), quiet=True):
    """Compile the in-tree libraries (hipcc --offload-arch=gfx950 for the solver)."""
    cmd = ["make", "-C", _CSRC] + list(targets)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)


def _load(path, what):
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f"{what} not found at {path}. Build it with `make -C {_CSRC}` "
            f"(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            f"There is no CPU fallback for the solver.")
    try:
        return ctypes.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise NativeLibraryMissing(f"cannot load {what} ({path}): {e}") from e


_solver = None
_host = None


def solver_lib():
    global _solver
    if _solver is None:
        _solver = _load(SOLVER_LIB, "libmi355sat.so (HIP solver)")
    return _solver


def host_lib():
    global _host
    if _host is None:
        _host = _load(HOST_LIB, "libtbs_host.so (host-side encoder)")
    return _host
