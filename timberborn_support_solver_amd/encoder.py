"""Python view of the host-side encoder mirror (include/tbs_host.h, C++ in
csrc/host/).  Class and method names follow the reference:

    WorldGrid                       src/world.rs:21-94
    PLATFORMS_DEFAULT               src/platform.rs:23-32
    Encoding.encode / with_limits   src/encoder.rs:435-667
    PlatformLimits                  src/encoder/platform_limits.rs
    PlatformLayout                  src/encoder/platform_layout.rs
"""
import ctypes

import numpy as np

from . import _lib

PLATFORMS_DEFAULT = [(1, 1), (1, 2), (1, 3), (1, 4), (1, 5), (1, 6), (3, 3), (5, 5)]
FAMILIES = ("dag_impl", "dag_pair", "coverage", "terrain_layer", "top_unit", "overlap_1x1", "overlap_cross", "oob")


class EncoderError(RuntimeError):
    pass


_bound = None


def _H():
    global _bound
    if _bound is None:
        L = _lib.host_lib()
        vp, i32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64
        L.tbs_last_error.restype = ctypes.c_char_p
        L.tbs_grid_from_toml.argtypes = [ctypes.c_char_p, vp, vp, vp, u64]
        L.tbs_encode.restype = vp
        L.tbs_encode.argtypes = [vp, i32, vp, i32, i32]
        L.tbs_encoding_free.argtypes = [vp]
        L.tbs_encoding_n_vars.argtypes = [vp]
        L.tbs_encoding_n_vars.restype = ctypes.c_uint32
        L.tbs_encoding_n_dims.argtypes = [vp]
        L.tbs_encoding_dims.argtypes = [vp, vp]
        L.tbs_encoding_family_counts.argtypes = [vp, vp]
        L.tbs_encoding_platform_var.argtypes = [vp, i32, i32, i32, i32]
        L.tbs_encoding_terrain_var.argtypes = [vp, i32, i32, i32]
        L.tbs_encoding_var_info.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.tbs_encoding_n_plat_edges.argtypes = [vp]
        L.tbs_encoding_plat_edges.argtypes = [vp, vp]
        L.tbs_encoding_n_point_edges.argtypes = [vp]
        L.tbs_encoding_point_edges.argtypes = [vp, vp]
        L.tbs_encoding_base_cnf.restype = vp
        L.tbs_encoding_base_cnf.argtypes = [vp]
        L.tbs_with_limits_into_cnf.restype = vp
        L.tbs_with_limits_into_cnf.argtypes = [vp, vp, i32, i32]
        L.tbs_with_limits_weights_into_cnf.restype = vp
        L.tbs_with_limits_weights_into_cnf.argtypes = [vp, vp, i32, vp, i32, i32, ctypes.c_int64, i32]
        L.tbs_layout_total_weight.argtypes = [vp, vp, i32]
        L.tbs_layout_total_weight.restype = ctypes.c_int64
        L.tbs_cnf_free.argtypes = [vp]
        L.tbs_cnf_n_vars.argtypes = [vp]
        L.tbs_cnf_n_vars.restype = ctypes.c_uint32
        for f in ("tbs_cnf_n_clauses", "tbs_cnf_n_lits", "tbs_cnf_n_card_outputs"):
            getattr(L, f).argtypes = [vp]
            getattr(L, f).restype = u64
        L.tbs_cnf_lits.argtypes = [vp]
        L.tbs_cnf_lits.restype = ctypes.POINTER(ctypes.c_int32)
        L.tbs_cnf_offsets.argtypes = [vp]
        L.tbs_cnf_offsets.restype = ctypes.POINTER(ctypes.c_uint64)
        L.tbs_cnf_card_outputs.argtypes = [vp]
        L.tbs_cnf_card_outputs.restype = ctypes.POINTER(ctypes.c_int32)
        L.tbs_layout_from_model.restype = vp
        L.tbs_layout_from_model.argtypes = [vp, vp, u64]
        L.tbs_layout_from_platforms.restype = vp
        L.tbs_layout_from_platforms.argtypes = [vp, i32]
        L.tbs_layout_free.argtypes = [vp]
        L.tbs_layout_count.argtypes = [vp]
        L.tbs_layout_platforms.argtypes = [vp, vp]
        L.tbs_layout_validate.argtypes = [vp, vp, i32, i32, vp]
        L.tbs_layout_trivial_optimization.argtypes = [vp, vp, i32, i32]
        _bound = L
    return _bound


def _err():
    return (_H().tbs_last_error() or b"").decode()


class WorldGrid:
    """Row-major bool grid; rows of 'X' (terrain) / ' ' (src/world.rs:49-79)."""

    def __init__(self, cells, width, height):
        self.cells = np.ascontiguousarray(cells, dtype=np.uint8).reshape(height, width)
        self.width, self.height = width, height

    @staticmethod
    def rect(w, h):
        return WorldGrid(np.ones(w * h, dtype=np.uint8), w, h)

    @staticmethod
    def from_rows(rows):
        if not rows:
            raise EncoderError("invalid length 0, expected 1 or more")
        width = max(len(r) for r in rows)
        cells = np.zeros((len(rows), width), dtype=np.uint8)
        for y, r in enumerate(rows):
            for x, c in enumerate(r):
                if c == "X":
                    cells[y, x] = 1
                elif c != " ":
                    raise EncoderError(f"invalid value: character `{c}`, expected `X` or ` `")
        return WorldGrid(cells, width, len(rows))

    @staticmethod
    def from_toml(path):
        w, h = ctypes.c_int32(0), ctypes.c_int32(0)
        if _H().tbs_grid_from_toml(path.encode(), ctypes.byref(w), ctypes.byref(h), None, 0) != 0:
            raise EncoderError(_err())
        cells = np.zeros(w.value * h.value, dtype=np.uint8)
        _H().tbs_grid_from_toml(path.encode(), ctypes.byref(w), ctypes.byref(h), cells.ctypes.data, cells.size)
        return WorldGrid(cells, w.value, h.value)

    def rows(self):
        return ["".join("X" if c else " " for c in r) for r in self.cells]


class Cnf:
    """Plain CNF in CSR form (DIMACS literals)."""

    def __init__(self, handle):
        H = _H()
        self.n_vars = H.tbs_cnf_n_vars(handle)
        nc, nl = H.tbs_cnf_n_clauses(handle), H.tbs_cnf_n_lits(handle)
        self.lits = np.ctypeslib.as_array(H.tbs_cnf_lits(handle), shape=(nl,)).copy() if nl else np.zeros(0, np.int32)
        self.offsets = np.ctypeslib.as_array(H.tbs_cnf_offsets(handle), shape=(nc + 1,)).copy()
        no = H.tbs_cnf_n_card_outputs(handle)
        self.card_outputs = (np.ctypeslib.as_array(H.tbs_cnf_card_outputs(handle), shape=(no,)).copy()
                             if no else np.zeros(0, np.int32))
        H.tbs_cnf_free(handle)

    @property
    def n_clauses(self):
        return len(self.offsets) - 1

    def clauses(self):
        return [self.lits[self.offsets[i]:self.offsets[i + 1]].tolist() for i in range(self.n_clauses)]


class PlatformLimits:
    """card_limits: {(w,h) platform def dims: max count} (platform_limits.rs:6-12)."""

    def __init__(self, card_limits=None, weights=None, weight_limit=None):
        self.card_limits = dict(card_limits or {})
        self.weights = dict(weights or {})          # {(w,h): weight}   platform_limits.rs:10
        self.weight_limit = weight_limit            # Option<isize>     platform_limits.rs:12

    @staticmethod
    def new_unweighted(limits):
        return PlatformLimits(limits)


class Encoding:
    def __init__(self, handle, grid, defs):
        self._h, self.grid, self.defs = handle, grid, list(defs)

    def __del__(self):
        if getattr(self, "_h", None):
            _H().tbs_encoding_free(self._h)
            self._h = None

    @staticmethod
    def encode(platform_defs, grid):
        defs = np.asarray([x for d in platform_defs for x in d], dtype=np.int32)
        cells = np.ascontiguousarray(grid.cells, dtype=np.uint8)
        h = _H().tbs_encode(defs.ctypes.data if defs.size else None, len(platform_defs), cells.ctypes.data,
                            grid.width, grid.height)
        if not h:
            raise EncoderError(_err())
        return Encoding(h, grid, platform_defs if platform_defs else PLATFORMS_DEFAULT)

    @property
    def n_vars(self):
        return _H().tbs_encoding_n_vars(self._h)

    def platform_dims(self):
        n = _H().tbs_encoding_n_dims(self._h)
        out = np.zeros(2 * n, dtype=np.int32)
        _H().tbs_encoding_dims(self._h, out.ctypes.data)
        return [tuple(out[2 * i:2 * i + 2]) for i in range(n)]

    def family_counts(self):
        out = np.zeros(8, dtype=np.uint64)
        _H().tbs_encoding_family_counts(self._h, out.ctypes.data)
        return dict(zip(FAMILIES, (int(x) for x in out)))

    def platform_var(self, x, y, dims):
        return _H().tbs_encoding_platform_var(self._h, x, y, dims[0], dims[1])

    def terrain_var(self, x, y, layer):
        return _H().tbs_encoding_terrain_var(self._h, x, y, layer)

    def platform_edges_reduced(self):
        n = _H().tbs_encoding_n_plat_edges(self._h)
        out = np.zeros(4 * n, dtype=np.int32)
        _H().tbs_encoding_plat_edges(self._h, out.ctypes.data)
        return [((out[4 * i], out[4 * i + 1]), (out[4 * i + 2], out[4 * i + 3])) for i in range(n)]

    def point_platform_edges_reduced(self):
        n = _H().tbs_encoding_n_point_edges(self._h)
        out = np.zeros(4 * n, dtype=np.int32)
        _H().tbs_encoding_point_edges(self._h, out.ctypes.data)
        return [((out[4 * i], out[4 * i + 1]), (out[4 * i + 2], out[4 * i + 3])) for i in range(n)]

    def base_cnf(self):
        return Cnf(_H().tbs_encoding_base_cnf(self._h))

    def with_limits_into_cnf(self, limits, sweep=False):
        """Encoding::with_limits(&limits) followed by SatInstance::into_cnf()
        (crates/repl/src/main.rs:292-293).  sweep=True keeps the totalizer outputs in
        Cnf.card_outputs so tighter bounds can be posed as assumptions."""
        items = sorted(limits.card_limits.items())
        arr = np.asarray([x for (d, k) in items for x in (d[0], d[1], k)], dtype=np.int64)
        witems = sorted(getattr(limits, "weights", {}).items())
        warr = np.asarray([x for (d, wt) in witems for x in (d[0], d[1], wt)], dtype=np.int64)
        wl = getattr(limits, "weight_limit", None)
        h = _H().tbs_with_limits_weights_into_cnf(self._h, arr.ctypes.data if arr.size else None, len(items),
                                                  warr.ctypes.data if warr.size else None, len(witems),
                                                  0 if wl is None else 1, 0 if wl is None else int(wl), 1 if sweep else 0)
        if not h:
            raise EncoderError(_err())
        return Cnf(h)


class ValidationResult:
    def __init__(self, unsupported, overlapping, oob):
        self.n_unsupported_terrain, self.n_overlapping_platforms, self.n_out_of_bounds_platforms = unsupported, overlapping, oob

    def is_valid(self):
        return not (self.n_unsupported_terrain or self.n_overlapping_platforms or self.n_out_of_bounds_platforms)


class PlatformLayout:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            _H().tbs_layout_free(self._h)
            self._h = None

    @staticmethod
    def from_assignment(model, encoding):
        model = np.ascontiguousarray(model, dtype=np.int8)
        h = _H().tbs_layout_from_model(encoding._h, model.ctypes.data, len(model))
        if not h:
            raise EncoderError(_err())
        return PlatformLayout(h)

    @staticmethod
    def from_platforms(platforms):
        """platforms: iterable of (x, y, def_w, def_h, rotated)"""
        arr = np.asarray([v for p in platforms for v in p], dtype=np.int32)
        return PlatformLayout(_H().tbs_layout_from_platforms(arr.ctypes.data if arr.size else None, len(arr) // 5))

    def platform_count(self):
        return _H().tbs_layout_count(self._h)

    def platforms(self):
        n = self.platform_count()
        out = np.zeros(5 * n, dtype=np.int32)
        _H().tbs_layout_platforms(self._h, out.ctypes.data)
        return [tuple(int(v) for v in out[5 * i:5 * i + 5]) for i in range(n)]

    def platform_stats(self):
        stats = {}
        for (_, _, w, h, _) in self.platforms():
            stats[(w, h)] = stats.get((w, h), 0) + 1
        return stats

    def total_weight(self, weights):
        """platform_layout.rs:174-183: a platform counts the weight of every type contained in its definition dims."""
        items = sorted(weights.items())
        arr = np.asarray([x for (d, wt) in items for x in (d[0], d[1], wt)], dtype=np.int64)
        return int(_H().tbs_layout_total_weight(self._h, arr.ctypes.data if arr.size else None, len(items)))

    def run_trivial_optimization(self, grid):
        cells = np.ascontiguousarray(grid.cells, dtype=np.uint8)
        _H().tbs_layout_trivial_optimization(self._h, cells.ctypes.data, grid.width, grid.height)

    def validate(self, grid):
        counts = np.zeros(3, dtype=np.int32)
        cells = np.ascontiguousarray(grid.cells, dtype=np.uint8)
        rc = _H().tbs_layout_validate(self._h, cells.ctypes.data, grid.width, grid.height, counts.ctypes.data)
        if rc < 0:
            raise EncoderError(_err())
        return ValidationResult(int(counts[0]), int(counts[1]), int(counts[2]))
