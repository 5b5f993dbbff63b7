"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the reference's model -> layout -> validate post-processing,
the reference's own correctness check for SAT answers (crates/repl/src/main.rs:353-361).

Pinned by the four printed layouts of the reference README (README.md:46-116,
tests/golden/readme_layouts.json): all must validate with support distance 4.

Follows (file:line in /root/reference):
  src/encoder/platform_layout.rs:26-52    from_assignment
  src/encoder/platform_layout.rs:58-60    platform_count
  src/encoder/platform_layout.rs:85-149   validate
  src/encoder.rs:232-249                  var_to_platform
  src/platform.rs:118-120                 Platform::dims (rotation)
"""
TERRAIN_SUPPORT_DISTANCE = 4


def _dims_lt(a, b):
    return a != b and a[0] <= b[0] and a[1] <= b[1]


def from_assignment(model, enc):
    """model: sequence, model[v-1] in {1,-1,0}; enc: encoder_oracle.Encoding.
    Returns {(x,y): (def_dims, rotated)}."""
    platforms = {}
    for v in range(1, enc.n_vars + 1):
        if v - 1 >= len(model) or model[v - 1] <= 0:
            continue
        item = enc.var_item.get(v)
        if item is None or item[0] != "plat":
            continue
        _, point, dims = item
        d = next(p for p in enc.defs if p == dims or (p[1], p[0]) == dims)
        plat = (d, d != dims)
        prev = platforms.get(point)
        if prev is None or _dims_lt(prev[0], plat[0]):
            platforms[point] = plat
    return platforms


def platform_count(platforms):
    return len(platforms)


def validate(platforms, grid):
    """Returns (unsupported_terrain:set, overlapping:set, out_of_bounds:set)."""
    H, W = len(grid), len(grid[0])
    supported = {(x, y): False for y in range(H) for x in range(W) if grid[y][x]}
    occupied = {}
    overlapping, oob = set(), set()
    for point, (d, rotated) in platforms.items():
        w, h = (d[1], d[0]) if rotated else d
        key = (point, d, rotated)
        for oy in range(h):
            for ox in range(w):
                p = (point[0] + ox, point[1] + oy)
                if not (0 <= p[0] < W and 0 <= p[1] < H):
                    oob.add(key)
                    continue
                if p in occupied:
                    overlapping.add(key)
                    overlapping.add(occupied[p])
                else:
                    occupied[p] = key
                if p in supported:
                    supported[p] = True
    for _ in range(TERRAIN_SUPPORT_DISTANCE - 1):
        grow = set()
        for (x, y), s in supported.items():
            if s:
                grow.update([(x + 1, y), (x, y + 1), (x - 1, y), (x, y - 1)])
        for p in grow:
            if p in supported:
                supported[p] = True
    unsupported = {p for p, s in supported.items() if not s}
    return unsupported, overlapping, oob


def is_valid(result):
    return not any(result)
