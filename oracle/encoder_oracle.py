"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the reference's CNF generator and limit assembly.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED against the real reference: the reference is Rust, there is no
Rust toolchain here, and its own tests never touch the encoder.
What pins this file instead:
  * the exact V/C/L counts of SURVEY Appendix A (tests/golden/cnf_counts.json),
  * PicoSAT verdicts on the CNFs it generates (tests/golden/verdicts.json),
  * the README layouts for the validator (tests/golden/readme_layouts.json).

Unlike the product's C++ encoder (which derives the three edge sets directly
from the containment order), this restatement follows the reference's own
construction step by step: build the DAG over Platform and Point nodes from the
partial order, drop isolated nodes, take transitive closure and transitive
reduction of the explicit graph, then read the edge sets out of it.

Follows (file:line in /root/reference):
  src/encoder.rs:121-130   dims_platform_map (dims + flipped)
  src/encoder.rs:136-156   dag_by_partial_ord (edge smaller -> larger)
  src/encoder.rs:184-206   EncodingVars::new (variable allocation)
  src/encoder.rs:288-303   EncodingNode partial order
  src/encoder.rs:317-347   EncodingDag::new (retain_nodes, reduction, closure)
  src/encoder.rs:349-425   edge iterators, common successors, "maximal_from"
  src/encoder.rs:446-610   clause families
  src/encoder.rs:619-667   with_limits
  src/math/dimensions.rs:74-114  containment order; :138-156 row-major iteration
  src/math/point.rs:46-53  neighbour order (+x, +y, -x, -y)
  src/lib.rs:12            TERRAIN_SUPPORT_DISTANCE = 4
  src/platform.rs:23-32    PLATFORMS_DEFAULT
  src/world.rs:49-79       grid rows

Deterministic choices where the reference depends on HashMap order (documented
in DESIGN.md): dims sorted by (w, h); variables 1-based in allocation order;
graph nodes ordered platforms (sorted) then points row-major; edges iterated in
(source, target) node order.
"""
from itertools import combinations

TERRAIN_SUPPORT_DISTANCE = 4
PLATFORMS_DEFAULT = [(1, 1), (1, 2), (1, 3), (1, 4), (1, 5), (1, 6), (3, 3), (5, 5)]


def grid_rect(w, h):
    return [[True] * w for _ in range(h)]


def grid_from_rows(rows):
    """src/world.rs:49-79"""
    if not rows:
        raise ValueError("invalid length 0, expected 1 or more")
    width = max(len(r) for r in rows)
    grid = []
    for r in rows:
        row = []
        for c in r:
            if c == "X":
                row.append(True)
            elif c == " ":
                row.append(False)
            else:
                raise ValueError(f"invalid value: character `{c}`, expected `X` or ` `")
        row += [False] * (width - len(row))
        grid.append(row)
    return grid


def grid_from_toml(path):
    import re
    text = open(path, encoding="utf-8").read()
    m = re.search(r"grid\s*=\s*\[(.*?)\]", text, re.S)
    if not m:
        raise ValueError("Error parsing file: missing `grid`")
    return grid_from_rows(re.findall(r'"([^"]*)"', m.group(1)))


# ---- partial order (dimensions.rs:74-114, encoder.rs:288-303) -------------
def _dims_lt(a, b):
    return a != b and a[0] <= b[0] and a[1] <= b[1]


def _node_lt(a, b):
    ka, va = a
    kb, vb = b
    if ka == "plat" and kb == "plat":
        return _dims_lt(va, vb)
    if ka == "point" and kb == "plat":
        return 0 <= va[0] < vb[0] and 0 <= va[1] < vb[1]
    return False  # Platform < Point never; Point vs Point only Equal


class EncodingDag:
    """encoder.rs:307-426 on an explicit adjacency structure."""

    def __init__(self, dims):
        maxw = max([1] + [d[0] for d in dims])
        maxh = max([1] + [d[1] for d in dims])
        nodes = [("plat", d) for d in dims]
        nodes += [("point", (x, y)) for y in range(maxh) for x in range(maxw)]
        succ = {n: [m for m in nodes if _node_lt(n, m)] for n in nodes}
        pred = {n: [m for m in nodes if _node_lt(m, n)] for n in nodes}
        # retain_nodes: drop nodes with no neighbour at all (encoder.rs:331)
        nodes = [n for n in nodes if succ[n] or pred[n]]
        self.nodes = nodes
        # transitive closure by reachability over the explicit edges
        closure = {}
        for n in nodes:
            seen, stack = set(), list(succ[n])
            while stack:
                m = stack.pop()
                if m in seen:
                    continue
                seen.add(m)
                stack.extend(succ[m])
            closure[n] = seen
        self.closure = closure
        # transitive reduction: edge (u,v) survives iff v is not reachable from
        # another successor of u
        reduced = {}
        for u in nodes:
            reduced[u] = [v for v in succ[u]
                          if not any(v in closure[w] for w in succ[u] if w != v)]
        self.reduced = reduced

    def platform_edges_reduced(self):  # encoder.rs:355-362
        return [(u[1], v[1]) for u in self.nodes if u[0] == "plat"
                for v in self.reduced[u] if v[0] == "plat"]

    def point_platform_edges_reduced(self):  # encoder.rs:368-373
        return [(u[1], v[1]) for u in self.nodes if u[0] == "point"
                for v in self.reduced[u] if v[0] == "plat"]

    def platform_targets_by_source(self):  # encoder.rs:375-398
        return [[v for v in self.reduced[u]] for u in self.nodes if u[0] == "plat"]

    def common_platform_successors(self, a, b):  # encoder.rs:400-417
        return [n for n in self.nodes if n in self.closure[a] and n in self.closure[b] and n[0] == "plat"]

    def maximal_from(self, idx):  # encoder.rs:419-425 (keeps nodes without a predecessor in the set)
        return [n for n in idx if not any(n in self.closure[m] for m in idx)]


class Encoding:
    def __init__(self, platform_defs, grid):
        self.defs = list(platform_defs)
        self.grid = grid
        self.height = len(grid)
        self.width = len(grid[0])
        dims = sorted(set(self.defs) | {(h, w) for (w, h) in self.defs})
        self.dims = dims
        self.n_vars = 0
        self.plat_var = {}     # (x, y, dims) -> var
        self.terrain_var = {}  # (x, y) -> [vars]
        self.var_item = {}
        for y in range(self.height):
            for x in range(self.width):
                for d in dims:
                    self.n_vars += 1
                    self.plat_var[(x, y, d)] = self.n_vars
                    self.var_item[self.n_vars] = ("plat", (x, y), d)
                if grid[y][x]:
                    vs = []
                    for layer in range(TERRAIN_SUPPORT_DISTANCE):
                        self.n_vars += 1
                        vs.append(self.n_vars)
                        self.var_item[self.n_vars] = ("terrain", (x, y), layer)
                    self.terrain_var[(x, y)] = vs
        self.clauses = []
        self.family = {}
        self._encode()

    def _add(self, fam, lits):
        self.clauses.append(list(lits))
        self.family[fam] = self.family.get(fam, 0) + 1

    def _encode(self):
        dag = EncodingDag(self.dims)
        W, H = self.width, self.height
        plat_edges = dag.platform_edges_reduced()
        point_edges = dag.point_platform_edges_reduced()
        pair_specs = []
        for targets in dag.platform_targets_by_source():
            for a, b in combinations(targets, 2):
                common = dag.common_platform_successors(a, b)
                pair_specs.append((a[1], b[1], [n[1] for n in dag.maximal_from(common)]))
        self.plat_edges, self.point_edges, self.pair_specs = plat_edges, point_edges, pair_specs
        P = self.plat_var
        for y in range(H):
            for x in range(W):
                for smaller, larger in plat_edges:                       # :450-458
                    self._add("dag_impl", [-P[(x, y, larger)], P[(x, y, smaller)]])
                for a, b, succ in pair_specs:                             # :460-489
                    self._add("dag_pair", [-P[(x, y, a)], -P[(x, y, b)]] + [P[(x, y, s)] for s in succ])
                tv = self.terrain_var.get((x, y))
                if tv is not None:                                        # :500-516
                    cl = [-tv[TERRAIN_SUPPORT_DISTANCE - 1]]
                    for (ox, oy), d in point_edges:
                        v = P.get((x - ox, y - oy, d))
                        if v is not None:
                            cl.append(v)
                    self._add("coverage", cl)
                if tv is not None:                                        # :520-543
                    neigh = [(x + 1, y), (x, y + 1), (x - 1, y), (x, y - 1)]
                    nvars = [self.terrain_var[n] for n in neigh if n in self.terrain_var] + [tv]
                    for i in range(TERRAIN_SUPPORT_DISTANCE - 1):
                        self._add("terrain_layer", [-tv[i]] + [nv[i + 1] for nv in nvars])
                    self._add("top_unit", [tv[0]])
                for (ox, oy), d in point_edges:                           # :559-571
                    if (ox, oy) == (0, 0):
                        continue
                    if not (x + ox < W and y + oy < H):
                        continue
                    self._add("overlap_1x1", [-P[(x, y, d)], -P[(x + ox, y + oy, (1, 1))]])
                for (ox, oy), d in point_edges:                           # :576-596
                    if (ox, oy) == (0, 0) or oy != 0:
                        continue
                    for (qx, qy), d2 in point_edges:
                        if (qx, qy) == (0, 0) or qx != 0:
                            continue
                        v = P.get((x + ox - qx, y + oy - qy, d2))
                        if v is not None:
                            self._add("overlap_cross", [-P[(x, y, d)], -v])
                for (ox, oy), d in point_edges:                           # :601-609
                    if not (0 <= x + ox < W and 0 <= y + oy < H):
                        self._add("oob", [-P[(x, y, d)]])

    # with_limits (encoder.rs:619-667) restricted to card_limits; returns
    # (clauses, n_vars, [(lits, bound)])
    def with_limits(self, card_limits):
        clauses = [list(c) for c in self.clauses]
        n_vars = self.n_vars
        cards = []
        for d in sorted(card_limits):
            if d[0] != d[1]:
                lits = []
                for y in range(self.height):
                    for x in range(self.width):
                        n_vars += 1
                        lits.append(n_vars)
                        for dd in (d, (d[1], d[0])):
                            v = self.plat_var.get((x, y, dd))
                            if v is not None:
                                clauses.append([-v, n_vars])
            else:
                lits = [self.plat_var[(x, y, d)] for y in range(self.height) for x in range(self.width)
                        if (x, y, d) in self.plat_var]
            cards.append((lits, card_limits[d]))
        return clauses, n_vars, cards


def totalizer_ub(clauses, n_vars, inputs, max_out):
    """Upper-bound totalizer (sum >= j  =>  o_j), truncated at max_out outputs.
    Restates the published Bailleux-Boufkhad construction; rustsat's own encoder is
    [ext] and unavailable.  Returns (outputs, n_vars)."""
    def build(lo, hi):
        nonlocal n_vars
        if hi - lo == 1:
            return [inputs[lo]]
        mid = lo + (hi - lo) // 2
        a, b = build(lo, mid), build(mid, hi)
        m = min(len(a) + len(b), max_out)
        r = []
        for _ in range(m):
            n_vars += 1
            r.append(n_vars)
        for i in range(len(a) + 1):
            for j in range(len(b) + 1):
                s = i + j
                if s == 0 or s > m:
                    continue
                cl = []
                if i:
                    cl.append(-a[i - 1])
                if j:
                    cl.append(-b[j - 1])
                cl.append(r[s - 1])
                clauses.append(cl)
        return r
    if not inputs or max_out == 0:
        return [], n_vars
    return build(0, len(inputs)), n_vars


def gte_ub(clauses, n_vars, terms, bound):
    """Generalized totalizer for sum(w*l) <= bound (published construction, Joshi et al. 2015; rustsat's own
    is [ext]).  terms: [(lit, weight >= 0)].  Returns n_vars."""
    kept, total = [], 0
    for lit, wt in terms:
        if wt < 0:
            raise ValueError("negative weights are not supported")
        if wt == 0:
            continue
        if wt > bound:
            clauses.append([-lit])
            continue
        kept.append((lit, wt))
        total += wt
    if bound < 0:
        clauses.append([])
        return n_vars
    if total <= bound or not kept:
        return n_vars
    cap = bound + 1

    def build(lo, hi):
        nonlocal n_vars
        if hi - lo == 1:
            return {kept[lo][1]: kept[lo][0]}
        mid = lo + (hi - lo) // 2
        a, b, r = build(lo, mid), build(mid, hi), {}

        def out(w):
            nonlocal n_vars
            if w not in r:
                n_vars += 1
                r[w] = n_vars
            return r[w]
        for wa in sorted(a):
            clauses.append([-a[wa], out(min(wa, cap))])
        for wb in sorted(b):
            clauses.append([-b[wb], out(min(wb, cap))])
        for wa in sorted(a):
            for wb in sorted(b):
                clauses.append([-a[wa], -b[wb], out(min(wa + wb, cap))])
        return r
    root = build(0, len(kept))
    if cap in root:
        clauses.append([-root[cap]])
    return n_vars


def with_weights(enc, card_limits, weights, weight_limit):
    """Encoding::with_limits including weights (encoder.rs:619-667): returns (clauses, n_vars, cards, pb_terms)."""
    clauses = [list(c) for c in enc.clauses]
    n_vars = enc.n_vars
    cards, terms = [], []
    for d in sorted(set(card_limits) | set(weights)):
        if d[0] != d[1]:
            lits = []
            for y in range(enc.height):
                for x in range(enc.width):
                    n_vars += 1
                    lits.append(n_vars)
                    for dd in (d, (d[1], d[0])):
                        v = enc.plat_var.get((x, y, dd))
                        if v is not None:
                            clauses.append([-v, n_vars])
        else:
            lits = [enc.plat_var[(x, y, d)] for y in range(enc.height) for x in range(enc.width)]
        if d in card_limits:
            cards.append((lits, card_limits[d]))
        if weight_limit is not None and d in weights:
            terms += [(l, weights[d]) for l in lits]
    return clauses, n_vars, cards, terms


def into_cnf(clauses, n_vars, cards, keep_outputs=False, pbs=()):
    """SatInstance::into_cnf restatement ([ext]); degenerate cases as SURVEY §8c(iii)."""
    clauses = [list(c) for c in clauses]
    outputs = []
    for lits, k in cards:
        n = len(lits)
        outs = []
        if k >= n:
            pass
        elif k == 0:
            clauses += [[-l] for l in lits]
        elif k == n - 1 and not keep_outputs:
            clauses.append([-l for l in lits])
        else:
            outs, n_vars = totalizer_ub(clauses, n_vars, lits, k + 1)
            clauses.append([-outs[k]])
        outputs.append(outs)
    for terms, bound in pbs:
        n_vars = gte_ub(clauses, n_vars, terms, bound)
    return clauses, n_vars, outputs
