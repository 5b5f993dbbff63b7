// tbs_host.cpp — implementation of the host-side mirror declared in tbs_host.hpp.
// Every block cites the reference lines it follows; nothing is copied — the
// reference builds a petgraph DAG and reads edge sets out of it, here the same
// three edge sets are computed directly from the partial order.
#include "tbs_host.hpp"

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>

namespace tbs {

// ---------------------------------------------------------------- world
WorldGrid WorldGrid::rect(int w, int h) {
    WorldGrid g;
    g.width = w;
    g.height = h;
    g.cells.assign((size_t)w * h, 1);
    return g;
}

// src/world.rs:49-79 — 'X' -> true, ' ' -> false, anything else is an error;
// width = longest row, shorter rows are padded with false.
WorldGrid WorldGrid::from_rows(const std::vector<std::string>& rows) {
    if (rows.empty()) throw std::runtime_error("invalid length 0, expected 1 or more");
    size_t maxw = 0;
    for (auto& r : rows) maxw = std::max(maxw, r.size());
    WorldGrid g;
    g.width = (int)maxw;
    g.height = (int)rows.size();
    g.cells.assign(maxw * rows.size(), 0);
    for (size_t y = 0; y < rows.size(); y++)
        for (size_t x = 0; x < rows[y].size(); x++) {
            char c = rows[y][x];
            if (c == 'X') g.cells[y * maxw + x] = 1;
            else if (c != ' ')
                throw std::runtime_error(std::string("invalid value: character `") + c +
                                         "`, expected `X` or ` `");
        }
    return g;
}

// Project file: `[world]` table with `grid = [ "row", ... ]` (src/lib.rs:14-17).
WorldGrid WorldGrid::from_toml_file(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Error reading file: " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string s = ss.str();
    size_t p = s.find("grid");
    if (p == std::string::npos) throw std::runtime_error("Error parsing file: missing `grid`");
    p = s.find('[', p);
    if (p == std::string::npos) throw std::runtime_error("Error parsing file: missing `[`");
    std::vector<std::string> rows;
    for (size_t i = p + 1; i < s.size(); i++) {
        if (s[i] == ']') break;
        if (s[i] == '#') {  // comment to end of line
            while (i < s.size() && s[i] != '\n') i++;
            continue;
        }
        if (s[i] == '"') {
            size_t e = s.find('"', i + 1);
            if (e == std::string::npos) throw std::runtime_error("Error parsing file: unterminated string");
            rows.push_back(s.substr(i + 1, e - i - 1));
            i = e;
        }
    }
    return from_rows(rows);
}

std::vector<Dims> platforms_default() {  // src/platform.rs:23-32
    return {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {3, 3}, {5, 5}};
}

// ---------------------------------------------------------------- totalizer
// rustsat's own totalizer is not available ([ext]); any sound at-most-k
// encoding gives the same verdicts and the same feasible layouts (SURVEY §8c ii).
std::vector<int32_t> totalizer_ub(Cnf& cnf, uint32_t& n_vars, const std::vector<int32_t>& inputs,
                                  size_t max_out) {
    struct Rec {
        Cnf& cnf;
        uint32_t& n_vars;
        const std::vector<int32_t>& in;
        size_t max_out;
        std::vector<int32_t> build(size_t lo, size_t hi) {
            if (hi - lo == 1) return {in[lo]};
            size_t mid = lo + (hi - lo) / 2;
            std::vector<int32_t> a = build(lo, mid), b = build(mid, hi);
            size_t m = std::min(a.size() + b.size(), max_out);
            std::vector<int32_t> r(m);
            for (size_t i = 0; i < m; i++) r[i] = (int32_t)++n_vars;
            for (size_t i = 0; i <= a.size(); i++)
                for (size_t j = 0; j <= b.size(); j++) {
                    size_t s = i + j;
                    if (s == 0 || s > m) continue;
                    int32_t c[3];
                    size_t n = 0;
                    if (i) c[n++] = -a[i - 1];
                    if (j) c[n++] = -b[j - 1];
                    c[n++] = r[s - 1];
                    cnf.add(c, n);
                }
            return r;
        }
    } rec{cnf, n_vars, inputs, max_out};
    if (inputs.empty() || max_out == 0) return {};
    std::vector<int32_t> out = rec.build(0, inputs.size());
    if (n_vars > cnf.n_vars) cnf.n_vars = n_vars;
    return out;
}

// Generalized totalizer for sum(w_i * l_i) <= bound, w_i > 0 (the GUI's weight bound,
// src/encoder.rs:654-663 -> rustsat PbConstraint -> GTE in into_cnf [ext]; restated from the
// published construction: Joshi, Martins, Manquinho 2015).  Each node maps every reachable weight
// sum (capped at bound+1) to an output variable; only the sum >= w => out_w direction is encoded.
static void gte_ub(Cnf& cnf, uint32_t& n_vars, const std::vector<std::pair<int32_t, long>>& terms_in, long bound) {
    std::vector<std::pair<int32_t, long>> terms;
    long total = 0;
    for (auto& t : terms_in) {
        if (t.second < 0) throw std::runtime_error("into_cnf: negative platform weights are not supported");
        if (t.second == 0) continue;
        if (t.second > bound) { cnf.add({-t.first}); continue; }   // alone already over the bound
        terms.push_back(t);
        total += t.second;
    }
    if (bound < 0) { cnf.add(std::vector<int32_t>{}); return; }    // nothing can satisfy a negative bound
    if (total <= bound || terms.empty()) return;
    const long cap = bound + 1;
    typedef std::map<long, int32_t> Node;
    struct Rec {
        Cnf& cnf; uint32_t& n_vars; const std::vector<std::pair<int32_t, long>>& t; long cap;
        Node build(size_t lo, size_t hi) {
            if (hi - lo == 1) return Node{{t[lo].second, t[lo].first}};
            size_t mid = lo + (hi - lo) / 2;
            Node a = build(lo, mid), b = build(mid, hi), r;
            auto out = [&](long w) { auto it = r.find(w); if (it == r.end()) it = r.emplace(w, (int32_t)++n_vars).first; return it->second; };
            for (auto& x : a) cnf.add({-x.second, out(std::min(x.first, cap))});
            for (auto& y : b) cnf.add({-y.second, out(std::min(y.first, cap))});
            for (auto& x : a)
                for (auto& y : b) cnf.add({-x.second, -y.second, out(std::min(x.first + y.first, cap))});
            return r;
        }
    } rec{cnf, n_vars, terms, cap};
    Node root = rec.build(0, terms.size());
    auto it = root.find(cap);
    if (it != root.end()) cnf.add({-it->second});
    if (n_vars > cnf.n_vars) cnf.n_vars = n_vars;
}

// SatInstance::into_cnf ([ext], called at crates/repl/src/main.rs:293 and
// crates/gui/src/solver_backend.rs:78).  Degenerate bounds are simplified the way
// SURVEY §8c (iii) records for rustsat: k >= n dropped, k == 0 -> n negative
// units, k == n-1 -> one clause.
Cnf SatInstance::into_cnf(std::vector<std::vector<int32_t>>* out_card_outputs) const {
    Cnf out = cnf;
    uint32_t nv = n_vars;
    if (out.n_vars > nv) nv = out.n_vars;
    for (const CardUb& c : cards) {
        std::vector<int32_t> outs;
        size_t n = c.lits.size(), k = c.bound;
        if (k >= n) {
            // trivially satisfied
        } else if (k == 0) {
            for (int32_t l : c.lits) out.add({-l});
        } else if (k == n - 1 && !out_card_outputs) {
            std::vector<int32_t> cl;
            for (int32_t l : c.lits) cl.push_back(-l);
            out.add(cl);
        } else {
            outs = totalizer_ub(out, nv, c.lits, k + 1);
            out.add({-outs[k]});  // NOT (at least k+1)
        }
        if (out_card_outputs) out_card_outputs->push_back(outs);
    }
    for (const PbUb& pb : pbs) gte_ub(out, nv, pb.terms, pb.bound);
    out.n_vars = std::max(out.n_vars, nv);
    return out;
}

// ---------------------------------------------------------------- encoder
int Encoding::dim_index(Dims d) const {
    auto it = std::lower_bound(dims_.begin(), dims_.end(), d);
    if (it == dims_.end() || *it != d) return -1;
    return (int)(it - dims_.begin());
}

int32_t Encoding::var_for_dims_at(int x, int y, Dims d) const {
    if (!grid_.in_bounds(x, y)) return 0;
    int di = dim_index(d);
    if (di < 0) return 0;
    return tile_base_[(size_t)y * grid_.width + x] + di;
}

int32_t Encoding::terrain_var(int x, int y, int layer) const {
    if (!grid_.terrain(x, y) || layer < 0 || layer >= TERRAIN_SUPPORT_DISTANCE) return 0;
    return tile_base_[(size_t)y * grid_.width + x] + (int32_t)dims_.size() + layer;
}

const EncodedItem* Encoding::item(int32_t var) const {
    if (var < 1 || (size_t)var > items_.size()) return nullptr;
    return &items_[var - 1];
}

// encoder.rs:232-249: the platform def is the one whose dims or flipped dims equal
// the variable's dims; rotated = def dims differ from the variable's dims.
bool Encoding::var_to_platform(int32_t var, Platform* out) const {
    const EncodedItem* it = item(var);
    if (!it || !it->is_platform) return false;
    for (const Dims& def : defs_) {
        if (def == it->dims || def.flipped() == it->dims) {
            out->point = it->point;
            out->def = def;
            out->rotated = def != it->dims;
            return true;
        }
    }
    return false;
}

std::string Encoding::lit_readable_name(int32_t lit) const {  // encoder.rs:251-273
    const EncodedItem* it = item(lit < 0 ? -lit : lit);
    if (!it) return "";
    char buf[64];
    if (it->is_platform)
        snprintf(buf, sizeof buf, "%sP%dx%d(%d;%d)", lit < 0 ? "~" : "", it->dims.w, it->dims.h,
                 it->point.x, it->point.y);
    else
        snprintf(buf, sizeof buf, "%sT%d(%d;%d)", lit < 0 ? "~" : "", it->layer, it->point.x, it->point.y);
    return buf;
}

std::vector<int32_t> Encoding::iter_dims_vars(Dims d) const {  // encoder.rs:218-222
    std::vector<int32_t> v;
    int di = dim_index(d);
    if (di < 0) return v;
    for (int32_t b : tile_base_) v.push_back(b + di);
    return v;
}

// EncodingDag (encoder.rs:280-426).  Nodes: Platform(dims) for every dims incl.
// rotations, Point(p) for every p in the bounding box of all dims.  Order:
// platforms by containment; Point(p) < Platform(d) iff d contains p.  The
// reference takes the transitive reduction/closure of that DAG; since the relation
// is already transitive the closure is the relation itself and an edge a->b is in
// the reduction iff no c has a < c < b (c is necessarily a platform).
void Encoding::build_dag() {
    const size_t n = dims_.size();
    auto lt = [&](size_t a, size_t b) { return dims_[a].strictly_within(dims_[b]); };
    // platform edges (smaller -> larger), encoder.rs:355-362
    for (size_t a = 0; a < n; a++)
        for (size_t b = 0; b < n; b++) {
            if (!lt(a, b)) continue;
            bool between = false;
            for (size_t c = 0; c < n && !between; c++) between = lt(a, c) && lt(c, b);
            if (!between) plat_edges_.push_back({dims_[a], dims_[b]});
        }
    // point -> minimal containing platform, encoder.rs:368-373; points with no
    // containing platform drop out (encoder.rs:331)
    int maxw = 1, maxh = 1;
    for (auto& d : dims_) { maxw = std::max(maxw, d.w); maxh = std::max(maxh, d.h); }
    for (int y = 0; y < maxh; y++)
        for (int x = 0; x < maxw; x++)
            for (size_t b = 0; b < n; b++) {
                if (!dims_[b].contains_point(x, y)) continue;
                bool smaller = false;
                for (size_t c = 0; c < n && !smaller; c++)
                    smaller = dims_[c].contains_point(x, y) && lt(c, b);
                if (!smaller) point_edges_.push_back({Point{x, y}, dims_[b]});
            }
    // incomparable siblings, encoder.rs:460-489 with :375-425
    for (size_t s = 0; s < n; s++) {
        std::vector<size_t> targets;
        for (auto& e : plat_edges_)
            if (e.first == dims_[s]) targets.push_back((size_t)dim_index(e.second));
        for (size_t i = 0; i < targets.size(); i++)
            for (size_t j = i + 1; j < targets.size(); j++) {
                size_t a = targets[i], b = targets[j];
                std::vector<size_t> common;  // common_platform_successors
                for (size_t c = 0; c < n; c++)
                    if (lt(a, c) && lt(b, c)) common.push_back(c);
                PairClause pc{dims_[a], dims_[b], {}};
                for (size_t c : common) {  // "maximal_from": keep those with no predecessor in the set
                    bool has_pred = false;
                    for (size_t m : common) has_pred = has_pred || lt(m, c);
                    if (!has_pred) pc.succ.push_back(dims_[c]);
                }
                pair_clauses_.push_back(pc);
            }
    }
}

Encoding Encoding::encode(const std::vector<Dims>& platform_defs, const WorldGrid& terrain) {
    Encoding e;
    e.grid_ = terrain;
    e.defs_ = platform_defs;
    {  // dims_platform_map, encoder.rs:121-130 (sorted instead of HashMap order)
        std::set<Dims> s;
        for (auto& d : platform_defs) {
            if (d.w <= 0 || d.h <= 0) throw std::runtime_error("empty platform dimensions");
            s.insert(d);
            s.insert(d.flipped());
        }
        e.dims_.assign(s.begin(), s.end());
    }
    const int W = terrain.width, H = terrain.height;
    const int nd = (int)e.dims_.size();
    // EncodingVars::new, encoder.rs:184-206: per tile, one var per dims then 4
    // terrain-layer vars if the tile is terrain; tiles row-major (dimensions.rs:138-156)
    e.tile_base_.resize((size_t)W * H);
    SatInstance& inst = e.instance_;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            e.tile_base_[(size_t)y * W + x] = (int32_t)inst.n_vars + 1;
            for (int d = 0; d < nd; d++) {
                inst.new_var();
                e.items_.push_back(EncodedItem{true, Point{x, y}, e.dims_[d], 0});
            }
            if (terrain.terrain(x, y))
                for (int l = 0; l < TERRAIN_SUPPORT_DISTANCE; l++) {
                    inst.new_var();
                    e.items_.push_back(EncodedItem{false, Point{x, y}, Dims{}, l});
                }
        }
    e.build_dag();
    const Dims one{1, 1};
    Cnf& cnf = inst.cnf;
    auto P = [&](int x, int y, Dims d) { return e.var_for_dims_at(x, y, d); };
    std::vector<int32_t> cl;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            // ===== Platform selection DAG: larger -> smaller, encoder.rs:450-458
            for (auto& ed : e.plat_edges_) {
                cnf.add({-P(x, y, ed.second), P(x, y, ed.first)});
                e.fam_[F_DAG_IMPL]++;
            }
            // (a & b) -> (minimal common successors), encoder.rs:460-489
            for (auto& pc : e.pair_clauses_) {
                cl.clear();
                cl.push_back(-P(x, y, pc.a));
                cl.push_back(-P(x, y, pc.b));
                for (auto& d : pc.succ) cl.push_back(P(x, y, d));
                cnf.add(cl);
                e.fam_[F_DAG_PAIR]++;
            }
            const bool is_terrain = terrain.terrain(x, y);
            // ===== Platform-terrain: T_last(p) -> OR of platforms covering p, encoder.rs:500-516
            if (is_terrain) {
                cl.clear();
                cl.push_back(-e.terrain_var(x, y, TERRAIN_SUPPORT_DISTANCE - 1));
                for (auto& pe : e.point_edges_) {
                    int32_t v = P(x - pe.first.x, y - pe.first.y, pe.second);
                    if (v) cl.push_back(v);
                }
                cnf.add(cl);
                e.fam_[F_COVERAGE]++;
            }
            // ===== Terrain support: T_i(p) -> OR T_{i+1}(n), n in N4(p)+{p}, encoder.rs:520-543
            if (is_terrain) {
                const int nx[4] = {x + 1, x, x - 1, x}, ny[4] = {y, y + 1, y, y - 1};  // point.rs:46-53
                for (int i = 0; i + 1 < TERRAIN_SUPPORT_DISTANCE; i++) {
                    cl.clear();
                    cl.push_back(-e.terrain_var(x, y, i));
                    for (int k = 0; k < 4; k++) {
                        int32_t v = e.terrain_var(nx[k], ny[k], i + 1);
                        if (v) cl.push_back(v);
                    }
                    cl.push_back(e.terrain_var(x, y, i + 1));
                    cnf.add(cl);
                    e.fam_[F_TERRAIN_LAYER]++;
                }
                cnf.add({e.terrain_var(x, y, 0)});
                e.fam_[F_TOP_UNIT]++;
            }
            // ===== Overlap: another platform's corner inside this one, encoder.rs:559-571
            for (auto& pe : e.point_edges_) {
                if (pe.first.x == 0 && pe.first.y == 0) continue;
                int32_t other = P(x + pe.first.x, y + pe.first.y, one);
                if (x + pe.first.x >= W || y + pe.first.y >= H) continue;
                if (!other) throw std::runtime_error("encode: the overlap clauses need a 1x1 platform type");
                cnf.add({-P(x, y, pe.second), -other});
                e.fam_[F_OVERLAP_1X1]++;
            }
            // ===== Overlap: row-like platform crossing a column-like one, encoder.rs:576-596
            for (auto& pe : e.point_edges_) {
                if ((pe.first.x == 0 && pe.first.y == 0) || pe.first.y != 0) continue;
                for (auto& qe : e.point_edges_) {
                    if ((qe.first.x == 0 && qe.first.y == 0) || qe.first.x != 0) continue;
                    int32_t other = P(x + pe.first.x - qe.first.x, y + pe.first.y - qe.first.y, qe.second);
                    if (!other) continue;
                    cnf.add({-P(x, y, pe.second), -other});
                    e.fam_[F_OVERLAP_CROSS]++;
                }
            }
            // ===== Out of bounds, encoder.rs:601-609
            for (auto& pe : e.point_edges_) {
                if (terrain.in_bounds(x + pe.first.x, y + pe.first.y)) continue;
                cnf.add({-P(x, y, pe.second)});
                e.fam_[F_OOB]++;
            }
        }
    cnf.n_vars = inst.n_vars;
    return e;
}

// Encoding::with_limits, encoder.rs:619-667.  Keys are visited in sorted order
// (the reference's HashMap order is unspecified).
SatInstance Encoding::with_limits(const PlatformLimits& limits) const {
    SatInstance inst = instance_;
    std::set<Dims> keys;
    for (auto& kv : limits.card_limits) keys.insert(kv.first);
    for (auto& kv : limits.weights) keys.insert(kv.first);
    PbUb weight_pb;
    weight_pb.bound = limits.weight_limit;
    for (const Dims& type : keys) {
        std::vector<int32_t> lits;
        if (type.rectangular()) {
            // a fresh per-tile var implied by either orientation, encoder.rs:629-641
            for (size_t t = 0; t < tile_base_.size(); t++) {
                int32_t lim = inst.new_var();
                lits.push_back(lim);
                int x = (int)(t % grid_.width), y = (int)(t / grid_.width);
                for (Dims d : {type, type.flipped()}) {
                    int32_t v = var_for_dims_at(x, y, d);
                    if (v) inst.cnf.add({-v, lim});
                }
            }
        } else {
            lits = iter_dims_vars(type);  // encoder.rs:642-647
        }
        auto c = limits.card_limits.find(type);
        if (c != limits.card_limits.end()) inst.cards.push_back(CardUb{lits, c->second});
        auto w = limits.weights.find(type);
        if (limits.has_weight_limit && w != limits.weights.end())
            for (int32_t l : lits) weight_pb.terms.push_back({l, w->second});
    }
    if (limits.has_weight_limit) inst.pbs.push_back(weight_pb);
    inst.cnf.n_vars = std::max(inst.cnf.n_vars, inst.n_vars);
    return inst;
}

// ---------------------------------------------------------------- layout
// PlatformLayout::from_assignment, platform_layout.rs:26-52: every true platform
// variable proposes a platform at its tile; a tile keeps the proposal whose
// *definition* dims are strictly larger (partial order) than the current one.
PlatformLayout PlatformLayout::from_assignment(const int8_t* model, size_t n_vars, const Encoding& enc) {
    PlatformLayout lay;
    for (size_t v = 1; v <= n_vars; v++) {
        if (model[v - 1] <= 0) continue;
        Platform plat;
        if (!enc.var_to_platform((int32_t)v, &plat)) continue;
        auto it = lay.platforms_.find(plat.point);
        if (it == lay.platforms_.end()) lay.platforms_[plat.point] = plat;
        else if (it->second.def.strictly_within(plat.def)) it->second = plat;
    }
    return lay;
}

PlatformLayout PlatformLayout::from_platforms(const std::vector<Platform>& p) {
    PlatformLayout lay;
    for (auto& x : p) lay.platforms_[x.point] = x;
    return lay;
}

std::map<Dims, size_t> PlatformLayout::platform_stats() const {  // platform_layout.rs:67-79
    std::map<Dims, size_t> m;
    for (auto& kv : platforms_) m[kv.second.def]++;
    return m;
}

// PlatformLayout::validate, platform_layout.rs:85-149
ValidationResult PlatformLayout::validate(const WorldGrid& world) const {
    const int W = world.width, H = world.height;
    std::vector<int8_t> supported((size_t)W * H);  // -1 no terrain, 0 unsupported, 1 supported
    std::vector<const Platform*> occupied((size_t)W * H, nullptr);
    for (size_t i = 0; i < supported.size(); i++) supported[i] = world.cells[i] ? 0 : -1;
    std::set<Platform> overlapping, oob;
    for (auto& kv : platforms_) {
        const Platform& plat = kv.second;
        Dims d = plat.dims();
        for (int oy = 0; oy < d.h; oy++)
            for (int ox = 0; ox < d.w; ox++) {
                int x = plat.point.x + ox, y = plat.point.y + oy;
                if (!world.in_bounds(x, y)) { oob.insert(plat); continue; }
                size_t i = (size_t)y * W + x;
                if (occupied[i]) { overlapping.insert(plat); overlapping.insert(*occupied[i]); }
                else occupied[i] = &plat;
                if (supported[i] == 0) supported[i] = 1;
            }
    }
    // extend support TERRAIN_SUPPORT_DISTANCE-1 times by one 4-neighbour step
    for (int round = 0; round + 1 < TERRAIN_SUPPORT_DISTANCE; round++) {
        std::vector<size_t> grow;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                if (supported[(size_t)y * W + x] != 1) continue;
                const int nx[4] = {x + 1, x, x - 1, x}, ny[4] = {y, y + 1, y, y - 1};
                for (int k = 0; k < 4; k++)
                    if (world.in_bounds(nx[k], ny[k])) grow.push_back((size_t)ny[k] * W + nx[k]);
            }
        for (size_t i : grow)
            if (supported[i] == 0) supported[i] = 1;
    }
    ValidationResult r;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            if (supported[(size_t)y * W + x] == 0) r.unsupported_terrain.push_back(Point{x, y});
    r.overlapping_platforms.assign(overlapping.begin(), overlapping.end());
    r.out_of_bounds_platforms.assign(oob.begin(), oob.end());
    return r;
}

void PlatformLayout::run_trivial_optimization(const WorldGrid& world) {  // platform_layout.rs:151-172
    for (auto it = platforms_.begin(); it != platforms_.end();) {
        Dims d = it->second.dims();
        bool any = false;
        for (int oy = 0; oy < d.h && !any; oy++)
            for (int ox = 0; ox < d.w && !any; ox++)
                any = world.terrain(it->first.x + ox, it->first.y + oy);
        if (any) ++it;
        else it = platforms_.erase(it);
    }
}

long PlatformLayout::total_weight(const std::map<Dims, long>& weights) const {  // platform_layout.rs:174-183
    long sum = 0;
    for (auto& kv : platforms_)
        for (auto& w : weights)
            if (w.first == kv.second.def || w.first.strictly_within(kv.second.def)) sum += w.second;
    return sum;
}

}  // namespace tbs
