// C view of tbs_host.hpp (include/tbs_host.h).
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../../include/tbs_host.h"
#include "tbs_host.hpp"

using namespace tbs;

struct tbs_encoding { Encoding enc; };
struct tbs_cnf { Cnf cnf; std::vector<int32_t> card_outputs; };
struct tbs_layout { PlatformLayout lay; };

static thread_local std::string g_err;
#define TBS_TRY try {
#define TBS_CATCH(ret) } catch (const std::exception& ex) { g_err = ex.what(); return ret; }

static WorldGrid make_grid(const uint8_t* cells, int32_t w, int32_t h) {
    if (w <= 0 || h <= 0 || !cells) throw std::runtime_error("bad grid");
    WorldGrid g;
    g.width = w;
    g.height = h;
    g.cells.assign(cells, cells + (size_t)w * h);
    for (auto& c : g.cells) c = c ? 1 : 0;
    return g;
}

extern "C" {

const char* tbs_last_error(void) { return g_err.c_str(); }

int tbs_grid_from_toml(const char* path, int32_t* width, int32_t* height, uint8_t* out_cells, uint64_t cap) {
    TBS_TRY
    WorldGrid g = WorldGrid::from_toml_file(path);
    *width = g.width;
    *height = g.height;
    if (out_cells && cap >= g.cells.size()) memcpy(out_cells, g.cells.data(), g.cells.size());
    return 0;
    TBS_CATCH(-1)
}

tbs_encoding* tbs_encode(const int32_t* defs_wh, int32_t n_defs, const uint8_t* cells, int32_t width,
                         int32_t height) {
    TBS_TRY
    std::vector<Dims> defs;
    if (n_defs == 0) defs = platforms_default();
    for (int i = 0; i < n_defs; i++) defs.push_back(Dims{defs_wh[2 * i], defs_wh[2 * i + 1]});
    return new tbs_encoding{Encoding::encode(defs, make_grid(cells, width, height))};
    TBS_CATCH(nullptr)
}
void tbs_encoding_free(tbs_encoding* e) { delete e; }
uint32_t tbs_encoding_n_vars(const tbs_encoding* e) { return e->enc.instance().n_vars; }
int32_t tbs_encoding_n_dims(const tbs_encoding* e) { return (int32_t)e->enc.platform_dims().size(); }
int tbs_encoding_dims(const tbs_encoding* e, int32_t* out_wh) {
    size_t i = 0;
    for (auto& d : e->enc.platform_dims()) { out_wh[i++] = d.w; out_wh[i++] = d.h; }
    return 0;
}
int tbs_encoding_family_counts(const tbs_encoding* e, uint64_t out[8]) {
    for (int i = 0; i < F_COUNT; i++) out[i] = e->enc.family_counts()[i];
    return 0;
}
int32_t tbs_encoding_platform_var(const tbs_encoding* e, int32_t x, int32_t y, int32_t w, int32_t h) {
    return e->enc.var_for_dims_at(x, y, Dims{w, h});
}
int32_t tbs_encoding_terrain_var(const tbs_encoding* e, int32_t x, int32_t y, int32_t layer) {
    return e->enc.terrain_var(x, y, layer);
}
int tbs_encoding_var_info(const tbs_encoding* e, int32_t var, int32_t* kind, int32_t* x, int32_t* y,
                          int32_t* a, int32_t* b) {
    const EncodedItem* it = e->enc.item(var);
    if (!it) { *kind = 0; return 0; }
    *x = it->point.x;
    *y = it->point.y;
    if (it->is_platform) { *kind = 1; *a = it->dims.w; *b = it->dims.h; }
    else { *kind = 2; *a = it->layer; *b = 0; }
    return 0;
}
int tbs_encoding_n_plat_edges(const tbs_encoding* e) { return (int)e->enc.platform_edges_reduced().size(); }
int tbs_encoding_plat_edges(const tbs_encoding* e, int32_t* o) {
    for (auto& ed : e->enc.platform_edges_reduced()) {
        *o++ = ed.first.w; *o++ = ed.first.h; *o++ = ed.second.w; *o++ = ed.second.h;
    }
    return 0;
}
int tbs_encoding_n_point_edges(const tbs_encoding* e) { return (int)e->enc.point_platform_edges_reduced().size(); }
int tbs_encoding_point_edges(const tbs_encoding* e, int32_t* o) {
    for (auto& ed : e->enc.point_platform_edges_reduced()) {
        *o++ = ed.first.x; *o++ = ed.first.y; *o++ = ed.second.w; *o++ = ed.second.h;
    }
    return 0;
}

tbs_cnf* tbs_encoding_base_cnf(const tbs_encoding* e) {
    TBS_TRY
    return new tbs_cnf{e->enc.instance().cnf, {}};
    TBS_CATCH(nullptr)
}

tbs_cnf* tbs_with_limits_into_cnf(const tbs_encoding* e, const int64_t* lim, int32_t n_limits, int32_t sweep) {
    return tbs_with_limits_weights_into_cnf(e, lim, n_limits, nullptr, 0, 0, 0, sweep);
}

tbs_cnf* tbs_with_limits_weights_into_cnf(const tbs_encoding* e, const int64_t* lim, int32_t n_limits, const int64_t* wts,
                                          int32_t n_weights, int32_t has_weight_limit, int64_t weight_limit, int32_t sweep) {
    TBS_TRY
    PlatformLimits limits;
    auto resolve = [&](Dims d) {   // dims -> platform def, crates/repl/src/main.rs:85-101
        for (auto& p : e->enc.platform_defs())
            if (p == d || p.flipped() == d) return p;
        throw std::runtime_error("no platform with dimensions `" + std::to_string(d.w) + "x" + std::to_string(d.h) + "` found");
    };
    for (int i = 0; i < n_weights; i++) limits.weights[resolve(Dims{(int)wts[3 * i], (int)wts[3 * i + 1]})] = (long)wts[3 * i + 2];
    limits.has_weight_limit = has_weight_limit != 0;
    limits.weight_limit = (long)weight_limit;
    for (int i = 0; i < n_limits; i++) {
        Dims d{(int)lim[3 * i], (int)lim[3 * i + 1]};
        if (lim[3 * i + 2] < 0) throw std::runtime_error("expected non-negative integer");
        // crates/repl/src/main.rs:85-101: unknown dims and duplicates are errors
        bool known = false;
        Dims def = d;
        for (auto& p : e->enc.platform_defs())
            if (p == d || p.flipped() == d) { known = true; def = p; }
        if (!known)
            throw std::runtime_error("no platform with dimensions `" + std::to_string(d.w) + "x" +
                                     std::to_string(d.h) + "` found");
        if (limits.card_limits.count(def))
            throw std::runtime_error("duplicate limit for `" + std::to_string(def.w) + "x" +
                                     std::to_string(def.h) + "`");
        limits.card_limits[def] = (size_t)lim[3 * i + 2];
    }
    SatInstance inst = e->enc.with_limits(limits);
    auto* out = new tbs_cnf;
    if (sweep) {
        std::vector<std::vector<int32_t>> outs;
        out->cnf = inst.into_cnf(&outs);
        if (!outs.empty()) out->card_outputs = outs[0];
    } else {
        out->cnf = inst.into_cnf();
    }
    return out;
    TBS_CATCH(nullptr)
}
void tbs_cnf_free(tbs_cnf* c) { delete c; }
uint32_t tbs_cnf_n_vars(const tbs_cnf* c) { return c->cnf.n_vars; }
uint64_t tbs_cnf_n_clauses(const tbs_cnf* c) { return c->cnf.n_clauses(); }
uint64_t tbs_cnf_n_lits(const tbs_cnf* c) { return c->cnf.lits.size(); }
const int32_t* tbs_cnf_lits(const tbs_cnf* c) { return c->cnf.lits.data(); }
const uint64_t* tbs_cnf_offsets(const tbs_cnf* c) { return c->cnf.offsets.data(); }
uint64_t tbs_cnf_n_card_outputs(const tbs_cnf* c) { return c->card_outputs.size(); }
const int32_t* tbs_cnf_card_outputs(const tbs_cnf* c) { return c->card_outputs.data(); }

tbs_layout* tbs_layout_from_model(const tbs_encoding* e, const int8_t* model, uint64_t n_vars) {
    TBS_TRY
    return new tbs_layout{PlatformLayout::from_assignment(model, n_vars, e->enc)};
    TBS_CATCH(nullptr)
}
tbs_layout* tbs_layout_from_platforms(const int32_t* p, int32_t n) {
    TBS_TRY
    std::vector<Platform> v;
    for (int i = 0; i < n; i++)
        v.push_back(Platform{Point{p[5 * i], p[5 * i + 1]}, Dims{p[5 * i + 2], p[5 * i + 3]}, p[5 * i + 4] != 0});
    return new tbs_layout{PlatformLayout::from_platforms(v)};
    TBS_CATCH(nullptr)
}
void tbs_layout_free(tbs_layout* l) { delete l; }
int32_t tbs_layout_count(const tbs_layout* l) { return (int32_t)l->lay.platform_count(); }
int tbs_layout_platforms(const tbs_layout* l, int32_t* o) {
    for (auto& kv : l->lay.platforms()) {
        *o++ = kv.second.point.x; *o++ = kv.second.point.y;
        *o++ = kv.second.def.w; *o++ = kv.second.def.h; *o++ = kv.second.rotated;
    }
    return 0;
}
int tbs_layout_validate(const tbs_layout* l, const uint8_t* cells, int32_t width, int32_t height,
                        int32_t counts[3]) {
    TBS_TRY
    ValidationResult r = l->lay.validate(make_grid(cells, width, height));
    counts[0] = (int32_t)r.unsupported_terrain.size();
    counts[1] = (int32_t)r.overlapping_platforms.size();
    counts[2] = (int32_t)r.out_of_bounds_platforms.size();
    return r.is_valid() ? 1 : 0;
    TBS_CATCH(-1)
}
int64_t tbs_layout_total_weight(const tbs_layout* l, const int64_t* wts, int32_t n) {
    std::map<Dims, long> w;
    for (int i = 0; i < n; i++) w[Dims{(int)wts[3 * i], (int)wts[3 * i + 1]}] = (long)wts[3 * i + 2];
    return l->lay.total_weight(w);
}
int tbs_layout_trivial_optimization(tbs_layout* l, const uint8_t* cells, int32_t width, int32_t height) {
    TBS_TRY
    l->lay.run_trivial_optimization(make_grid(cells, width, height));
    return 0;
    TBS_CATCH(-1)
}

}  // extern "C"
