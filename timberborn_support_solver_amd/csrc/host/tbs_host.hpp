// tbs_host.hpp — host side of the solve path, C++ mirror of the reference's
// operator interface for it (the reference is Rust; no Rust toolchain exists in
// this build environment).  Names and argument meaning follow
// the reference:
//
//   Encoding::encode / vars / with_limits      src/encoder.rs:435-667
//   SatInstance::into_cnf  (rustsat, [ext])    crates/repl/src/main.rs:293
//   PlatformLimits                             src/encoder/platform_limits.rs:6-26
//   PlatformLayout::{from_assignment,platform_count,platform_stats,validate}
//                                              src/encoder/platform_layout.rs:26-149
//   WorldGrid (rows of 'X'/' ')                src/world.rs:49-79
//   PLATFORMS_DEFAULT                          src/platform.rs:23-32
//   TERRAIN_SUPPORT_DISTANCE = 4               src/lib.rs:12
//
// Nothing here touches the GPU; the solver itself is behind include/mi355sat.h.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace tbs {

constexpr int TERRAIN_SUPPORT_DISTANCE = 4;  // src/lib.rs:12

struct Dims {
    int w = 0, h = 0;
    bool operator==(const Dims& o) const { return w == o.w && h == o.h; }
    bool operator!=(const Dims& o) const { return !(*this == o); }
    // total order used only to make variable numbering deterministic
    bool operator<(const Dims& o) const { return w != o.w ? w < o.w : h < o.h; }
    Dims flipped() const { return Dims{h, w}; }
    bool contains_point(int x, int y) const { return x >= 0 && x < w && y >= 0 && y < h; }
    // strict containment order of src/math/dimensions.rs:74-114 (no empty dims here)
    bool strictly_within(const Dims& o) const { return w <= o.w && h <= o.h && !(*this == o); }
    bool rectangular() const { return w != h; }  // src/platform.rs:52-54
};

struct Point {
    int x = 0, y = 0;
    bool operator==(const Point& o) const { return x == o.x && y == o.y; }
    bool operator<(const Point& o) const { return y != o.y ? y < o.y : x < o.x; }
};

struct Platform {  // src/platform.rs:63-69
    Point point;
    Dims def;      // definition dims (unrotated)
    bool rotated = false;
    Dims dims() const { return rotated ? def.flipped() : def; }
    bool operator<(const Platform& o) const {
        if (!(point == o.point)) return point < o.point;
        if (def != o.def) return def < o.def;
        return rotated < o.rotated;
    }
};

// src/world.rs: row-major bool grid, rows left aligned, short rows padded false
struct WorldGrid {
    int width = 0, height = 0;
    std::vector<uint8_t> cells;  // 1 = terrain
    bool in_bounds(int x, int y) const { return x >= 0 && x < width && y >= 0 && y < height; }
    bool terrain(int x, int y) const { return in_bounds(x, y) && cells[(size_t)y * width + x]; }
    static WorldGrid rect(int w, int h);
    // rows of 'X' / ' ' only (world.rs:57-60); throws std::runtime_error otherwise
    static WorldGrid from_rows(const std::vector<std::string>& rows);
    // minimal reader for `[world] grid = [ "..", ... ]` project files (test/*.toml)
    static WorldGrid from_toml_file(const std::string& path);
};

std::vector<Dims> platforms_default();  // src/platform.rs:23-32

// A plain CNF in CSR form; literals are DIMACS (+v/-v, v>=1).
struct Cnf {
    std::vector<int32_t> lits;
    std::vector<uint64_t> offsets{0};
    uint32_t n_vars = 0;
    size_t n_clauses() const { return offsets.size() - 1; }
    void add(const int32_t* l, size_t n) {
        lits.insert(lits.end(), l, l + n);
        offsets.push_back(lits.size());
        for (size_t i = 0; i < n; i++) {
            uint32_t v = (uint32_t)(l[i] < 0 ? -l[i] : l[i]);
            if (v > n_vars) n_vars = v;
        }
    }
    void add(std::initializer_list<int32_t> l) { add(l.begin(), l.size()); }
    void add(const std::vector<int32_t>& l) { add(l.data(), l.size()); }
};

// src/encoder/platform_limits.rs
struct PlatformLimits {
    std::map<Dims, size_t> card_limits;   // key = platform def dims
    std::map<Dims, long> weights;
    bool has_weight_limit = false;
    long weight_limit = 0;
};

struct EncodedItem {  // encoder.rs:170-174
    bool is_platform;
    Point point;
    Dims dims;   // platform
    int layer;   // terrain
};

// Clause-family tags, only for statistics / tests (SURVEY Appendix A breakdown)
enum Family { F_DAG_IMPL, F_DAG_PAIR, F_COVERAGE, F_TERRAIN_LAYER, F_TOP_UNIT,
              F_OVERLAP_1X1, F_OVERLAP_CROSS, F_OOB, F_COUNT };

// Cardinality side of a SatInstance, kept symbolic until into_cnf()
struct CardUb { std::vector<int32_t> lits; size_t bound; };
struct PbUb { std::vector<std::pair<int32_t, long>> terms; long bound; };

struct SatInstance {  // the part of rustsat::instances::SatInstance the path uses
    Cnf cnf;                      // plain clauses
    std::vector<CardUb> cards;    // add_card_constr(CardConstraint::new_ub)
    std::vector<PbUb> pbs;        // add_pb_constr(PbConstraint::new_ub)
    uint32_t n_vars = 0;
    int32_t new_var() { return (int32_t)++n_vars; }
    // into_cnf: expands every cardinality / PB constraint (totalizer /
    // generalized totalizer).  If `out_card_outputs` is given it receives, per
    // CardUb, the totalizer root outputs o_1..o_m (o_j = "at least j inputs
    // true"), which the sharded sweep uses as per-k assumption literals.
    Cnf into_cnf(std::vector<std::vector<int32_t>>* out_card_outputs = nullptr) const;
};

// Appends clauses of an upper-bound totalizer over `inputs`, truncated at
// `max_out` outputs, allocating aux vars from n_vars.  Returns root outputs
// o_1..o_m, m = min(inputs.size(), max_out).  Only the sum>=j => o_j direction
// is encoded (enough for upper bounds).
std::vector<int32_t> totalizer_ub(Cnf& cnf, uint32_t& n_vars, const std::vector<int32_t>& inputs,
                                  size_t max_out);

class Encoding {  // src/encoder.rs:428-668
public:
    // Encoding::encode(platform_defs, terrain)
    static Encoding encode(const std::vector<Dims>& platform_defs, const WorldGrid& terrain);

    // Encoding::with_limits(&limits) -> SatInstance
    SatInstance with_limits(const PlatformLimits& limits) const;

    // EncodingVars
    const std::vector<Dims>& platform_dims() const { return dims_; }           // sorted, incl. rotations
    int32_t var_for_dims_at(int x, int y, Dims d) const;                        // 0 if none
    int32_t terrain_var(int x, int y, int layer) const;                         // 0 if none
    bool var_to_platform(int32_t var, Platform* out) const;                     // encoder.rs:232-249
    const EncodedItem* item(int32_t var) const;
    std::string lit_readable_name(int32_t lit) const;                           // encoder.rs:251-273
    std::vector<int32_t> iter_dims_vars(Dims d) const;                          // encoder.rs:218-222

    const SatInstance& instance() const { return instance_; }
    const WorldGrid& grid() const { return grid_; }
    const std::vector<Dims>& platform_defs() const { return defs_; }
    const size_t* family_counts() const { return fam_; }

    // EncodingDag edge sets (exposed for tests; encoder.rs:349-425)
    const std::vector<std::pair<Dims, Dims>>& platform_edges_reduced() const { return plat_edges_; }
    const std::vector<std::pair<Point, Dims>>& point_platform_edges_reduced() const { return point_edges_; }

private:
    WorldGrid grid_;
    std::vector<Dims> defs_;
    std::vector<Dims> dims_;
    std::vector<int32_t> tile_base_;    // first var of each tile
    std::vector<EncodedItem> items_;    // index var-1
    SatInstance instance_;
    size_t fam_[F_COUNT] = {0};
    std::vector<std::pair<Dims, Dims>> plat_edges_;          // (smaller, larger)
    std::vector<std::pair<Point, Dims>> point_edges_;        // (offset, minimal containing dims)
    struct PairClause { Dims a, b; std::vector<Dims> succ; };
    std::vector<PairClause> pair_clauses_;
    int dim_index(Dims d) const;
    void build_dag();
};

struct ValidationResult {  // platform_layout.rs:186-199
    std::vector<Point> unsupported_terrain;
    std::vector<Platform> overlapping_platforms;
    std::vector<Platform> out_of_bounds_platforms;
    bool is_valid() const {
        return unsupported_terrain.empty() && overlapping_platforms.empty() &&
               out_of_bounds_platforms.empty();
    }
};

class PlatformLayout {  // src/encoder/platform_layout.rs
public:
    // model[v-1] = 1 true, -1 false, 0 don't-care (Assignment)
    static PlatformLayout from_assignment(const int8_t* model, size_t n_vars, const Encoding& enc);
    static PlatformLayout from_platforms(const std::vector<Platform>& p);
    const std::map<Point, Platform>& platforms() const { return platforms_; }
    size_t platform_count() const { return platforms_.size(); }
    std::map<Dims, size_t> platform_stats() const;
    ValidationResult validate(const WorldGrid& world) const;
    void run_trivial_optimization(const WorldGrid& world);
    long total_weight(const std::map<Dims, long>& weights) const;

private:
    std::map<Point, Platform> platforms_;
};

}  // namespace tbs
