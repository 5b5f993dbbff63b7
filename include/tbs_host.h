/*
 * tbs_host.h — C view of the host side of the solve path (libtbs_host.so), for
 * bindings (ctypes in timberborn_support_solver_amd/, tests).  It exposes the
 * C++ mirror of the reference's encoder-side operators; the solver boundary
 * itself is include/mi355sat.h.
 *
 *   tbs_encode                 Encoding::encode            src/encoder.rs:435-613
 *   tbs_with_limits_into_cnf   Encoding::with_limits + SatInstance::into_cnf
 *                                                          src/encoder.rs:619-667, crates/repl/src/main.rs:292-293
 *   tbs_layout_from_model      PlatformLayout::from_assignment   src/encoder/platform_layout.rs:26-52
 *   tbs_layout_validate        PlatformLayout::validate          src/encoder/platform_layout.rs:85-149
 *   tbs_grid_from_toml         WorldGrid deserialisation         src/world.rs:49-79
 *   tbs_solver_loop            solver_loop                       crates/repl/src/main.rs:280-366
 *
 * Errors: functions returning pointers return NULL, functions returning int
 * return < 0; tbs_last_error() gives the message (thread-local).
 */
#ifndef TBS_HOST_H
#define TBS_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tbs_encoding tbs_encoding;
typedef struct tbs_cnf tbs_cnf;
typedef struct tbs_layout tbs_layout;

const char* tbs_last_error(void);

/* World grid from a project file; returns 0 and fills width/height; cells (row-major,
 * 1 = terrain) are copied into out_cells if it has room for cap bytes. */
int tbs_grid_from_toml(const char* path, int32_t* width, int32_t* height, uint8_t* out_cells, uint64_t cap);

/* defs_wh = n_defs pairs (w,h); n_defs == 0 means PLATFORMS_DEFAULT */
tbs_encoding* tbs_encode(const int32_t* defs_wh, int32_t n_defs, const uint8_t* cells, int32_t width,
                         int32_t height);
void tbs_encoding_free(tbs_encoding* e);
uint32_t tbs_encoding_n_vars(const tbs_encoding* e);
int32_t tbs_encoding_n_dims(const tbs_encoding* e);
int tbs_encoding_dims(const tbs_encoding* e, int32_t* out_wh);             /* n_dims pairs */
int tbs_encoding_family_counts(const tbs_encoding* e, uint64_t out[8]);    /* Family order of tbs_host.hpp */
int32_t tbs_encoding_platform_var(const tbs_encoding* e, int32_t x, int32_t y, int32_t w, int32_t h);
int32_t tbs_encoding_terrain_var(const tbs_encoding* e, int32_t x, int32_t y, int32_t layer);
/* kind: 0 unknown, 1 platform (a,b = dims w,h), 2 terrain (a = layer) */
int tbs_encoding_var_info(const tbs_encoding* e, int32_t var, int32_t* kind, int32_t* x, int32_t* y,
                          int32_t* a, int32_t* b);
int tbs_encoding_n_plat_edges(const tbs_encoding* e);
int tbs_encoding_plat_edges(const tbs_encoding* e, int32_t* out4);         /* (sw,sh,lw,lh) per edge */
int tbs_encoding_n_point_edges(const tbs_encoding* e);
int tbs_encoding_point_edges(const tbs_encoding* e, int32_t* out4);        /* (x,y,w,h) per edge */

tbs_cnf* tbs_encoding_base_cnf(const tbs_encoding* e);
/* limits: n_limits triples (w, h, k) = card_limits.  sweep != 0 keeps the
 * totalizer outputs (tbs_cnf_card_outputs) so that bounds k' <= k can be posed as
 * assumptions -outputs[k'] on the same CNF. */
tbs_cnf* tbs_with_limits_into_cnf(const tbs_encoding* e, const int64_t* limits_whk, int32_t n_limits,
                                  int32_t sweep);
/* The same with platform weights and a bound on the total weight (the GUI's loop,
 * crates/gui/src/app.rs:235-245): weights_whw = n_weights triples (w, h, weight >= 0). */
tbs_cnf* tbs_with_limits_weights_into_cnf(const tbs_encoding* e, const int64_t* limits_whk, int32_t n_limits,
                                          const int64_t* weights_whw, int32_t n_weights, int32_t has_weight_limit,
                                          int64_t weight_limit, int32_t sweep);
void tbs_cnf_free(tbs_cnf* c);
uint32_t tbs_cnf_n_vars(const tbs_cnf* c);
uint64_t tbs_cnf_n_clauses(const tbs_cnf* c);
uint64_t tbs_cnf_n_lits(const tbs_cnf* c);
const int32_t* tbs_cnf_lits(const tbs_cnf* c);
const uint64_t* tbs_cnf_offsets(const tbs_cnf* c);
uint64_t tbs_cnf_n_card_outputs(const tbs_cnf* c);      /* outputs of the first cardinality constraint */
const int32_t* tbs_cnf_card_outputs(const tbs_cnf* c);  /* o_1..o_m, o_j = at least j inputs true */

tbs_layout* tbs_layout_from_model(const tbs_encoding* e, const int8_t* model, uint64_t n_vars);
/* platforms5 = n rows of (x, y, def_w, def_h, rotated) */
tbs_layout* tbs_layout_from_platforms(const int32_t* platforms5, int32_t n);
void tbs_layout_free(tbs_layout* l);
int32_t tbs_layout_count(const tbs_layout* l);
int tbs_layout_platforms(const tbs_layout* l, int32_t* out5);
/* counts[0..2] = unsupported terrain tiles, overlapping platforms, out-of-bounds platforms */
int tbs_layout_validate(const tbs_layout* l, const uint8_t* cells, int32_t width, int32_t height,
                        int32_t counts[3]);
/* PlatformLayout::total_weight (platform_layout.rs:174-183); weights_whw = n triples (w, h, weight) */
int64_t tbs_layout_total_weight(const tbs_layout* l, const int64_t* weights_whw, int32_t n);
int tbs_layout_trivial_optimization(tbs_layout* l, const uint8_t* cells, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif
