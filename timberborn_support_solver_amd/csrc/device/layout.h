// layout.h — data layout in HBM shared by the host driver and the HIP kernels.
//
// One *worker* = one 64-lane wavefront running a complete CDCL search (or a
// scripted BCP) on its own copy of the mutable solver state.  Everything a
// worker mutates lives in ONE contiguous slab of HBM (`slab_bytes` per
// worker); everything immutable exists once per GPU and is read by all workers,
// so it is served from L2 / Infinity Cache while the private slabs stream from
// HBM.  Immutable, shared:
//   * binary clauses   as an implication CSR  (literal p true -> implied literals)
//   * ternary clauses  as a CSR of literal PAIRS (p true, i.e. ~p false -> the
//                      other two literals of each clause containing ~p); they need
//                      no watches, no writes and no private memory at all
//   * long clauses (>= 4 literals): literals + offsets; each worker keeps only the
//     two watched literals per clause and its watch lists privately.
//
// Literal encoding on the device: lit = 2*var + neg, var 0-based.
#pragma once
#include <stdint.h>

#define MS_WAVE 64
#define MS_LDS_RING 256           // per-wave propagation queue window in LDS (entries)
#define MS_CLAIM_SLOTS 256        // per-wave implication claim set in LDS (<= 192 candidates per commit)
#define MS_OVERFLOW_CAP 192       // watcher pushes that found their list full, per chunk
#define MS_LBDQ 50                // Glucose restart window
#define MS_MAX_GROUPS 32          // queue literals propagated per step (lane groups per wave)
#define MS_SPLIT_MAX 8            // decisions a worker offers per slice for splitting its cube
// Learnt-clause exchange between the workers of one GPU: fixed 128-byte records
//   word 0 = size | lbd << 6 | producer << 14,  words 1..31 = literals
// A worker appends the short / low-LBD clauses it learns to its private export buffer; between two
// slices ms_share_collect_kernel moves the new ones (deduplicated) into one global ring of records,
// and every worker attaches the records it has not seen yet the next time it stands at level 0.
#define MS_SHARE_REC 32           // int32 words per record
#define MS_SHARE_MAXLEN 31        // longest clause that is exchanged
#define MS_EXPORT_RECS 64         // per-worker export buffer (records per slice; more are dropped)

// lit_value() results
#define MS_VAL_TRUE 0
#define MS_VAL_FALSE 1
#define MS_VAL_UNDEF 2
// stored assignment per variable (a byte in the slab, 2 bits in LDS): bit1 = assigned, bit0 = sign
#define MS_ASG_UNDEF 0
#define MS_ASG_TRUE 2
#define MS_ASG_FALSE 3

// reason[] encoding
#define MS_REASON_NONE (-1)
// binary reason: the clause (x | other) implied x; stored as -2 - other
#define MS_REASON_BIN(other) (-2 - (other))
// ternary reason: entry e of the ternary CSR implied one literal of its pair
#define MS_TERN_BASE (-(1 << 30))
#define MS_REASON_TERN(e) (MS_TERN_BASE - (int)(e))
#define MS_IS_TERN_REASON(r) ((r) <= MS_TERN_BASE)
#define MS_TERN_REASON_ENTRY(r) (MS_TERN_BASE - (r))
#define MS_IS_BIN_REASON(r) ((r) <= -2 && (r) > MS_TERN_BASE)
#define MS_BIN_REASON_LIT(r) (-2 - (r))
#define MS_MAX_VARS (1u << 24)

enum {
    MS_ST_RUNNING = 0,
    MS_ST_SAT = 10,
    MS_ST_UNSAT = 20,          // the formula itself is refuted (conflict without any decision)
    MS_ST_REFUTED = 21,        // this worker's cube (assumption list) is refuted
    MS_ST_PARKED = 120,        // another worker already decided this instance
    MS_ST_ERR_POOL = -1,       // watch pool exhausted
    MS_ST_ERR_LEARNT = -2,     // learnt clause store exhausted
    MS_ST_ERR_INTERNAL = -3,
};

struct ms_int2 { int32_t x, y; };
// Per-literal list header of the shared CSRs: ONE 16-byte load per dequeued literal.
struct MsLitHdr { uint32_t bin_off, bin_n, tern_off, tern_n; };
// Per-literal header, private: the watch list (`size` is the atomic push counter) AND a copy of the literal's
// immutable binary / ternary list header, so that a dequeued literal costs ONE 32-byte access to one line
// instead of a private and a shared one (+16 B x 2 x n_vars per worker: 3 MB of 37 at rect 64x64).
struct MsWatchHdr { uint32_t base, size, cap, pad; uint32_t bin_off, bin_n, tern_off, tern_n; };
// Where literal t's header is in `whdr`: polarity-major (all positive literals, then all negative ones), so that the
// headers of CONSECUTIVE variables of one polarity are neighbours, two per 64-byte line.  A platform placed on the grid
// falsifies the whole run of placement variables of every tile it covers at once (overlap clauses + the size chain): in
// literal order (2v, 2v+1) each of those dequeued literals had a line of its own, half of it the header of the opposite
// polarity that is not touched then.
#ifndef MS_HDR_POLARITY_MAJOR
#define MS_HDR_POLARITY_MAJOR 1
#endif
#if MS_HDR_POLARITY_MAJOR
#define MS_HIDX(t, nv) ((((uint32_t)(t)) & 1u) * (uint32_t)(nv) + (((uint32_t)(t)) >> 1))
#else
#define MS_HIDX(t, nv) ((uint32_t)(t))
#endif
// Per-variable record: everything backtracking and analysis need to know about an assigned variable sits in ONE
// 16-byte slot, written by ONE store when the variable is assigned: its level, its reason and - for a long / learnt
// reason - where that clause's literals are (start, size; size 0 = look the clause record up), so that conflict
// analysis goes from a literal to its reason's literals in one dependent round trip.  phase = its polarity now, which
// is what phase saving would record at unassignment, so backtracking writes nothing here.  seen: analysis mark of
// the builds that keep the assignment in HBM (the LDS builds keep a bitmap).
struct MsVarRec { int32_t level, reason; uint32_t start; uint16_t size; uint8_t phase, seen; };
// Long / learnt clause header: literals start 16-byte aligned (4 literals) so that a lane reads 4 at a time.
struct MsClauseHdr { uint32_t start, size; };
// Per worker and clause (original long clauses first, then learnt ones): the two watched literals AND where the
// literals are - ONE 16-byte access when a watcher's blocker is not true (was: a private pair + a header elsewhere).
struct MsClauseRec { int32_t w0, w1; uint32_t start, size; };

// Immutable, one per GPU.
struct MsShared {
    uint32_t n_vars;
    uint32_t n_orig;               // long (>= 4 literal) original clauses, cref 0..n_orig-1
    const int32_t* cl_lits;        // literals of the long original clauses (each clause 16-byte aligned)
    const int32_t* bin_lits;       // implied literals q  (clause  ~p | q) of p being TRUE
    const ms_int2* tern_pairs;     // the other two literals (b, c) of clause (~p | b | c)
    const int32_t* tern_owner;     // per entry: the literal p whose list it is in (conflict analysis)
};

// Byte offsets of the private arrays inside a worker slab.
struct MsLayout {
    uint64_t slab_bytes;
    uint64_t state;       // MsState
    uint64_t val;         // uint8 [n_vars, padded to 16]  assignment, one byte per variable (MS_ASG_*), plain loads / stores;
                          //        the LDS builds stage it as 2 bits per variable for the slice
    uint64_t vrec;        // MsVarRec [n_vars]  level, reason (+ its literal range), saved phase, seen mark
    uint64_t vm_pos;      // int32  [n_vars]  position of the variable's live entry in vm_order
    uint64_t best;        // uint8  [n_vars]  polarity of the variable in the longest conflict-free assignment seen (255: none)
    uint64_t trail;       // int32  [n_vars]
    uint64_t trail_lim;   // int32  [n_vars+1]
    uint64_t vm_order;    // int32  [vm_cap]   move-to-front queue as an append-only array
    uint64_t wl;          // MsClauseRec [n_orig + learnt_cap]  watched pair + literal range per clause
    uint64_t whdr;        // MsWatchHdr [2*n_vars]  the literal's watch list (slot in pool, size, capacity) + where its
                          //        binary / ternary lists are in the shared CSRs
    uint64_t pool;        // int4   [pool_cap]  watcher = (cref, blocker, start, size of the clause's literals); cref < 0 = tombstone
    uint64_t lc_lbd;      // uint32 [learnt_cap]  lbd | used<<31
    uint64_t lc_lits;     // int32  [learnt_lit_cap]  (each clause 16-byte aligned)
    uint64_t learnt_buf;  // int32  [n_vars+1]   clause under construction
    uint64_t toclear;     // int32  [n_vars+1]   vars touched by analysis
    uint64_t lvl_stamp;   // uint32 [n_vars+2]   LBD computation
    uint64_t remap;       // uint32 [learnt_cap] reduceDB old->new
    uint64_t overflow;    // int32  [3*MS_OVERFLOW_CAP]  (list, cref, blocker)
    uint64_t assumps;     // int32  [assump_cap]
    uint64_t script;      // int32  [script_cap]  decisions for propagate_batch
    uint64_t exp;         // int32  [MS_EXPORT_RECS*MS_SHARE_REC]  clauses learnt in this slice that other workers get
    uint32_t n_vars, n_orig, learnt_cap, learnt_lit_cap, pool_cap, vm_cap, assump_cap, script_cap;
};

// Per-worker scalar state (lives at slab + layout.state); 8-byte aligned.
struct MsState {
    int32_t status;            // MS_ST_*
    int32_t trail_n, qhead, n_levels;
    int32_t n_assumps;
    int32_t n_script;
    int32_t restart_req;       // host assigned a new cube: backtrack to level 0 before continuing
    int32_t n_split;           // split[0..n_split): this worker's decisions right above its cube, oldest first
    int32_t split[MS_SPLIT_MAX];
    // decision queue
    int32_t vm_end, vm_search;
    // learnt store
    uint32_t n_learnts, lc_lits_n, pool_top;
    uint32_t exp_n;            // records in the export buffer
    // restarts (Glucose K=0.8 on LBD window, R=1.4 trail blocking)
    uint32_t lbdq[MS_LBDQ];
    uint32_t lbdq_n, lbdq_i;
    uint64_t lbdq_sum, lbd_total;
    double trail_avg;          // exponential moving average stands in for Glucose's 5000-entry queue
    uint64_t next_reduce;
    uint32_t lvl_stamp_ctr, pad2;
    uint64_t rng;
    // counters
    uint64_t propagations, decisions, conflicts, restarts, reduce_dbs;
    uint64_t n_watch, n_cl_lit, n_move, n_enq;
    uint64_t learnt_total, learnt_lits_total;
    uint64_t slice_cycles;
    uint64_t n_steps;          // BCP steps (each propagates up to MS_MAX_GROUPS literals)
    uint64_t n_redo;           // literals re-queued because two groups met in one clause
    uint64_t prof[16];         // per-phase cycle totals and counts (profiling build only)
    // clause exchange
    uint64_t share_pos;        // records of the global ring this worker has looked at
    uint64_t n_exported, n_imported, n_imported_units;
    uint64_t last_import_confl;
    // best-phase rephasing
    int32_t best_trail; uint32_t n_rephase;
    uint64_t next_rephase;
    // vivification
    uint64_t next_vivify, n_vivified, n_viv_lits;
};

// Launch parameters of one slice.
struct MsParams {
    uint32_t n_workers;
    uint32_t slice_conflicts;      // stop the slice after this many conflicts per worker
    uint64_t slice_props;          // ... or this many propagations (0 = unlimited)
    uint64_t slice_ticks;          // ... or this much wall time, in 10 ns ticks of the 100 MHz counter (0 = unlimited)
    const volatile int32_t* stop_flag;  // pinned host int: nonzero -> leave the slice early
    int32_t stop_on_any;           // leave when any worker has finished (any_done)
    int32_t max_groups;            // 1..MS_MAX_GROUPS queue literals per BCP step
    int32_t* any_done;             // device int, set when a worker reaches SAT/UNSAT
    int32_t done_on_refuted;       // a refuted cube also raises any_done (portfolio mode: it decides the instance)
    int32_t pad;
    uint32_t reduce_first, reduce_inc;
    int32_t* proof_buf;            // optional DRUP log: proof_cap words per worker; learnt clauses as internal literals, -1 terminated
    uint32_t* proof_len;           // per worker: words used / wanted since the last drain (overflow: proof_len > proof_cap)
    uint32_t proof_cap, pad2;
    // clause exchange (share_pool == nullptr: off)
    const int32_t* share_pool;     // ring of share_slots records
    const unsigned long long* share_n;  // records ever appended (constant during a slice)
    uint32_t share_slots;
    uint32_t share_max_lbd;        // clauses with lbd <= this (or size <= 2) and size <= share_max_len are exported
    uint32_t share_interval;       // a worker with unseen records restarts to import them after this many conflicts
    uint32_t share_max_len;
    int32_t rephase;               // 0: off, 1: every worker rephases to its best assignment, 2: workers with an odd index
    int32_t restart_k_pct;         // Glucose restart factor K in percent (0 = 100)
    int32_t restart_k2_pct;        // > 0: workers with an odd index use this K
    int32_t vivify, pad5;          // learnt clauses vivified per pass (0 / -1 = off)
    int32_t import_pct;            // share (percent) of the exchanged clauses of >= 3 literals a worker attaches (0 = 50)
};
