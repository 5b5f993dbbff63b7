/*
 * mi355sat.h — C ABI of the MI355X-native SAT solve loop (libmi355sat.so).
 *
 * This is the drop-in boundary behind timberborn_support_solver's solver
 * backend.  The reference is generic over `S: Solve + Interrupt (+ Default +
 * SolveStats) + Send + 'static` and instantiates it with
 * `rustsat_glucose::simp::Glucose`:
 *
 *   crates/repl/src/solver_runner.rs:8-20   run_solver<S>: add_cnf, interrupter, solve
 *   crates/repl/src/main.rs:17,295          GlucoseSimp::default()
 *   crates/repl/src/main.rs:329,363         full_solution(), stats()
 *   crates/gui/src/solver_backend.rs:69-97  S::default(), add_cnf, interrupter, solve
 *   crates/gui/src/main.rs:2,26             App::<GlucoseSimp>
 *
 * rustsat-glucose talks to its C++ solver through an IPASIR-shaped C API
 * (init / add / solve / val / interrupt / release; solve returns 10/20/0).
 * The functions below have the same shape so that a Rust `Solve` impl over
 * this library is a thin clone of that wrapper.
 *
 * Conventions
 *   - literals are IPASIR/DIMACS: +v / -v, v >= 1; 0 terminates a clause in
 *     mi355sat_add().  (rustsat `Lit` = (idx<<1)|neg with 0-based idx maps to
 *     ±(idx+1).)
 *   - no exceptions cross the ABI; errors are negative return codes and
 *     mi355sat_last_error() gives the text.
 *   - a handle may be moved between OS threads between calls
 *     (solver_runner.rs:15 moves the solver into tokio's blocking pool); every
 *     entry point binds its device itself.  The ONLY function that may run
 *     concurrently with another call on the same handle is
 *     mi355sat_interrupt() (main.rs:310-317 calls it from another task while
 *     solve() runs).
 *   - clause memory is copied on add; the caller may free it immediately
 *     (add_cnf consumes its argument, solver_runner.rs:12).
 *   - there is NO CPU fallback: if no HIP device is usable, mi355sat_new()
 *     returns NULL and mi355sat_last_error(NULL) says why.
 */
#ifndef MI355SAT_H
#define MI355SAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355SAT_SAT 10
#define MI355SAT_UNSAT 20
#define MI355SAT_INTERRUPTED 0
#define MI355SAT_ERR_OOM (-1)
#define MI355SAT_ERR_HIP (-2)
#define MI355SAT_ERR_STATE (-3)
#define MI355SAT_ERR_ARG (-4)

typedef struct mi355sat mi355sat;

/* Options for mi355sat_new(); zero-initialise and set what you need.
 * A NULL pointer means all defaults. */
typedef struct mi355sat_opts {
    int32_t device;            /* HIP device ordinal; -1 = current device (default 0) */
    int32_t workers;           /* concurrent search workers (one wavefront each); 0 = default: 4096 (16 per CU) above
                                  100k clauses, 1024 above 20k, else 256 */
    int64_t conflict_budget;   /* per-solve conflict limit summed over workers; 0 = none.
                                  Exhausted budget -> MI355SAT_INTERRUPTED */
    int32_t slice_conflicts;   /* conflicts per worker per kernel launch; 0 = default */
    uint64_t seed;             /* diversification seed (phases / decision order of workers > 0) */
    int32_t verbose;           /* 0 quiet, 1 progress on stderr */
    int32_t reduce_first;      /* conflicts before the first learnt-clause reduction; 0 = 2000 */
    int32_t reduce_inc;        /* growth of the reduction interval; 0 = 300 */
    int32_t lds_val;           /* assignment in LDS (2 bits/var): 0 auto, 1 force, -1 never */
    int32_t max_groups;        /* queue literals propagated per BCP step: 1..32; 0 = 32 */
    int32_t slice_ms;          /* wall-time bound of one kernel launch in ms (all workers stop together); 0 = default: 20 in a
                                  solve's first second of kernel time, then 50, after ten seconds 100 */
    int32_t cube_split;        /* 0 (default): portfolio, every worker of an instance searches the whole instance with its
                                  own decision order; 1: idle workers steal sub-cubes of running ones between slices
                                  (correct but, as measured in round 1, slower: DESIGN.md) */
    int32_t share;             /* learnt-clause exchange between the workers of one GPU (units, binaries and clauses of
                                  at most share_len literals with LBD <= share_lbd, passed on between kernel launches):
                                  0 = default (on), -1 = off.  Off automatically with one worker.  (A proof log keeps it on: every worker
                                  logs what it learns, mi355sat_set_proof_path.) */
    int32_t share_lbd;         /* 0 = 4 (measured: rect 24 k=8 1.0 vs 1.3 s, rect 16 1x1 k=14 14-18 vs 24 s with 2; 6-12 no better) */
    int32_t share_len;         /* longest exchanged clause, <= 31; 0 = 31 */
    int32_t share_interval;    /* > 0: a worker with unseen exchanged clauses restarts to attach them after this many of its
                                  own conflicts; 0 = default: only at its own (Glucose) restarts - forced restarts measured
                                  3-10x slower on the rect 24x24 ladder */
    int32_t var_order;         /* 0 = default: the device keeps the caller's variable numbering; 1 = it renumbers variables so
                                  that those meeting in clauses are neighbours (256-variable blobs grown breadth-first
                                  through the clauses).  Invisible at this interface: literals, models and proofs are
                                  always in the caller's numbering.  Measured on rect 64x64: no gain. */
    int32_t ramp;              /* 0 = default (on): the first 100 ms of kernel time of a solve run 256 workers (one per
                                  CU, each ~3x faster than one of 16), the next 300 ms 1024, then all - easy instances
                                  are decided by one worker's few hundred conflicts; -1 = the whole fleet at once */
    int32_t one_per_simd;      /* 0 = default: a launch of at most 1024 workers (one per SIMD) runs the build of the search
                                  kernel that owns the SIMD's whole register file (no spills, everything inlined), one of at
                                  most 2048 the 2-waves-per-SIMD build, larger ones the 4-waves build;
                                  -1 = always the 4-waves-per-SIMD build, 2 / 4 = at least the 2- / 4-waves build (A/B) */
    int32_t simp;              /* formula simplification before search (the reference's backend is `simp::Glucose`): 0 = default
                                  (on): equivalent-literal substitution, failed-literal probing on the device, subsumption and
                                  self-subsuming resolution on the device; 2 = the same plus bounded variable elimination
                                  (`SimpSolver::eliminate`: grow 0, resolvents of at most 20 literals; the variables of the
                                  assumptions are kept, eliminated ones get their values back when a model is read);
                                  -1 = only level-0 unit propagation.  Invisible at this interface: models, assumptions and
                                  proofs stay in the caller's variables. */
    int32_t phase_mix;         /* 0 = default: every worker starts with all saved phases FALSE (no platform anywhere);
                                  1 = portfolio of initial phases: a quarter of the workers start TRUE, a quarter at random */
    int32_t rephase;           /* rephasing to the best assignment (the polarities of the longest conflict-free assignment a worker
                                  has seen become its saved phases every 2000, 4000, 6000, ... conflicts, at a restart):
                                  0 = default: off (measured on rect 28x28 k = 12, a hard satisfiable bound: 4.3-6.3 s without,
                                  5.4-7.5 s with), 1 = every worker, 2 = every second worker */
    int32_t restart_k_pct;     /* Glucose's restart factor K in percent (restart when the LBD average of the last 50 conflicts times
                                  K exceeds the global average); 0 = 100 (round 3, with the sorted bump order: rect 28 k = 11 / rect 32
                                  k = 14 with 1024 workers 11.2 / 15.7 s at 100, 11.2-12.9 / 16.4-17.6 s at 90, 14.3 / 21.7 s at 80,
                                  20.0 / 34.5 s at 70, 11.7 / 17.8 s at 110; profiles/r03_m_knob_sweep*.log) */
    int32_t restart_k2_pct;    /* > 0: every second worker uses this K instead (a portfolio of restart policies); 0 = same K */
    int32_t import_pct;        /* share (percent) of the exchanged clauses of 3 and more literals each worker attaches (every worker
                                  another share; units and binaries always); 0 = default 50: a worker that attaches everything
                                  1023 others export spends its time on their clauses (measured, 1024 workers: rect 16 1x1 k = 14
                                  9.8-10.2 s at 50 %, 9.0-9.8 s at 25 %, 10.0-13.2 s at 100 %; rect 26 k = 10 46-56 s at 50 %,
                                  60-72 s at 100 %; rect 28 k = 11 within the run-to-run spread) */
    int32_t vivify;            /* vivification of learnt clauses: at a restart, every 400 conflicts, up to this many recent learnt
                                  clauses of LBD <= 6 (at most 64 literals) are re-derived literal by literal under unit
                                  propagation and replaced by the shorter clause that implies them (a RUP lemma, exported like a
                                  freshly learnt clause); > 0 = that many per pass; 0 = default: off (round 2 measured a quarter less
                                  time with 4 per pass; on top of round 3's recursive minimisation it is the other way round: rect 28
                                  k = 11 / rect 32 k = 14 10.5 / 15.9 s without, 12.9 / 17.6 s with, profiles/r03_m_knob_sweep2.log).  No effect in
                                  launches of more than 2048 workers: the full-fleet build leaves the code out */
    int32_t rebalance;         /* batched solves: 0 = default (on): workers of decided / withdrawn instances move to the open
                                  ones; -1 = they park */
    int32_t deterministic;     /* 0 = default: time-bounded slices (all workers stop together; what a worker does in a slice, and
                                  with it the whole trajectory, depends on timing); 1 = a reproducible mode for benchmarks and
                                  A/B comparisons: slices are bounded by conflicts per worker (slice_conflicts, default 200), no
                                  worker leaves a slice because another one finished, the exchanged clauses are collected in
                                  worker order by one thread, the whole fleet runs from the first slice (no ramp-up): two runs
                                  with the same options and seed make the same decisions and report the same counters.  Slower
                                  (a conflict-bounded slice waits for its slowest worker). */
} mi355sat_opts;

/* Counters.  n_deq .. n_enq are the five event counters of SURVEY.md §8(d)
 * from which algorithmic bytes are computed:
 *   bytes_alg = 12*n_deq + 9*n_watch + 5*n_cl_lit + 8*n_move + 13*n_enq      */
typedef struct mi355sat_stats_t {
    uint64_t propagations;     /* trail literals dequeued by BCP (== n_deq) */
    uint64_t decisions;
    uint64_t conflicts;
    uint64_t restarts;
    uint64_t learnts;          /* learnt clauses currently kept (sum over workers) */
    uint64_t learnt_literals;
    uint64_t reduce_dbs;
    uint64_t n_clauses;        /* clauses added by the caller (rustsat SolverStats.n_clauses) */
    uint64_t max_var;          /* highest variable index seen, 1-based (0 = none) */
    double   avg_clause_len;
    double   solve_seconds;    /* wall-clock inside solve()/solve_batch()/propagate_batch() */
    double   kernel_seconds;   /* device time of the search / BCP kernels (HIP events) */
    uint64_t kernel_launches;
    uint64_t n_deq, n_watch, n_cl_lit, n_move, n_enq;
    uint64_t n_sat, n_unsat, n_terminated; /* rustsat SolverStats: results returned so far */
    uint64_t bcp_steps;        /* BCP steps; each propagates up to 32 queue literals (one per lane group) */
    uint64_t bcp_requeued;     /* literals re-queued because two groups met in one clause */
    uint64_t shared_exported;  /* clauses workers offered to the exchange / clauses (and units) attached from it, */
    uint64_t shared_imported;  /* summed over workers */
    uint64_t shared_imported_units;
    uint64_t simp_units;           /* simplification before search: failed literals + necessary assignments found by probing */
    uint64_t simp_equivalences;    /* variables replaced by an equivalent literal */
    uint64_t simp_clauses_removed; /* clauses subsumed or strengthened */
    uint64_t workers;              /* search workers (wavefronts) of the last solve / batch / sweep: what was asked for, or
                                      what device memory had room for */
    uint64_t simp_eliminated;      /* variables resolved away before search (bounded variable elimination) */
} mi355sat_stats_t;

/* --- lifecycle (Default::default / Drop) --------------------------------- */
mi355sat* mi355sat_new(const mi355sat_opts* opts);
void mi355sat_free(mi355sat* s);
/* mi355sat_free parks the handle's worker slabs (the one large device allocation: up to 147 GiB, 2-5 s of
 * hipMalloc) for the next handle of the process on the same device - the refinement loop makes a fresh
 * solver per bound (crates/repl/src/main.rs:295).  This returns the parked buffer to the driver. */
void mi355sat_release_cached_memory(void);
const char* mi355sat_signature(void);                 /* Solve::signature */
/* sizeof(mi355sat_opts) (returned) and sizeof(mi355sat_stats_t) (*stats_size) of THIS build: a binding whose
 * mirror structs have other sizes was written against another header and must refuse to run. */
uint64_t mi355sat_abi_sizes(uint64_t* stats_size);
const char* mi355sat_last_error(const mi355sat* s);   /* s may be NULL (error of the last failed new) */

/* --- clause input (Solve::add_cnf / add_clause_ref) ---------------------- */
/* Bulk CSR: clause i = lits[offsets[i] .. offsets[i+1]), offsets has n_clauses+1 entries. */
int mi355sat_add_cnf(mi355sat* s, const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses);
/* IPASIR-style incremental add: literals, then 0 to close the clause. */
int mi355sat_add(mi355sat* s, int32_t lit_or_0);
/* Make sure variables 1..n exist even if they occur in no clause (rustsat reserve). */
int mi355sat_reserve(mi355sat* s, uint64_t n_vars);

/* --- solve (Solve::solve) ------------------------------------------------- */
/* Returns MI355SAT_SAT / MI355SAT_UNSAT / MI355SAT_INTERRUPTED or a negative error. */
int mi355sat_solve(mi355sat* s);

/* Batched solve under assumptions: instance i = formula AND assumption literals
 * assumps[assump_offsets[i] .. assump_offsets[i+1]).  This is what the sharded
 * decreasing-k sweep uses: the clause database (base CNF + one totalizer built
 * for k_max) is uploaded once and instance i assumes the negated totalizer
 * output for its own bound (solver_loop, crates/repl/src/main.rs:290-346, solves
 * one fresh CNF per k; the k's are independent).  results[i] receives
 * 10/20/0.  If stop_at_first != 0 the call returns as soon as one instance has a
 * verdict (others report 0).  Returns 0 or a negative error. */
int mi355sat_solve_batch(mi355sat* s, const int32_t* assumps, const uint64_t* assump_offsets,
                         uint64_t n_instances, int32_t* results, int stop_at_first);

/* The same batch, one kernel slice at a time (what bench.py times): begin uploads
 * the formula and creates the workers, each step runs every worker for
 * `slice_conflicts` conflicts (or to its verdict) and refreshes results / stats,
 * end fetches the models of SAT instances (mi355sat_model_of) and drops the batch. */
int mi355sat_sweep_begin(mi355sat* s, const int32_t* assumps, const uint64_t* assump_offsets, uint64_t n_instances);
int mi355sat_sweep_step(mi355sat* s, int32_t* results /* may be NULL */, uint64_t* n_decided /* may be NULL */);
/* Withdraw instances whose answer the caller no longer needs (in the decreasing-k sweep: every k above a
 * SAT one and every k below an UNSAT one is implied).  Their result stays 0, they count as decided, and
 * their workers move to the instances still open - as do the workers of every instance that gets its verdict. */
int mi355sat_sweep_drop(mi355sat* s, const uint64_t* instances, uint64_t n);
/* Priorities: open instance i gets about weights[i] / sum(weights of open instances) of the workers from the next
 * step on (workers move between open instances if need be; they keep their learnt clauses).  The decreasing-k loop
 * concentrates the fleet on the two bounds that decide it: the highest open one (a model there lowers the ceiling)
 * and the lowest open one (a refutation there raises the floor).  n must be the number of instances. */
int mi355sat_sweep_set_weights(mi355sat* s, const double* weights, uint64_t n);
/* Take withdrawn, still undecided instances up again: idle workers (parked, or of decided / withdrawn instances)
 * move to them.  The sharded sweep (one process per GPU) begins every rank with all bounds, withdraws
 * the other ranks' shards, and reopens the bounds still open anywhere once its own shard is decided. */
int mi355sat_sweep_reopen(mi355sat* s, const uint64_t* instances, uint64_t n);
/* Model of an instance that already reported SAT, while the sweep is still running (the loop needs the
 * layout's platform count to know which bounds it answers). */
int mi355sat_sweep_model_of(mi355sat* s, uint64_t instance, int8_t* out, uint64_t n_vars);
int mi355sat_sweep_end(mi355sat* s);

/* Batched unit propagation (BCP only, no search): instance i enqueues its
 * decision literals one decision level at a time, propagating to fixpoint after
 * each, and stops at the first conflict.  out_conflict[i] = 0 (fixpoint) or 1.
 * out_values (may be NULL) is n_instances rows of n_vars bytes:
 * 1 true, -1 false, 0 unassigned; for a conflicting instance the row holds the
 * assignment at the moment the conflict was found and is not comparable.
 * out_trail_len[i] (may be NULL) = number of assigned literals.
 * `repeat` > 1 re-runs the same batch that many times inside the call (device
 * state reset each time) for timing.  Returns 0 or a negative error. */
int mi355sat_propagate_batch(mi355sat* s, const int32_t* decisions, const uint64_t* decision_offsets,
                             uint64_t n_instances, int8_t* out_values, uint64_t n_vars,
                             int32_t* out_conflict, int32_t* out_trail_len, int32_t repeat);

/* --- model (Solve::lit_val / full_solution) ------------------------------ */
/* After SAT: returns +lit if lit is true, -lit if false, 0 if unknown var / no model. */
int32_t mi355sat_val(mi355sat* s, int32_t lit);
/* Bulk model: out[v-1] = 1 / -1 (0 for a variable the solver never saw). */
int mi355sat_model(mi355sat* s, int8_t* out, uint64_t n_vars);
/* Model of instance i of the last mi355sat_solve_batch(). */
int mi355sat_model_of(mi355sat* s, uint64_t instance, int8_t* out, uint64_t n_vars);

/* --- Interrupt::interrupter / InterruptSolver::interrupt ------------------ */
/* Async, thread-safe, idempotent: only sets a flag that solve() polls between
 * kernel launches (and the kernels poll from pinned host memory).  The solve it stops returns
 * MI355SAT_INTERRUPTED and consumes the flag; an interrupt that arrives while no solve is running
 * stops the next solve / batch / sweep at once (it is not lost) and is consumed by that one. */
void mi355sat_interrupt(mi355sat* s);

/* --- SolveStats::stats ----------------------------------------------------- */
int mi355sat_stats(const mi355sat* s, mi355sat_stats_t* out);

/* Test hook: the learnt clauses currently in the exchange ring of the last solve / batch / sweep, as DIMACS
 * literals, each clause 0-terminated.  out may be NULL to size the buffer; *n_records receives the number of
 * clauses.  Every one of them must be a consequence of the caller's formula alone (workers attach them under
 * any assumption set) - tests/ prove that with the oracle.  Returns 0, or MI355SAT_ERR_ARG if cap_words is
 * too small. */
int mi355sat_debug_share_ring(mi355sat* s, int32_t* out, uint64_t cap_words, uint64_t* n_records);

/* Clause exchange BETWEEN handles that search the SAME formula - the replicas of the sharded loop's last bounds, one
 * handle per GPU (every rank poses the reference's next bound, crates/repl/src/main.rs:292-295, with its own
 * seed).  Inside one handle the workers pass their short / low-LBD learnt clauses on through a ring on the device;
 * export hands out the records that entered the ring since the last export (never one that was imported), import
 * appends records from another handle so that this handle's workers attach them like each other's.  Wire format, in
 * the caller's variables: [lbd >= 1, DIMACS literals ..., 0] per clause.  Every record is a consequence of the
 * caller's formula alone (the tests prove it with the oracle), so it may be attached under any assumptions.  Both
 * calls are made between two mi355sat_sweep_step() of a running sweep; without one (or with the exchange off) they
 * do nothing.  export: out may be NULL to size the buffer (nothing is consumed); records that do not fit cap_words
 * wait for the next call. */
int mi355sat_share_export(mi355sat* s, int32_t* out, uint64_t cap_words, uint64_t* n_words, uint64_t* n_records);
int mi355sat_share_import(mi355sat* s, const int32_t* clauses, uint64_t n_words, uint64_t* n_records /* may be NULL */);

/* Optional DRUP proof (text, DIMACS literals, one lemma per line, the empty clause last) of the next plain
 * solve(), in its default configuration: all workers, clause exchange on.  Order of the lines: what the
 * simplification derived, then after every kernel slice the clauses each worker learnt in it; every line is a
 * RUP consequence of the lines before it (the exchange only hands on clauses of earlier slices).  Deletion lines
 * ("d ...") are written for the clauses a worker drops when it reduces its clause database - except those another
 * worker may hold a copy of (exchanged or imported ones).  Must be called before solve(); path NULL disables. */
int mi355sat_set_proof_path(mi355sat* s, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* MI355SAT_H */
