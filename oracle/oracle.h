/*
 * ORACLE (test infrastructure; never shipped, never on the product path).
 * C API of liboracle.so: CPU CDCL restatement (cdcl.c) and independent
 * checkers (check.c).  See the headers of those files for what each follows.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_solver ora_solver;

typedef struct ora_stats_t {
    uint64_t propagations, decisions, conflicts, restarts, learnts, learnt_literals, reduce_dbs;
    uint64_t n_clauses, max_var;
    double avg_clause_len, solve_seconds;
    uint64_t n_deq, n_watch, n_cl_lit, n_move, n_enq;
    uint64_t n_sat, n_unsat, n_terminated;
} ora_stats_t;

ora_solver* ora_new(void);
void ora_free(ora_solver* s);
void ora_reserve(ora_solver* s, uint64_t n_vars);
int ora_add_cnf(ora_solver* s, const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses);
/* 10 SAT, 20 UNSAT (under the assumptions), 0 budget exhausted / interrupted */
int ora_solve(ora_solver* s, const int32_t* assumps, int32_t n_assumps, int64_t conflict_budget);
int ora_model(ora_solver* s, int8_t* out, uint64_t n_vars);
void ora_stats(ora_solver* s, ora_stats_t* out);
void ora_interrupt(ora_solver* s);
void ora_enable_proof(ora_solver* s);
int64_t ora_proof_len(ora_solver* s);
const int32_t* ora_proof(ora_solver* s);

/* check.c ---------------------------------------------------------------- */
/* index of the first clause not satisfied by model (model[v-1] = 1/-1/0), or -1 */
int64_t ora_check_model(const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses,
                        const int8_t* model, uint64_t n_vars);
/* Unit propagation to fixpoint with plain occurrence lists (no watches):
 * formula units first, then each decision literal in turn (skipped if already
 * true, conflict if already false).  Returns 1 on conflict, 0 at fixpoint.
 * out_values[v-1] = 1/-1/0.  counters (may be NULL): [0] literals dequeued. */
int ora_bcp(const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses, uint64_t n_vars,
            const int32_t* decisions, uint64_t n_decisions, int8_t* out_values, int32_t* out_trail_len,
            uint64_t* counters);
/* Forward RUP check of a clausal proof (clauses 0-terminated; a clause
 * preceded by INT32_MIN is a deletion).  Returns 1 if every lemma is RUP and the
 * empty clause (or a conflict at top level) is derived, 0 otherwise. */
int ora_check_rup(const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses, uint64_t n_vars,
                  const int32_t* proof, int64_t proof_len);
#ifdef __cplusplus
}
#endif
#endif
