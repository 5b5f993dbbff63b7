/*
 * ORACLE (test infrastructure; never shipped, never on the product path).
 *
 * check.c — solver-independent checkers, deliberately written WITHOUT watched
 * literals so that they share no mechanism with either the CPU CDCL (cdcl.c) or
 * the HIP kernels they are used to check:
 *   ora_check_model  a model is correct iff it satisfies every clause of the
 *                    CNF it was given (SURVEY §8c "self-certifying SAT")
 *   ora_bcp          unit propagation to fixpoint over plain occurrence lists;
 *                    the fixpoint of unit propagation is unique, and whether a
 *                    conflict is reachable does not depend on propagation order,
 *                    so it is comparable bit-for-bit with the batched HIP BCP
 *                    (BASELINE.json configs[1])
 *   ora_check_rup    forward reverse-unit-propagation check of a clausal
 *                    (DRUP) proof: pins UNSAT verdicts on small instances
 * These define "correct" for the solver boundary used at
 * crates/repl/src/solver_runner.rs:16 (solve) and crates/repl/src/main.rs:329
 * (full_solution); the reference itself has no test at that boundary.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

int64_t ora_check_model(const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses,
                        const int8_t* model, uint64_t n_vars) {
    for (uint64_t c = 0; c < n_clauses; c++) {
        int sat = 0;
        for (uint64_t k = offsets[c]; k < offsets[c + 1] && !sat; k++) {
            int32_t l = lits[k];
            uint64_t v = (uint64_t)(l < 0 ? -l : l);
            if (v == 0 || v > n_vars) continue;
            sat = (l > 0) ? (model[v - 1] > 0) : (model[v - 1] < 0);
        }
        if (!sat) return (int64_t)c;
    }
    return -1;
}

/* ---- occurrence-list propagation engine ------------------------------- */
typedef struct {
    uint64_t n_vars, n_clauses, cap_clauses;
    int32_t* cl_lits;   /* concatenated */
    uint64_t* cl_off;   /* n_clauses+1 */
    uint64_t lits_n, lits_cap;
    uint8_t* cl_dead;
    /* occurrence lists as linked lists so clauses can be appended (RUP lemmas) */
    int64_t* occ_head;  /* per literal index (2*(v-1)+neg) -> entry */
    int64_t* ent_next;  /* per literal occurrence */
    int64_t* ent_clause;
    uint64_t ent_n, ent_cap;
    int8_t* val;        /* per var 1/-1/0 */
    int32_t* trail;
    uint64_t trail_n, qhead;
    uint64_t deq;
} occ_engine;

static inline uint64_t lidx(int32_t l) { return l > 0 ? 2 * (uint64_t)(l - 1) : 2 * (uint64_t)(-l - 1) + 1; }
static inline int lval(const occ_engine* e, int32_t l) { int8_t v = e->val[(l < 0 ? -l : l) - 1]; return l > 0 ? v : -v; }

static void eng_init(occ_engine* e, uint64_t n_vars) {
    memset(e, 0, sizeof *e);
    e->n_vars = n_vars;
    e->occ_head = (int64_t*)malloc(sizeof(int64_t) * 2 * (n_vars + 1));
    for (uint64_t i = 0; i < 2 * (n_vars + 1); i++) e->occ_head[i] = -1;
    e->val = (int8_t*)calloc(n_vars + 1, 1);
    e->trail = (int32_t*)malloc(sizeof(int32_t) * (n_vars + 1));
    e->cl_off = (uint64_t*)malloc(sizeof(uint64_t) * 2);
    e->cl_off[0] = 0;
    e->cap_clauses = 1;
}
static void eng_free(occ_engine* e) {
    free(e->cl_lits); free(e->cl_off); free(e->cl_dead); free(e->occ_head); free(e->ent_next);
    free(e->ent_clause); free(e->val); free(e->trail);
}
static uint64_t eng_add(occ_engine* e, const int32_t* l, uint64_t n) {
    if (e->n_clauses + 1 >= e->cap_clauses) {
        e->cap_clauses = e->cap_clauses * 2 + 16;
        e->cl_off = (uint64_t*)realloc(e->cl_off, sizeof(uint64_t) * (e->cap_clauses + 1));
        e->cl_dead = (uint8_t*)realloc(e->cl_dead, e->cap_clauses);
    }
    if (e->lits_n + n > e->lits_cap) {
        e->lits_cap = (e->lits_n + n) * 2 + 64;
        e->cl_lits = (int32_t*)realloc(e->cl_lits, sizeof(int32_t) * e->lits_cap);
    }
    if (e->ent_n + n > e->ent_cap) {
        e->ent_cap = (e->ent_n + n) * 2 + 64;
        e->ent_next = (int64_t*)realloc(e->ent_next, sizeof(int64_t) * e->ent_cap);
        e->ent_clause = (int64_t*)realloc(e->ent_clause, sizeof(int64_t) * e->ent_cap);
    }
    uint64_t c = e->n_clauses++;
    e->cl_dead[c] = 0;
    for (uint64_t k = 0; k < n; k++) {
        e->cl_lits[e->lits_n + k] = l[k];
        uint64_t li = lidx(l[k]);
        e->ent_clause[e->ent_n] = (int64_t)c;
        e->ent_next[e->ent_n] = e->occ_head[li];
        e->occ_head[li] = (int64_t)e->ent_n++;
    }
    e->lits_n += n;
    e->cl_off[c + 1] = e->lits_n;
    return c;
}
/* examine clause c: returns 1 on conflict; enqueues the unit literal if any */
static int eng_examine(occ_engine* e, uint64_t c) {
    if (e->cl_dead[c]) return 0;
    int32_t unit = 0;
    int n_free = 0;
    for (uint64_t k = e->cl_off[c]; k < e->cl_off[c + 1]; k++) {
        int v = lval(e, e->cl_lits[k]);
        if (v > 0) return 0;
        if (v == 0) { if (unit != e->cl_lits[k]) { n_free++; unit = e->cl_lits[k]; } if (n_free > 1) return 0; }
    }
    if (n_free == 0) return 1;
    e->val[(unit < 0 ? -unit : unit) - 1] = unit > 0 ? 1 : -1;
    e->trail[e->trail_n++] = unit;
    return 0;
}
static int eng_propagate(occ_engine* e) {
    while (e->qhead < e->trail_n) {
        int32_t p = e->trail[e->qhead++];
        e->deq++;
        for (int64_t en = e->occ_head[lidx(-p)]; en >= 0; en = e->ent_next[en])
            if (eng_examine(e, (uint64_t)e->ent_clause[en])) return 1;
    }
    return 0;
}
static void eng_backtrack(occ_engine* e, uint64_t to) {
    while (e->trail_n > to) { int32_t l = e->trail[--e->trail_n]; e->val[(l < 0 ? -l : l) - 1] = 0; }
    e->qhead = to;
}
static int eng_assume(occ_engine* e, int32_t l) { /* 1 = conflict */
    int v = lval(e, l);
    if (v > 0) return 0;
    if (v < 0) return 1;
    e->val[(l < 0 ? -l : l) - 1] = l > 0 ? 1 : -1;
    e->trail[e->trail_n++] = l;
    return 0;
}

int ora_bcp(const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses, uint64_t n_vars,
            const int32_t* decisions, uint64_t n_decisions, int8_t* out_values, int32_t* out_trail_len,
            uint64_t* counters) {
    occ_engine e;
    eng_init(&e, n_vars);
    int confl = 0;
    for (uint64_t c = 0; c < n_clauses; c++) eng_add(&e, lits + offsets[c], offsets[c + 1] - offsets[c]);
    /* every clause is examined once: unit clauses seed the queue, an empty clause is a conflict */
    for (uint64_t c = 0; c < n_clauses && !confl; c++) confl = eng_examine(&e, c);
    if (!confl) confl = eng_propagate(&e);
    for (uint64_t d = 0; d < n_decisions && !confl; d++) {
        confl = eng_assume(&e, decisions[d]);
        if (!confl) confl = eng_propagate(&e);
    }
    if (out_values) memcpy(out_values, e.val, n_vars);
    if (out_trail_len) *out_trail_len = (int32_t)e.trail_n;
    if (counters) counters[0] = e.deq;
    eng_free(&e);
    return confl;
}

int ora_check_rup(const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses, uint64_t n_vars,
                  const int32_t* proof, int64_t proof_len) {
    occ_engine e;
    eng_init(&e, n_vars);
    int confl = 0, ok = 1, derived_empty = 0;
    for (uint64_t c = 0; c < n_clauses; c++) eng_add(&e, lits + offsets[c], offsets[c + 1] - offsets[c]);
    for (uint64_t c = 0; c < n_clauses && !confl; c++) confl = eng_examine(&e, c);
    if (!confl) confl = eng_propagate(&e);
    if (confl) derived_empty = 1;
    int64_t i = 0;
    while (i < proof_len && ok && !derived_empty) {
        int del = 0;
        if (proof[i] == INT32_MIN) { del = 1; i++; }
        int64_t st = i;
        while (i < proof_len && proof[i] != 0) i++;
        int64_t n = i - st;
        i++; /* skip 0 */
        if (del) {
            /* find one live clause with exactly this literal set (order-insensitive) and kill it;
             * literals it already implied at top level stay (they were derived validly) */
            if (n == 0) continue;
            for (int64_t en = e.occ_head[lidx(proof[st])]; en >= 0; en = e.ent_next[en]) {
                uint64_t c = (uint64_t)e.ent_clause[en];
                if (e.cl_dead[c] || (int64_t)(e.cl_off[c + 1] - e.cl_off[c]) != n) continue;
                int same = 1;
                for (int64_t a = 0; a < n && same; a++) {
                    int f = 0;
                    for (uint64_t k = e.cl_off[c]; k < e.cl_off[c + 1]; k++) f = f || e.cl_lits[k] == proof[st + a];
                    same = f;
                }
                if (same) { e.cl_dead[c] = 1; break; }
            }
            continue;
        }
        /* RUP: assume the negation of every literal, propagate, expect a conflict */
        uint64_t mark = e.trail_n;
        int c2 = 0;
        for (int64_t a = 0; a < n && !c2; a++) c2 = eng_assume(&e, -proof[st + a]);
        if (!c2) c2 = eng_propagate(&e);
        eng_backtrack(&e, mark);
        if (!c2) { ok = 0; break; }
        if (n == 0) { derived_empty = 1; break; }
        uint64_t c = eng_add(&e, proof + st, (uint64_t)n);
        if (eng_examine(&e, c) || eng_propagate(&e)) derived_empty = 1;
    }
    eng_free(&e);
    return ok && derived_empty;
}
