/*
 * ORACLE (test infrastructure; never shipped, never on the product path).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * cdcl.c — single-thread CPU CDCL solver, "Glucose-class".
 *
 * PARITY UNPINNED against the real reference solver: the reference's solve() is
 * rustsat-glucose 0.7.2 (Cargo.lock:2841-2853), a wrapper around a vendored
 * Glucose 4 C++ tree that is NOT under /root/reference and cannot be built here
 * (no cargo/rustc, no Glucose source).  This file restates the published
 * MiniSat/Glucose algorithm (Een & Sorensson 2003; Audemard & Simon 2009):
 * two-watched-literal propagation with blockers, separate binary watches,
 * first-UIP learning with recursive minimisation, VSIDS with phase saving,
 * LBD-scored learnt-clause reduction and Glucose's dynamic (K/R) restarts.
 * It has no SimpSolver preprocessing (no BVE).  It is anchored on the
 * reference's call sites:
 *   crates/repl/src/solver_runner.rs:12   add_cnf
 *   crates/repl/src/solver_runner.rs:16   solve  -> Sat / Unsat / Interrupted
 *   crates/repl/src/main.rs:329           full_solution
 *   crates/repl/src/main.rs:363           stats
 * What pins it: self-certifying SAT models (every clause satisfied), agreement
 * with PicoSAT verdicts committed in tests/golden/verdicts.json, and the RUP
 * checker below for UNSAT proofs on small instances.
 *
 * Event counters follow SURVEY.md §8(d): one propagation = one trail literal
 * dequeued; n_watch = watchers inspected (binary included), n_cl_lit = clause
 * literals examined after a blocker miss, n_move = watch relocations,
 * n_enq = implied literals.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "oracle.h"

typedef int32_t lit_t; /* 2*var + sign(1 = negative) */
#define LIT_UNDEF (-1)
#define VAR(l) ((l) >> 1)
#define SIGN(l) ((l) & 1)
#define NEG(l) ((l) ^ 1)
#define MKLIT(v, s) (((v) << 1) | (s))
#define L_TRUE 0
#define L_FALSE 1
#define L_UNDEF 2
#define CREF_UNDEF (-1)

typedef struct { int32_t cref; lit_t blocker; } watcher;
typedef struct { watcher* a; int32_t n, cap; } wvec;
typedef struct { int32_t* a; int64_t n, cap; } ivec;

/* clause layout in arena: [0] size | learnt<<30 | deleted<<31, [1] lbd | canbedel<<31, [2] activity(float), [3..] lits */
#define C_HDR 3
#define C_SIZE(c) ((uint32_t)(c)[0] & 0x3fffffffu)
#define C_LEARNT(c) (((uint32_t)(c)[0] >> 30) & 1u)
#define C_DELETED(c) (((uint32_t)(c)[0] >> 31) & 1u)
#define C_LBD(c) ((uint32_t)(c)[1] & 0x7fffffffu)
#define C_LITS(c) ((c) + C_HDR)

struct ora_solver {
    int32_t n_vars;
    ivec arena;
    ivec clauses, learnts; /* crefs */
    wvec* watches;         /* per literal: long clauses watching NEG(lit)... indexed by the literal that became TRUE */
    wvec* watches_bin;
    int8_t* assigns;       /* per var lbool */
    int8_t* polarity;      /* saved phase: 1 = assign false */
    int32_t* level;
    int32_t* reason;
    lit_t* trail;
    int32_t trail_n, qhead;
    ivec trail_lim;
    double* activity;
    double var_inc, var_decay, max_var_decay;
    float cla_inc;
    int32_t* heap;  /* heap of vars */
    int32_t* heap_idx;
    int32_t heap_n;
    int8_t* seen;
    ivec analyze_stack, analyze_toclear, learnt_tmp, last_dl;
    uint32_t* perm_diff;
    uint32_t perm_flag;
    int ok;
    /* restarts */
    uint32_t lbdq[50]; int lbdq_n, lbdq_i; uint64_t lbdq_sum;
    uint32_t* trailq; int trailq_n, trailq_i; uint64_t trailq_sum;
    double sum_lbd;
    /* reduce */
    int64_t next_reduce, cur_restart; int64_t n_reduce_inc;
    ora_stats_t st;
    int8_t* model;
    int32_t model_n;
    volatile int interrupt;
    ivec add_tmp;
    ivec assumptions;
    ivec proof; /* optional DRUP log: clauses as lits (dimacs) 0-terminated */
    int log_proof;
    int32_t max_var_seen;
};

static void iv_push(ivec* v, int32_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 16; v->a = (int32_t*)realloc(v->a, (size_t)v->cap * 4); }
    v->a[v->n++] = x;
}
static void wv_push(wvec* v, watcher w) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 4; v->a = (watcher*)realloc(v->a, (size_t)v->cap * sizeof(watcher)); }
    v->a[v->n++] = w;
}
static inline int value_lit(const ora_solver* s, lit_t l) { return s->assigns[VAR(l)] ^ SIGN(l); } /* 0 T, 1 F, >=2 U */
static inline int32_t* CL(ora_solver* s, int32_t cref) { return s->arena.a + cref; }
static inline int dlevel(const ora_solver* s) { return (int)s->trail_lim.n; }

/* ---- heap (max-activity) ---- */
static inline int heap_lt(ora_solver* s, int32_t a, int32_t b) { return s->activity[a] > s->activity[b]; }
static void heap_up(ora_solver* s, int i) {
    int32_t x = s->heap[i];
    while (i > 0) {
        int p = (i - 1) >> 1;
        if (!heap_lt(s, x, s->heap[p])) break;
        s->heap[i] = s->heap[p]; s->heap_idx[s->heap[i]] = i; i = p;
    }
    s->heap[i] = x; s->heap_idx[x] = i;
}
static void heap_down(ora_solver* s, int i) {
    int32_t x = s->heap[i];
    for (;;) {
        int l = 2 * i + 1, r = l + 1;
        if (l >= s->heap_n) break;
        int c = (r < s->heap_n && heap_lt(s, s->heap[r], s->heap[l])) ? r : l;
        if (!heap_lt(s, s->heap[c], x)) break;
        s->heap[i] = s->heap[c]; s->heap_idx[s->heap[i]] = i; i = c;
    }
    s->heap[i] = x; s->heap_idx[x] = i;
}
static void heap_insert(ora_solver* s, int32_t v) {
    if (s->heap_idx[v] >= 0) return;
    s->heap[s->heap_n] = v; s->heap_idx[v] = s->heap_n; s->heap_n++;
    heap_up(s, s->heap_n - 1);
}
static int32_t heap_pop(ora_solver* s) {
    int32_t x = s->heap[0];
    s->heap_n--;
    s->heap_idx[x] = -1;
    if (s->heap_n > 0) { s->heap[0] = s->heap[s->heap_n]; s->heap_idx[s->heap[0]] = 0; heap_down(s, 0); }
    return x;
}

static void grow_vars(ora_solver* s, int32_t n) {
    if (n <= s->n_vars) return;
    int32_t o = s->n_vars;
    s->watches = (wvec*)realloc(s->watches, sizeof(wvec) * 2 * (size_t)n);
    s->watches_bin = (wvec*)realloc(s->watches_bin, sizeof(wvec) * 2 * (size_t)n);
    memset(s->watches + 2 * (size_t)o, 0, sizeof(wvec) * 2 * (size_t)(n - o));
    memset(s->watches_bin + 2 * (size_t)o, 0, sizeof(wvec) * 2 * (size_t)(n - o));
#define GROW(p, T) s->p = (T*)realloc(s->p, sizeof(T) * (size_t)n)
    GROW(assigns, int8_t); GROW(polarity, int8_t); GROW(level, int32_t); GROW(reason, int32_t);
    GROW(trail, lit_t); GROW(activity, double); GROW(heap, int32_t); GROW(heap_idx, int32_t);
    GROW(seen, int8_t);
    s->perm_diff = (uint32_t*)realloc(s->perm_diff, sizeof(uint32_t) * (size_t)(n + 1));
#undef GROW
    for (int32_t v = o; v < n; v++) {
        s->assigns[v] = L_UNDEF; s->polarity[v] = 1; s->level[v] = 0; s->reason[v] = CREF_UNDEF;
        s->activity[v] = 0; s->heap_idx[v] = -1; s->seen[v] = 0; s->perm_diff[v] = 0;
    }
    s->perm_diff[n] = 0;
    s->n_vars = n;
    for (int32_t v = o; v < n; v++) heap_insert(s, v);
}

ora_solver* ora_new(void) {
    ora_solver* s = (ora_solver*)calloc(1, sizeof(ora_solver));
    s->ok = 1;
    s->var_inc = 1.0; s->var_decay = 0.8; s->max_var_decay = 0.95; s->cla_inc = 1.0f;
    s->trailq = (uint32_t*)calloc(5000, sizeof(uint32_t));
    s->next_reduce = 2000; s->n_reduce_inc = 300;
    return s;
}

void ora_free(ora_solver* s) {
    if (!s) return;
    for (int32_t i = 0; i < 2 * s->n_vars; i++) { free(s->watches[i].a); free(s->watches_bin[i].a); }
    free(s->watches); free(s->watches_bin); free(s->assigns); free(s->polarity); free(s->level); free(s->reason);
    free(s->trail); free(s->activity); free(s->heap); free(s->heap_idx); free(s->seen); free(s->perm_diff);
    free(s->arena.a); free(s->clauses.a); free(s->learnts.a); free(s->trail_lim.a); free(s->analyze_stack.a);
    free(s->analyze_toclear.a); free(s->learnt_tmp.a); free(s->last_dl.a); free(s->trailq); free(s->model);
    free(s->add_tmp.a); free(s->assumptions.a); free(s->proof.a);
    free(s);
}

void ora_reserve(ora_solver* s, uint64_t n_vars) { grow_vars(s, (int32_t)n_vars); if ((int32_t)n_vars > s->max_var_seen) s->max_var_seen = (int32_t)n_vars; }
void ora_interrupt(ora_solver* s) { s->interrupt = 1; }
void ora_enable_proof(ora_solver* s) { s->log_proof = 1; }

static inline void unchecked_enqueue(ora_solver* s, lit_t p, int32_t from) {
    int32_t v = VAR(p);
    s->assigns[v] = (int8_t)SIGN(p);
    s->level[v] = dlevel(s);
    s->reason[v] = from;
    s->trail[s->trail_n++] = p;
}

static int32_t alloc_clause(ora_solver* s, const lit_t* lits, int n, int learnt) {
    int32_t cref = (int32_t)s->arena.n;
    iv_push(&s->arena, (int32_t)((uint32_t)n | ((uint32_t)learnt << 30)));
    iv_push(&s->arena, 0);
    float z = 0; int32_t zi; memcpy(&zi, &z, 4);
    iv_push(&s->arena, zi);
    for (int i = 0; i < n; i++) iv_push(&s->arena, lits[i]);
    return cref;
}

static void attach_clause(ora_solver* s, int32_t cref) {
    int32_t* c = CL(s, cref);
    lit_t* l = C_LITS(c);
    if (C_SIZE(c) == 2) {
        wv_push(&s->watches_bin[NEG(l[0])], (watcher){cref, l[1]});
        wv_push(&s->watches_bin[NEG(l[1])], (watcher){cref, l[0]});
    } else {
        wv_push(&s->watches[NEG(l[0])], (watcher){cref, l[1]});
        wv_push(&s->watches[NEG(l[1])], (watcher){cref, l[0]});
    }
}

static int lit_cmp(const void* a, const void* b) { return (*(const int32_t*)a > *(const int32_t*)b) - (*(const int32_t*)a < *(const int32_t*)b); }

/* add a clause of DIMACS literals at decision level 0 */
static int add_clause_dimacs(ora_solver* s, const int32_t* dl, uint64_t n) {
    if (!s->ok) return 0;
    s->st.n_clauses++;
    s->st.avg_clause_len += (double)n; /* finalised in stats */
    ivec* t = &s->add_tmp;
    t->n = 0;
    for (uint64_t i = 0; i < n; i++) {
        int32_t d = dl[i];
        int32_t v = (d < 0 ? -d : d);
        if (v > s->max_var_seen) s->max_var_seen = v;
        if (v > s->n_vars) grow_vars(s, v);
        iv_push(t, MKLIT(v - 1, d < 0));
    }
    qsort(t->a, (size_t)t->n, 4, lit_cmp);
    int j = 0;
    lit_t prev = LIT_UNDEF;
    for (int i = 0; i < t->n; i++) {
        lit_t l = t->a[i];
        int val = value_lit(s, l);
        if (val == L_TRUE || l == NEG(prev)) return 1; /* satisfied / tautology */
        if (val != L_FALSE && l != prev) { t->a[j++] = l; prev = l; }
    }
    t->n = j;
    if (j == 0) { s->ok = 0; return 0; }
    if (j == 1) {
        unchecked_enqueue(s, t->a[0], CREF_UNDEF);
        return 1; /* propagated lazily at solve start */
    }
    int32_t cref = alloc_clause(s, t->a, j, 0);
    iv_push(&s->clauses, cref);
    attach_clause(s, cref);
    return 1;
}

int ora_add_cnf(ora_solver* s, const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses) {
    for (uint64_t i = 0; i < n_clauses; i++) add_clause_dimacs(s, lits + offsets[i], offsets[i + 1] - offsets[i]);
    return 0;
}

/* ---- propagate ---- */
static int32_t propagate(ora_solver* s) {
    int32_t confl = CREF_UNDEF;
    while (s->qhead < s->trail_n) {
        lit_t p = s->trail[s->qhead++];
        s->st.propagations++;
        s->st.n_deq++;
        /* binary clauses first */
        wvec* wb = &s->watches_bin[p];
        for (int k = 0; k < wb->n; k++) {
            lit_t imp = wb->a[k].blocker;
            s->st.n_watch++;
            int v = value_lit(s, imp);
            if (v == L_FALSE) { s->qhead = s->trail_n; return wb->a[k].cref; }
            if (v >= L_UNDEF) { unchecked_enqueue(s, imp, wb->a[k].cref); s->st.n_enq++; }
        }
        wvec* ws = &s->watches[p];
        watcher *i = ws->a, *j = ws->a, *end = ws->a + ws->n;
        lit_t false_lit = NEG(p);
        while (i != end) {
            s->st.n_watch++;
            lit_t blocker = i->blocker;
            if (value_lit(s, blocker) == L_TRUE) { *j++ = *i++; continue; }
            int32_t cref = i->cref;
            int32_t* c = CL(s, cref);
            lit_t* l = C_LITS(c);
            if (l[0] == false_lit) { l[0] = l[1]; l[1] = false_lit; }
            i++;
            lit_t first = l[0];
            watcher w = {cref, first};
            s->st.n_cl_lit += 2;
            if (first != blocker && value_lit(s, first) == L_TRUE) { *j++ = w; continue; }
            int sz = (int)C_SIZE(c), found = 0;
            for (int k = 2; k < sz; k++) {
                s->st.n_cl_lit++;
                if (value_lit(s, l[k]) != L_FALSE) {
                    l[1] = l[k]; l[k] = false_lit;
                    wv_push(&s->watches[NEG(l[1])], w);
                    s->st.n_move++;
                    found = 1;
                    break;
                }
            }
            if (found) continue;
            *j++ = w;
            if (value_lit(s, first) == L_FALSE) {
                confl = cref;
                s->qhead = s->trail_n;
                while (i < end) *j++ = *i++;
            } else {
                unchecked_enqueue(s, first, cref);
                s->st.n_enq++;
            }
        }
        ws->n = (int32_t)(j - ws->a);
        if (confl != CREF_UNDEF) break;
    }
    return confl;
}

static void cancel_until(ora_solver* s, int lvl) {
    if (dlevel(s) <= lvl) return;
    int32_t lim = s->trail_lim.a[lvl];
    for (int32_t c = s->trail_n - 1; c >= lim; c--) {
        int32_t x = VAR(s->trail[c]);
        s->assigns[x] = L_UNDEF;
        s->polarity[x] = (int8_t)SIGN(s->trail[c]);
        heap_insert(s, x);
    }
    s->qhead = lim;
    s->trail_n = lim;
    s->trail_lim.n = lvl;
}

static void var_bump(ora_solver* s, int32_t v) {
    if ((s->activity[v] += s->var_inc) > 1e100) {
        for (int32_t i = 0; i < s->n_vars; i++) s->activity[i] *= 1e-100;
        s->var_inc *= 1e-100;
    }
    if (s->heap_idx[v] >= 0) heap_up(s, s->heap_idx[v]);
}
static void cla_bump(ora_solver* s, int32_t* c) {
    float a; memcpy(&a, &c[2], 4);
    a += s->cla_inc;
    memcpy(&c[2], &a, 4);
    if (a > 1e20f) {
        for (int64_t i = 0; i < s->learnts.n; i++) {
            int32_t* d = CL(s, s->learnts.a[i]);
            float b; memcpy(&b, &d[2], 4); b *= 1e-20f; memcpy(&d[2], &b, 4);
        }
        s->cla_inc *= 1e-20f;
    }
}

static uint32_t compute_lbd(ora_solver* s, const lit_t* lits, int n) {
    uint32_t nb = 0;
    s->perm_flag++;
    for (int i = 0; i < n; i++) {
        int l = s->level[VAR(lits[i])];
        if (s->perm_diff[l] != s->perm_flag) { s->perm_diff[l] = s->perm_flag; nb++; }
    }
    return nb;
}

static inline uint32_t abstract_level(ora_solver* s, int32_t v) { return 1u << (s->level[v] & 31); }

static int lit_redundant(ora_solver* s, lit_t p, uint32_t abstract_levels) {
    s->analyze_stack.n = 0;
    iv_push(&s->analyze_stack, p);
    int64_t top = s->analyze_toclear.n;
    while (s->analyze_stack.n > 0) {
        lit_t q = s->analyze_stack.a[--s->analyze_stack.n];
        int32_t* c = CL(s, s->reason[VAR(q)]);
        lit_t* l = C_LITS(c);
        int sz = (int)C_SIZE(c);
        if (sz == 2 && value_lit(s, l[0]) == L_FALSE) { lit_t t = l[0]; l[0] = l[1]; l[1] = t; }
        for (int i = 1; i < sz; i++) {
            lit_t r = l[i];
            int32_t v = VAR(r);
            if (!s->seen[v] && s->level[v] > 0) {
                if (s->reason[v] != CREF_UNDEF && (abstract_level(s, v) & abstract_levels) != 0) {
                    s->seen[v] = 1;
                    iv_push(&s->analyze_stack, r);
                    iv_push(&s->analyze_toclear, r);
                } else {
                    for (int64_t j = top; j < s->analyze_toclear.n; j++) s->seen[VAR(s->analyze_toclear.a[j])] = 0;
                    s->analyze_toclear.n = top;
                    return 0;
                }
            }
        }
    }
    return 1;
}

static void analyze(ora_solver* s, int32_t confl, ivec* out, int* out_bt, uint32_t* out_lbd) {
    int path_c = 0;
    lit_t p = LIT_UNDEF;
    out->n = 0;
    iv_push(out, 0);
    int index = s->trail_n - 1;
    s->last_dl.n = 0;
    do {
        int32_t* c = CL(s, confl);
        lit_t* l = C_LITS(c);
        int sz = (int)C_SIZE(c);
        if (p != LIT_UNDEF && sz == 2 && value_lit(s, l[0]) == L_FALSE) { lit_t t = l[0]; l[0] = l[1]; l[1] = t; }
        if (C_LEARNT(c)) {
            cla_bump(s, c);
            if (C_LBD(c) > 2) { /* dynamic LBD update */
                uint32_t nb = compute_lbd(s, l, sz);
                if (nb + 1 < C_LBD(c)) { /* improved: protect once from reduceDB if lbd <= 30 */
                    uint32_t keep = C_LBD(c) <= 30 ? 0x80000000u : ((uint32_t)c[1] & 0x80000000u);
                    c[1] = (int32_t)(keep | nb);
                }
            }
        }
        for (int j = (p == LIT_UNDEF) ? 0 : 1; j < sz; j++) {
            lit_t q = l[j];
            int32_t v = VAR(q);
            if (!s->seen[v] && s->level[v] > 0) {
                var_bump(s, v);
                s->seen[v] = 1;
                if (s->level[v] >= dlevel(s)) {
                    path_c++;
                    if (s->reason[v] != CREF_UNDEF && C_LEARNT(CL(s, s->reason[v]))) iv_push(&s->last_dl, q);
                } else iv_push(out, q);
            }
        }
        while (!s->seen[VAR(s->trail[index--])]) {}
        p = s->trail[index + 1];
        confl = s->reason[VAR(p)];
        s->seen[VAR(p)] = 0;
        path_c--;
    } while (path_c > 0);
    out->a[0] = NEG(p);
    /* minimise (recursive) */
    s->analyze_toclear.n = 0;
    for (int64_t i = 0; i < out->n; i++) iv_push(&s->analyze_toclear, out->a[i]);
    uint32_t abs = 0;
    for (int64_t i = 1; i < out->n; i++) abs |= abstract_level(s, VAR(out->a[i]));
    int64_t j = 1;
    for (int64_t i = 1; i < out->n; i++)
        if (s->reason[VAR(out->a[i])] == CREF_UNDEF || !lit_redundant(s, out->a[i], abs)) out->a[j++] = out->a[i];
    out->n = j;
    /* backtrack level */
    if (out->n == 1) *out_bt = 0;
    else {
        int64_t mx = 1;
        for (int64_t i = 2; i < out->n; i++)
            if (s->level[VAR(out->a[i])] > s->level[VAR(out->a[mx])]) mx = i;
        lit_t t = out->a[mx]; out->a[mx] = out->a[1]; out->a[1] = t;
        *out_bt = s->level[VAR(t)];
    }
    *out_lbd = compute_lbd(s, out->a, (int)out->n);
    /* Glucose: extra bump for last-level vars propagated by good learnt clauses */
    for (int64_t i = 0; i < s->last_dl.n; i++) {
        int32_t v = VAR(s->last_dl.a[i]);
        if (s->reason[v] != CREF_UNDEF && C_LBD(CL(s, s->reason[v])) < *out_lbd) var_bump(s, v);
    }
    for (int64_t i = 0; i < s->analyze_toclear.n; i++) s->seen[VAR(s->analyze_toclear.a[i])] = 0;
}

static int locked(ora_solver* s, int32_t cref) {
    int32_t* c = CL(s, cref);
    lit_t* l = C_LITS(c);
    if (C_SIZE(c) == 2) {
        for (int k = 0; k < 2; k++)
            if (value_lit(s, l[k]) == L_TRUE && s->reason[VAR(l[k])] == cref) return 1;
        return 0;
    }
    return value_lit(s, l[0]) == L_TRUE && s->reason[VAR(l[0])] == cref;
}

static ora_solver* g_sort_s;
static int reduce_cmp(const void* pa, const void* pb) {
    int32_t* x = CL(g_sort_s, *(const int32_t*)pa);
    int32_t* y = CL(g_sort_s, *(const int32_t*)pb);
    /* "x < y" means x is removed before y */
    int xs2 = C_SIZE(x) > 2, ys2 = C_SIZE(y) > 2;
    if (xs2 && !ys2) return -1;
    if (!xs2 && ys2) return 1;
    if (C_LBD(x) > C_LBD(y)) return -1;
    if (C_LBD(x) < C_LBD(y)) return 1;
    float ax, ay; memcpy(&ax, &x[2], 4); memcpy(&ay, &y[2], 4);
    return (ax < ay) ? -1 : (ax > ay);
}

static void detach_all_deleted(ora_solver* s) {
    for (int32_t li = 0; li < 2 * s->n_vars; li++) {
        for (int pass = 0; pass < 2; pass++) {
            wvec* w = pass ? &s->watches_bin[li] : &s->watches[li];
            int j = 0;
            for (int i = 0; i < w->n; i++)
                if (!C_DELETED(CL(s, w->a[i].cref))) w->a[j++] = w->a[i];
            w->n = j;
        }
    }
}

static void reduce_db(ora_solver* s) {
    s->st.reduce_dbs++;
    g_sort_s = s;
    qsort(s->learnts.a, (size_t)s->learnts.n, 4, reduce_cmp);
    int64_t n = s->learnts.n;
    if (n == 0) return;
    if (C_LBD(CL(s, s->learnts.a[n / 2])) <= 3) s->next_reduce += 1000;
    if (C_LBD(CL(s, s->learnts.a[n - 1])) <= 5) s->next_reduce += 1000;
    int64_t limit = n / 2, j = 0;
    for (int64_t i = 0; i < n; i++) {
        int32_t cref = s->learnts.a[i];
        int32_t* c = CL(s, cref);
        int canbedel = !(((uint32_t)c[1] >> 31) & 1u);
        if (C_LBD(c) > 2 && C_SIZE(c) > 2 && canbedel && !locked(s, cref) && i < limit) {
            c[0] = (int32_t)((uint32_t)c[0] | 0x80000000u);
            s->st.learnt_literals -= C_SIZE(c);
            if (s->log_proof) { iv_push(&s->proof, INT32_MIN); for (uint32_t k = 0; k < C_SIZE(c); k++) { lit_t l = C_LITS(c)[k]; iv_push(&s->proof, SIGN(l) ? -(VAR(l) + 1) : VAR(l) + 1); } iv_push(&s->proof, 0); }
        } else {
            if (!canbedel) limit++;
            c[1] = (int32_t)((uint32_t)c[1] & 0x7fffffffu);
            s->learnts.a[j++] = cref;
        }
    }
    s->learnts.n = j;
    detach_all_deleted(s);
    /* arena is not compacted: simplicity over memory in the oracle */
}

static lit_t pick_branch_lit(ora_solver* s) {
    int32_t next = -1;
    while (next == -1 || s->assigns[next] != L_UNDEF) {
        if (s->heap_n == 0) return LIT_UNDEF;
        next = heap_pop(s);
    }
    return MKLIT(next, s->polarity[next]);
}

/* returns 10 / 20 / 0(restart or budget) ; -1 = continue */
static int search(ora_solver* s, int64_t* budget) {
    ivec* learnt = &s->learnt_tmp;
    for (;;) {
        int32_t confl = propagate(s);
        if (confl != CREF_UNDEF) {
            s->st.conflicts++;
            if (s->st.conflicts % 5000 == 0 && s->var_decay < s->max_var_decay) s->var_decay += 0.01;
            if (dlevel(s) == 0) { s->ok = 0; return 20; }   /* the formula itself is refuted, whatever the assumptions: later calls must say so too */
            /* trail queue + restart blocking */
            s->trailq_sum += (uint32_t)s->trail_n;
            if (s->trailq_n == 5000) s->trailq_sum -= s->trailq[s->trailq_i]; else s->trailq_n++;
            s->trailq[s->trailq_i] = (uint32_t)s->trail_n;
            s->trailq_i = (s->trailq_i + 1) % 5000;
            if (s->st.conflicts > 10000 && s->lbdq_n == 50 &&
                (double)s->trail_n > 1.4 * ((double)s->trailq_sum / s->trailq_n)) {
                s->lbdq_n = 0; s->lbdq_i = 0; s->lbdq_sum = 0;
            }
            int bt; uint32_t lbd;
            analyze(s, confl, learnt, &bt, &lbd);
            s->lbdq_sum += lbd;
            if (s->lbdq_n == 50) s->lbdq_sum -= s->lbdq[s->lbdq_i]; else s->lbdq_n++;
            s->lbdq[s->lbdq_i] = lbd;
            s->lbdq_i = (s->lbdq_i + 1) % 50;
            s->sum_lbd += lbd;
            cancel_until(s, bt);
            if (s->log_proof) { for (int64_t k = 0; k < learnt->n; k++) { lit_t l = learnt->a[k]; iv_push(&s->proof, SIGN(l) ? -(VAR(l) + 1) : VAR(l) + 1); } iv_push(&s->proof, 0); }
            if (learnt->n == 1) {
                unchecked_enqueue(s, learnt->a[0], CREF_UNDEF);
            } else {
                int32_t cref = alloc_clause(s, learnt->a, (int)learnt->n, 1);
                int32_t* c = CL(s, cref);
                c[1] = (int32_t)lbd;
                iv_push(&s->learnts, cref);
                attach_clause(s, cref);
                cla_bump(s, c);
                unchecked_enqueue(s, learnt->a[0], cref);
                s->st.learnt_literals += (uint64_t)learnt->n;
            }
            s->var_inc *= 1.0 / s->var_decay;
            s->cla_inc *= 1.0f / 0.999f;
            if (*budget > 0 && --(*budget) == 0) return 0;
        } else {
            if (s->interrupt) return 0;
            /* dynamic restart */
            if (s->lbdq_n == 50 && ((double)s->lbdq_sum / 50) * 0.8 > (s->sum_lbd / (double)s->st.conflicts)) {
                s->lbdq_n = 0; s->lbdq_i = 0; s->lbdq_sum = 0;
                cancel_until(s, 0);
                s->st.restarts++;
                return -1;
            }
            if ((int64_t)s->st.conflicts >= s->next_reduce) {
                s->next_reduce = (int64_t)s->st.conflicts + 2000 + s->n_reduce_inc * (int64_t)(s->st.reduce_dbs + 1);
                reduce_db(s);
            }
            lit_t next = LIT_UNDEF;
            while (dlevel(s) < (int)s->assumptions.n) {
                lit_t p = s->assumptions.a[dlevel(s)];
                int v = value_lit(s, p);
                if (v == L_TRUE) iv_push(&s->trail_lim, s->trail_n);
                else if (v == L_FALSE) return 20; /* UNSAT under assumptions */
                else { next = p; break; }
            }
            if (next == LIT_UNDEF) {
                s->st.decisions++;
                next = pick_branch_lit(s);
                if (next == LIT_UNDEF) return 10;
            }
            iv_push(&s->trail_lim, s->trail_n);
            unchecked_enqueue(s, next, CREF_UNDEF);
        }
    }
}

int ora_solve(ora_solver* s, const int32_t* assumps, int32_t n_assumps, int64_t conflict_budget) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    s->interrupt = 0;
    s->assumptions.n = 0;
    for (int32_t i = 0; i < n_assumps; i++) {
        int32_t d = assumps[i], v = d < 0 ? -d : d;
        if (v > s->n_vars) grow_vars(s, v);
        iv_push(&s->assumptions, MKLIT(v - 1, d < 0));
    }
    int res = -1;
    if (!s->ok) res = 20;
    else if (propagate(s) != CREF_UNDEF) { s->ok = 0; res = 20; }
    int64_t budget = conflict_budget;
    while (res == -1) res = search(s, &budget);
    if (res == 10) {
        s->model = (int8_t*)realloc(s->model, (size_t)s->n_vars + 1);
        s->model_n = s->n_vars;
        for (int32_t v = 0; v < s->n_vars; v++) s->model[v] = s->assigns[v] == L_TRUE ? 1 : -1;
        s->st.n_sat++;
    } else if (res == 20) {
        s->st.n_unsat++;
        if (n_assumps == 0) s->ok = 0;
    } else s->st.n_terminated++;
    cancel_until(s, 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    s->st.solve_seconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return res;
}

int ora_model(ora_solver* s, int8_t* out, uint64_t n_vars) {
    if (!s->model) return -1;
    for (uint64_t v = 0; v < n_vars; v++) out[v] = v < (uint64_t)s->model_n ? s->model[v] : 0;
    return 0;
}

void ora_stats(ora_solver* s, ora_stats_t* out) {
    *out = s->st;
    out->learnts = (uint64_t)s->learnts.n;
    out->max_var = (uint64_t)s->max_var_seen;
    out->avg_clause_len = s->st.n_clauses ? s->st.avg_clause_len / (double)s->st.n_clauses : 0.0;
}

int64_t ora_proof_len(ora_solver* s) { return s->proof.n; }
const int32_t* ora_proof(ora_solver* s) { return s->proof.a; }
