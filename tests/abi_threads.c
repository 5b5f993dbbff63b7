/*
 * TEST INFRASTRUCTURE: replays, against the C ABI of libmi355sat.so, the exact call and threading sequence the
 * reference's Rust side makes through a `Solve` shim (rust/mi355sat/src/lib.rs):
 *
 *   thread A   S::default()                         crates/repl/src/main.rs:295        mi355sat_new
 *   thread A   add_cnf -> add_clause_ref per clause crates/repl/src/solver_runner.rs:12 mi355sat_add per literal
 *   thread A   interrupter()                        solver_runner.rs:13
 *   thread B   solve()   (tokio blocking pool)      solver_runner.rs:15-17             mi355sat_solve
 *   thread C   interrupt() while B solves           main.rs:298-323                    mi355sat_interrupt
 *   thread A   full_solution() = lit_val x V        main.rs:329                        mi355sat_val
 *   thread A   stats(), drop                        main.rs:363                        mi355sat_stats, mi355sat_free
 *
 * usage: abi_threads <cnf.bin> <expect: 10|20|0> <interrupt_after_ms or -1>
 * cnf.bin: int64 n_vars, int64 n_clauses, uint64 offsets[n_clauses+1], int32 lits[]   (DIMACS literals)
 * Exit 0 when the verdict is the expected one and (for SAT) the model read through mi355sat_val satisfies
 * every clause; the model check here is independent of the library (plain loops over the input).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

#include "../include/mi355sat.h"

static mi355sat* g_h;
static int g_result;
static volatile int g_solving;

static void* solve_thread(void* arg) {
    (void)arg;
    g_solving = 1;
    g_result = mi355sat_solve(g_h);
    g_solving = 0;
    return NULL;
}

static void* interrupt_thread(void* arg) {
    long ms = (long)(intptr_t)arg;
    struct timespec ts = {ms / 1000, (ms % 1000) * 1000000L};
    nanosleep(&ts, NULL);
    mi355sat_interrupt(g_h);   /* from a third thread, concurrently with solve() */
    return NULL;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t h[2];
    if (fread(h, 8, 2, f) != 2) return 2;
    const int64_t nv = h[0], nc = h[1];
    uint64_t* offs = malloc(8 * (size_t)(nc + 1));
    if (fread(offs, 8, (size_t)nc + 1, f) != (size_t)nc + 1) return 2;
    int32_t* lits = malloc(4 * (size_t)offs[nc] + 4);
    if (fread(lits, 4, offs[nc], f) != offs[nc]) return 2;
    fclose(f);
    const int expect = atoi(argv[2]);
    const long intr_ms = atol(argv[3]);

    /* thread A */
    g_h = mi355sat_new(NULL);
    if (!g_h) { fprintf(stderr, "mi355sat_new: %s\n", mi355sat_last_error(NULL)); return 3; }
    for (int64_t c = 0; c < nc; c++) {
        for (uint64_t k = offs[c]; k < offs[c + 1]; k++)
            if (mi355sat_add(g_h, lits[k]) < 0) return 4;
        if (mi355sat_add(g_h, 0) < 0) return 4;
    }
    pthread_t tb, tc;
    int have_c = 0;
    if (pthread_create(&tb, NULL, solve_thread, NULL)) return 5;                       /* thread B */
    if (intr_ms >= 0) { have_c = !pthread_create(&tc, NULL, interrupt_thread, (void*)(intptr_t)intr_ms); }   /* thread C */
    pthread_join(tb, NULL);
    if (have_c) pthread_join(tc, NULL);
    /* back on thread A */
    if (g_result < 0) { fprintf(stderr, "solve failed (%d): %s\n", g_result, mi355sat_last_error(g_h)); return 6; }
    printf("result %d\n", g_result);
    int rc = g_result == expect ? 0 : 7;
    if (g_result == MI355SAT_SAT) {
        int8_t* val = malloc((size_t)nv + 1);
        for (int64_t v = 1; v <= nv; v++) {   /* full_solution(): one lit_val per variable */
            int32_t r = mi355sat_val(g_h, (int32_t)v);
            val[v] = r == (int32_t)v ? 1 : (r == -(int32_t)v ? -1 : 0);
        }
        for (int64_t c = 0; c < nc && rc == 0; c++) {
            int sat = 0;
            for (uint64_t k = offs[c]; k < offs[c + 1]; k++) {
                int32_t l = lits[k];
                if ((l > 0 && val[l] > 0) || (l < 0 && val[-l] < 0)) { sat = 1; break; }
            }
            if (!sat) { fprintf(stderr, "clause %lld not satisfied by the model\n", (long long)c); rc = 8; }
        }
        free(val);
    }
    mi355sat_stats_t st;
    if (mi355sat_stats(g_h, &st) != 0) rc = rc ? rc : 9;
    printf("stats n_clauses=%llu max_var=%llu n_sat=%llu n_unsat=%llu n_terminated=%llu conflicts=%llu\n",
           (unsigned long long)st.n_clauses, (unsigned long long)st.max_var, (unsigned long long)st.n_sat,
           (unsigned long long)st.n_unsat, (unsigned long long)st.n_terminated, (unsigned long long)st.conflicts);
    if (st.n_clauses != (uint64_t)nc || st.n_sat + st.n_unsat + st.n_terminated != 1) rc = rc ? rc : 10;
    mi355sat_free(g_h);
    free(offs); free(lits);
    return rc;
}
