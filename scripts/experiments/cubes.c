// EXPERIMENT (not product, not oracle): CPU lookahead cube generator used to judge whether cube-and-conquer
// pays on the support-grid CNFs before the device version is written.
// usage: cubes <cnf.bin> <max_depth> <n_candidates> <min_assigned_frac_x1000> > cubes.txt
// cnf.bin: int64 n_vars, int64 n_clauses, uint64 offsets[n_clauses+1], int32 lits[]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t lit;
static int nv; static int64_t nc;
static uint64_t* offs; static int32_t* cl;   // internal lits
static int8_t* val;        // per var: 0 undef, 1 true, -1 false
static lit* trail; static int trail_n, qhead;
typedef struct { int32_t c; lit blocker; } W;
static W** wl; static int* wn; static int* wcap;
static inline int lv(lit l) { int8_t v = val[l >> 1]; return (l & 1) ? -v : v; }
static void wpush(lit l, W w) { if (wn[l] == wcap[l]) { wcap[l] = wcap[l] ? 2 * wcap[l] : 4; wl[l] = realloc(wl[l], sizeof(W) * wcap[l]); } wl[l][wn[l]++] = w; }
static inline void enq(lit l) { val[l >> 1] = (l & 1) ? -1 : 1; trail[trail_n++] = l; }
static uint64_t n_props;
static int propagate(void) {   // 1 = conflict
    while (qhead < trail_n) {
        lit p = trail[qhead++]; lit f = p ^ 1; n_props++;
        W* ws = wl[f]; int n = wn[f], j = 0, i = 0;
        for (; i < n; i++) {
            W w = ws[i];
            if (lv(w.blocker) > 0) { ws[j++] = w; continue; }
            int32_t* c = cl + offs[w.c]; int sz = (int)(offs[w.c + 1] - offs[w.c]);
            if (c[0] == f) { c[0] = c[1]; c[1] = f; }
            lit first = c[0];
            W nw = {w.c, first};
            if (first != w.blocker && lv(first) > 0) { ws[j++] = nw; continue; }
            int k;
            for (k = 2; k < sz; k++) if (lv(c[k]) >= 0) break;
            if (k < sz) { c[1] = c[k]; c[k] = f; wpush(c[1], nw); continue; }
            ws[j++] = nw;
            if (lv(first) < 0) { for (i++; i < n; i++) ws[j++] = ws[i]; wn[f] = j; qhead = trail_n; return 1; }
            enq(first);
        }
        wn[f] = j;
    }
    return 0;
}
static void backtrack(int to) { while (trail_n > to) { trail_n--; val[trail[trail_n] >> 1] = 0; } qhead = trail_n; }
static int32_t** occ; static int* occ_n; static uint32_t* cstamp; static uint32_t stamp_ctr; static double wlen[64]; static int use_crh;
static int probe_d(lit l, double* score) {
    int m = trail_n; enq(l); int cf = propagate();
    double s = 0;
    if (!cf) {
        if (!use_crh) s = trail_n - m;
        else {
            stamp_ctr++;
            for (int t = m; t < trail_n; t++) {
                lit f = trail[t] ^ 1;
                for (int e = 0; e < occ_n[f]; e++) {
                    int32_t c = occ[f][e]; if (cstamp[c] == stamp_ctr) continue; cstamp[c] = stamp_ctr;
                    int nfree = 0, sat = 0;
                    for (uint64_t k = offs[c]; k < offs[c + 1]; k++) { int v = lv(cl[k]); if (v > 0) { sat = 1; break; } if (v == 0) nfree++; }
                    if (!sat) s += wlen[nfree < 63 ? nfree : 63];
                }
            }
            s += 0.01 * (trail_n - m);
        }
    }
    backtrack(m); *score = s; return cf;
}

static int* cand; static int ncand;
static int max_depth, min_assigned;
static lit cube[4096]; static int cube_n;
static uint64_t n_cubes, n_refuted, n_probes;
static int root_trail;

static void emit(void) { n_cubes++; for (int i = 0; i < cube_n; i++) { lit l = cube[i]; printf("%d ", (l & 1) ? -((l >> 1) + 1) : (l >> 1) + 1); } printf("0\n"); }

static void node(int depth) {
    int mark = trail_n, cmark = cube_n;
    for (;;) {   // failed-literal closure over the candidates
        int best = -1; double best_s = -1; double bp = 0, bn = 0; int forced = 0;
        for (int i = 0; i < ncand; i++) {
            int v = cand[i]; if (val[v]) continue;
            double cp, cn; n_probes += 2;
            int fp = probe_d(2 * v, &cp), fn = probe_d(2 * v + 1, &cn);
            if (fp && fn) { n_refuted++; backtrack(mark); cube_n = cmark; return; }
            if (fp || fn) { lit l = fp ? 2 * v + 1 : 2 * v; enq(l); if (propagate()) { n_refuted++; backtrack(mark); cube_n = cmark; return; } cube[cube_n++] = l; forced++; continue; }
            double s = (double)cp * cn * 1024.0 + cp + cn;
            if (s > best_s) { best_s = s; best = v; bp = cp; bn = cn; }
        }
        if (forced) continue;
        if (best < 0) { emit(); break; }
        if (depth >= max_depth || trail_n - root_trail >= min_assigned) { emit(); break; }
        lit first = bp >= bn ? 2 * best : 2 * best + 1;
        for (int s = 0; s < 2; s++) {
            lit l = s ? (first ^ 1) : first;
            int m2 = trail_n;
            enq(l); cube[cube_n++] = l;
            if (!propagate()) node(depth + 1); else n_refuted++;
            cube_n--; backtrack(m2);
        }
        break;
    }
    backtrack(mark); cube_n = cmark;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); int64_t h[2]; if (fread(h, 8, 2, f) != 2) return 1; nv = (int)h[0]; nc = h[1];
    max_depth = atoi(argv[2]); int M = atoi(argv[3]); int frac = atoi(argv[4]);
    offs = malloc(8 * (nc + 1)); if (fread(offs, 8, nc + 1, f) != (size_t)(nc + 1)) return 1;
    cl = malloc(4 * offs[nc]); if (fread(cl, 4, offs[nc], f) != offs[nc]) return 1;
    for (uint64_t i = 0; i < offs[nc]; i++) { int d = cl[i]; cl[i] = d > 0 ? 2 * (d - 1) : 2 * (-d - 1) + 1; }
    val = calloc(nv, 1); trail = malloc(4 * nv); wl = calloc(2 * nv, sizeof(W*)); wn = calloc(2 * nv, 4); wcap = calloc(2 * nv, 4);
    for (int64_t c = 0; c < nc; c++) {
        int sz = (int)(offs[c + 1] - offs[c]); int32_t* p = cl + offs[c];
        if (sz == 1) { if (lv(p[0]) < 0) { fprintf(stderr, "unsat units\n"); return 0; } if (!lv(p[0])) enq(p[0]); continue; }
        wpush(p[0], (W){(int32_t)c, p[1]}); wpush(p[1], (W){(int32_t)c, p[0]});
    }
    if (propagate()) { fprintf(stderr, "root conflict\n"); return 0; }
    use_crh = argc > 5 ? atoi(argv[5]) : 0;
    occ = calloc(2 * nv, sizeof(int32_t*)); occ_n = calloc(2 * nv, 4); cstamp = calloc(nc, 4);
    for (uint64_t i = 0; i < offs[nc]; i++) occ_n[cl[i]]++;
    for (int l = 0; l < 2 * nv; l++) { occ[l] = malloc(4 * (occ_n[l] + 1)); occ_n[l] = 0; }
    for (int64_t c = 0; c < nc; c++) for (uint64_t k = offs[c]; k < offs[c + 1]; k++) occ[cl[k]][occ_n[cl[k]]++] = (int32_t)c;
    wlen[0] = wlen[1] = 0; wlen[2] = 1; for (int i = 3; i < 64; i++) wlen[i] = wlen[i - 1] / (use_crh > 1 ? use_crh : 5);
    root_trail = trail_n;
    min_assigned = (int)((int64_t)(nv - root_trail) * frac / 1000);
    // root ranking of all free variables
    double* sc = malloc(8 * nv); int* idx = malloc(4 * nv); int nf = 0;
    for (int v = 0; v < nv; v++) {
        if (val[v]) continue;
        double cp, cn; int fp = probe_d(2 * v, &cp), fn = probe_d(2 * v + 1, &cn);
        if (fp && fn) { fprintf(stderr, "root refuted\n"); return 0; }
        if (fp || fn) { enq(fp ? 2 * v + 1 : 2 * v); if (propagate()) { fprintf(stderr, "root refuted\n"); return 0; } continue; }
        sc[v] = (double)cp * cn * 1024.0 + cp + cn; idx[nf++] = v;
    }
    root_trail = trail_n;
    for (int i = 0; i < nf; i++) for (int j = i + 1; j < nf && i < M; j++) if (sc[idx[j]] > sc[idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    ncand = nf < M ? nf : M; cand = idx;
    fprintf(stderr, "vars %d free %d root units %d candidates %d (best score %.0f, worst %.0f)\n", nv, nf, root_trail, ncand, sc[idx[0]], sc[idx[ncand - 1]]);
    node(0);
    fprintf(stderr, "cubes %llu refuted-by-lookahead %llu probes %llu props %llu\n", (unsigned long long)n_cubes, (unsigned long long)n_refuted, (unsigned long long)n_probes, (unsigned long long)n_props);
    return 0;
}
