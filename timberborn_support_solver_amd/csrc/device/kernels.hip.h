// kernels.hip.h — hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// Replaces the search loop inside `Solve::solve` of the reference's backend
// (rustsat-glucose -> Glucose `search`/`propagate`/`analyze`/`cancelUntil`/
// `pickBranchLit`/`reduceDB`, [ext]; called from
// crates/repl/src/solver_runner.rs:16 and crates/gui/src/solver_backend.rs:90).
//
// Execution model: ONE 64-lane wavefront = one worker = one independent CDCL
// search over its private slab (layout.h).  A workgroup is a single wave, so
// there are no workgroup barriers; lanes cooperate through ballots, lane
// shuffles and a little LDS:
//   * BCP: the 64 lanes take 64 watchers (or 64 binary implications) of the
//     dequeued literal at a time.  Watch lists are contiguous (cref, blocker)
//     pairs -> one coalesced 512-byte read per chunk.  Kept watchers are
//     compacted in place with ballot + prefix popcount.
//   * conflict detection is a ballot over the lanes' clause states; implied
//     literals are deduplicated through a small LDS claim table (two lanes may
//     imply the same or complementary literals in one chunk) and appended to the
//     trail with a prefix popcount.
//   * the propagation queue (the not-yet-propagated suffix of the trail) is
//     staged in an LDS ring; the trail in HBM is only re-read on ring overflow.
//   * analysis, minimisation, LBD, backjump and the move-to-front decision
//     queue are wave-parallel over clause literals / trail segments.
// Integer / indexing work only: no MFMA.  The bound is memory latency and HBM
// bandwidth on the private slabs; the shared clause literals sit in L2/MALL.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

typedef unsigned long long u64;

#define DEV __device__ __forceinline__

struct Wk {
    // shared
    uint32_t n_vars, n_orig;
    const uint32_t* cl_off;
    const int32_t* cl_lits;
    const uint32_t* bin_off;
    const int32_t* bin_lits;
    // private
    MsState* st;
    uint8_t *val, *phase, *seen;
    int32_t *level, *reason, *trail, *trail_lim, *vm_pos, *vm_order;
    int2* wl;
    uint32_t *w_base, *w_size, *w_cap;
    int2* pool;
    uint32_t *lc_off, *lc_lbd;
    int32_t* lc_lits;
    int32_t *learnt_buf, *toclear;
    uint32_t *lvl_stamp, *remap;
    int32_t *overflow, *assumps, *script;
    uint32_t learnt_cap, learnt_lit_cap, pool_cap, vm_cap;
    // LDS
    volatile int32_t* ring;
    volatile uint32_t* claim;
    volatile uint32_t* ov_cnt;
    volatile uint32_t* hist;   // 64 words, reduce_db
    // hot uniform scalars
    int lane;
    int trail_n, qhead, n_levels, ring_lo;
    int vm_end, vm_search;
    uint32_t n_learnts, lc_lits_n, pool_top;
    int status;
    uint32_t lvl_stamp_ctr;
    // conflict
    int confl_kind, confl_cref, confl_a, confl_b;
    // counters
    u64 c_props, c_watch, c_move, c_enq, c_dec;
    uint32_t c_cl_lit;  // per lane
};

DEV u64 ballot(bool p) { return __ballot(p); }
DEV int popc64(u64 m) { return __popcll(m); }
DEV int first_lane(u64 m) { return __ffsll((long long)m) - 1; }
DEV u64 lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
DEV int bcast(int v, int src) { return __shfl(v, src, 64); }
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Compiler-level ordering between lanes of the same wave (no instruction: memory
// operations of one wave are issued in order; see DESIGN.md "intra-wave ordering").
DEV void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
DEV void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
}
DEV int wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
DEV u64 wave_sum_u32(uint32_t v) {
    u64 s = v;
    for (int o = 32; o > 0; o >>= 1) s += (u64)__shfl_xor((unsigned long long)s, o, 64);
    return s;
}

DEV int lit_value(const Wk& w, int lit) { return (int)w.val[lit >> 1] ^ (lit & 1); }  // 0 T, 1 F, >=2 U

DEV void clause_range(const Wk& w, int c, const int32_t*& lits, int& size) {
    if ((uint32_t)c < w.n_orig) {
        uint32_t o0 = w.cl_off[c], o1 = w.cl_off[c + 1];
        lits = w.cl_lits + o0;
        size = (int)(o1 - o0);
    } else {
        uint32_t k = (uint32_t)c - w.n_orig;
        uint32_t o0 = w.lc_off[k], o1 = w.lc_off[k + 1];
        lits = w.lc_lits + o0;
        size = (int)(o1 - o0);
    }
}

// ---- trail -------------------------------------------------------------
DEV void ring_note_growth(Wk& w) {
    if (w.trail_n - w.ring_lo > MS_LDS_RING) w.ring_lo = w.trail_n - MS_LDS_RING;
}

// all lanes call with identical arguments
DEV void enqueue_uniform(Wk& w, int lit, int reason) {
    wave_fence();  // every lane has finished reading the old assignment
    if (w.lane == 0) {
        int v = lit >> 1;
        w.val[v] = (uint8_t)(lit & 1);
        w.level[v] = w.n_levels;
        w.reason[v] = reason;
        w.trail[w.trail_n] = lit;
        w.ring[w.trail_n & (MS_LDS_RING - 1)] = lit;
    }
    w.trail_n++;
    ring_note_growth(w);
    wave_fence();
}

// Lanes with `want` each imply literal q with reason `reason` (a cref, or a
// binary reason code).  Several lanes may imply the same literal (keep one) or
// complementary literals (conflict).  One LDS claim per variable decides.
DEV void commit_implications(Wk& w, bool want, int q, int reason) {
    u64 m = ballot(want);
    if (m == 0) return;
    if ((m & (m - 1)) == 0) {  // single implication: no arbitration needed
        if (want) {
            int v = q >> 1;
            w.val[v] = (uint8_t)(q & 1);
            w.level[v] = w.n_levels;
            w.reason[v] = reason;
            w.trail[w.trail_n] = q;
            w.ring[w.trail_n & (MS_LDS_RING - 1)] = q;
        }
        w.trail_n++;
        w.c_enq++;
        ring_note_growth(w);
        wave_fence();
        return;
    }
    const uint32_t slot = (((uint32_t)(q >> 1) * 2654435761u) >> 20) & (MS_CLAIM_SLOTS - 1);
    while (m) {
        if (want) w.claim[slot] = ((uint32_t)q << 6) | (uint32_t)w.lane;
        lds_fence();
        bool won = false, cf = false;
        if (want) {
            uint32_t c = w.claim[slot];
            int cq = (int)(c >> 6), cl = (int)(c & 63);
            if (cl == w.lane) { won = true; want = false; }
            else if (cq == q) want = false;                    // same literal implied twice
            else if (cq == (q ^ 1)) { cf = true; want = false; }  // complementary: my clause is now falsified
        }
        u64 wm = ballot(won);
        if (won) {
            int v = q >> 1;
            int t = w.trail_n + popc64(wm & lanemask_lt(w.lane));
            w.val[v] = (uint8_t)(q & 1);
            w.level[v] = w.n_levels;
            w.reason[v] = reason;
            w.trail[t] = q;
            w.ring[t & (MS_LDS_RING - 1)] = q;
        }
        int nw = popc64(wm);
        w.trail_n += nw;
        w.c_enq += (u64)nw;
        ring_note_growth(w);
        u64 cm = ballot(cf);
        if (cm && !w.confl_kind) {
            int f = first_lane(cm);
            int r = bcast(reason, f), qf = bcast(q, f);
            if (MS_IS_BIN_REASON(r)) { w.confl_kind = 2; w.confl_a = MS_BIN_REASON_LIT(r); w.confl_b = qf; }
            else { w.confl_kind = 1; w.confl_cref = r; }
        }
        lds_fence();
        m = ballot(want);
    }
    wave_fence();
}

// ---- watch lists -------------------------------------------------------
// Append (cref, blocker) to the list of literal t (uniform call, rare path:
// learnt clause attach and overflow repair).  Grows the list from the bump pool.
DEV bool list_push_uniform(Wk& w, int t, int cref, int blocker) {
    uint32_t s = w.w_size[t], cap = w.w_cap[t];
    s = (uint32_t)uni((int)s);
    cap = (uint32_t)uni((int)cap);
    if (s > cap) s = cap;  // overshoot left by failed atomic pushes
    uint32_t base = (uint32_t)uni((int)w.w_base[t]);
    if (s == cap) {
        uint32_t ncap = cap < 4 ? 8 : cap * 2;
        if (w.pool_top + ncap > w.pool_cap) { w.status = MS_ST_ERR_POOL; return false; }
        uint32_t nb = w.pool_top;
        w.pool_top += ncap;
        for (uint32_t i = (uint32_t)w.lane; i < s; i += MS_WAVE) w.pool[nb + i] = w.pool[base + i];
        if (w.lane == 0) { w.w_base[t] = nb; w.w_cap[t] = ncap; }
        base = nb;
    }
    if (w.lane == 0) { w.pool[base + s] = make_int2(cref, blocker); w.w_size[t] = s + 1; }
    wave_fence();
    return true;
}

DEV void repair_overflow(Wk& w) {
    uint32_t n = *w.ov_cnt;
    n = (uint32_t)uni((int)n);
    if (n == 0) return;
    wave_fence();
    for (uint32_t e = 0; e < n && w.status == MS_ST_RUNNING; e++) {
        int t = uni(w.overflow[3 * e]), c = uni(w.overflow[3 * e + 1]), b = uni(w.overflow[3 * e + 2]);
        list_push_uniform(w, t, c, b);
    }
    if (w.lane == 0) *w.ov_cnt = 0;
    lds_fence();
}

// Unit propagation to fixpoint.  Returns true on conflict (w.confl_*).
DEV bool propagate(Wk& w) {
    w.confl_kind = 0;
    while (w.qhead < w.trail_n) {
        const int idx = w.qhead++;
        int p = (idx >= w.ring_lo) ? w.ring[idx & (MS_LDS_RING - 1)] : w.trail[idx];
        p = uni(p);
        w.c_props++;
        const int fl = p ^ 1;
        // ---- binary clauses: static CSR shared by all workers -------------
        {
            const uint32_t b0 = w.bin_off[p], b1 = w.bin_off[p + 1];
            for (uint32_t base = b0; base < b1; base += MS_WAVE) {
                uint32_t i = base + (uint32_t)w.lane;
                bool act = i < b1;
                int q = act ? w.bin_lits[i] : 0;
                int vq = act ? lit_value(w, q) : 0;
                w.c_watch += (u64)popc64(ballot(act));
                u64 cm = ballot(act && vq == MS_VAL_FALSE);
                if (cm) {
                    w.confl_kind = 2;
                    w.confl_a = fl;
                    w.confl_b = bcast(q, first_lane(cm));
                    w.qhead = w.trail_n;
                    return true;
                }
                commit_implications(w, act && vq >= MS_VAL_UNDEF, q, MS_REASON_BIN(fl));
                if (w.confl_kind) { w.qhead = w.trail_n; return true; }
            }
        }
        // ---- long clauses: two watched literals -----------------------------
        const uint32_t wb = (uint32_t)uni((int)w.w_base[p]);
        const int n = uni((int)w.w_size[p]);
        int j = 0;
        int i0 = 0;
        for (; i0 < n; i0 += MS_WAVE) {
            const int i = i0 + w.lane;
            const bool act = i < n;
            int2 wt = act ? w.pool[wb + i] : make_int2(0, 0);
            bool keep = act, want = false, cf = false;
            int imp = 0;
            if (act && lit_value(w, wt.y) != MS_VAL_TRUE) {
                const int c = wt.x;
                const int2 ww = w.wl[c];
                const int other = (ww.x == fl) ? ww.y : ww.x;
                uint32_t nl = 2;
                const int bl = wt.y;
                wt.y = other;
                if (other == bl || lit_value(w, other) != MS_VAL_TRUE) {
                    const int32_t* cl;
                    int size;
                    clause_range(w, c, cl, size);
                    int r = -1;
                    for (int k = 0; k < size; k++) {
                        int l = cl[k];
                        if (l == fl || l == other) continue;
                        nl++;
                        if (lit_value(w, l) != MS_VAL_FALSE) { r = l; break; }
                    }
                    if (r >= 0) {
                        w.wl[c] = make_int2(other, r);
                        const int t = r ^ 1;
                        uint32_t pos = atomicAdd(&w.w_size[t], 1u);
                        if (pos < w.w_cap[t]) w.pool[w.w_base[t] + pos] = wt;
                        else {
                            uint32_t o = atomicAdd((uint32_t*)w.ov_cnt, 1u);
                            w.overflow[3 * o] = t;
                            w.overflow[3 * o + 1] = c;
                            w.overflow[3 * o + 2] = other;
                        }
                        keep = false;
                    } else if (lit_value(w, other) == MS_VAL_FALSE) cf = true;
                    else { want = true; imp = other; }
                }
                w.c_cl_lit += nl;
            }
            w.c_watch += (u64)popc64(ballot(act));
            w.c_move += (u64)popc64(ballot(act && !keep));
            // compaction of kept watchers (dest index <= source index)
            u64 km = ballot(keep);
            if (keep) w.pool[wb + j + popc64(km & lanemask_lt(w.lane))] = wt;
            j += popc64(km);
            repair_overflow(w);
            u64 cm = ballot(cf);
            if (cm) {
                w.confl_kind = 1;
                w.confl_cref = bcast(wt.x, first_lane(cm));
            } else {
                commit_implications(w, want, imp, wt.x);
            }
            if (w.confl_kind || w.status != MS_ST_RUNNING) { i0 += MS_WAVE; break; }
        }
        // conflict: copy the unvisited tail down
        for (; i0 < n; i0 += MS_WAVE) {
            const int i = i0 + w.lane;
            if (i < n) {
                int2 wt = w.pool[wb + i];
                w.pool[wb + j + w.lane] = wt;
            }
            j += min(MS_WAVE, n - i0);
        }
        if (w.lane == 0) w.w_size[p] = (uint32_t)j;
        wave_fence();
        if (w.confl_kind || w.status != MS_ST_RUNNING) { w.qhead = w.trail_n; return w.confl_kind != 0; }
    }
    return false;
}

// ---- backtracking --------------------------------------------------------
DEV void cancel_until(Wk& w, int lvl) {
    if (w.n_levels <= lvl) return;
    const int lim = uni(w.trail_lim[lvl]);
    int maxpos = -1;
    for (int i = lim + w.lane; i < w.trail_n; i += MS_WAVE) {
        int l = w.trail[i];
        int v = l >> 1;
        w.val[v] = MS_VAL_UNDEF;
        w.phase[v] = (uint8_t)(l & 1);
        maxpos = max(maxpos, w.vm_pos[v]);
    }
    maxpos = wave_max(maxpos);
    if (maxpos > w.vm_search) w.vm_search = maxpos;
    w.trail_n = lim;
    w.qhead = lim;
    w.n_levels = lvl;
    if (w.ring_lo > lim) w.ring_lo = lim;
    wave_fence();
}

DEV void new_decision_level(Wk& w) {
    if (w.lane == 0) w.trail_lim[w.n_levels] = w.trail_n;
    w.n_levels++;
}

// ---- decision queue (move-to-front as an append-only array) ---------------
// vm_order[0..vm_end) holds variables; the entry of v is live iff vm_pos[v] is its
// index.  Later index = more recently bumped.  vm_search: every live entry above
// it is assigned.
DEV void vm_compact(Wk& w) {
    int j = 0;
    for (int i0 = 0; i0 < w.vm_end; i0 += MS_WAVE) {
        int i = i0 + w.lane;
        int v = -1;
        bool live = false;
        if (i < w.vm_end) { v = w.vm_order[i]; live = w.vm_pos[v] == i; }
        u64 m = ballot(live);
        if (live) {
            int d = j + popc64(m & lanemask_lt(w.lane));
            w.vm_order[d] = v;
            w.vm_pos[v] = d;
        }
        j += popc64(m);
        wave_fence();
    }
    w.vm_end = j;
    w.vm_search = j - 1;
}

DEV int pick_branch_var(Wk& w) {
    for (;;) {
        if (w.vm_search < 0) return -1;
        int idx = w.vm_search - w.lane;
        int v = -1;
        bool ok = false;
        if (idx >= 0) {
            v = w.vm_order[idx];
            ok = w.vm_pos[v] == idx && w.val[v] == MS_VAL_UNDEF;
        }
        u64 m = ballot(ok);
        if (m) {
            int f = first_lane(m);
            w.vm_search -= f;
            return bcast(v, f);
        }
        w.vm_search -= MS_WAVE;
    }
}

// ---- conflict analysis (first UIP) -----------------------------------------
struct Learnt { int n, bt_level; uint32_t lbd; };

DEV void analyze_visit(Wk& w, bool act, int q, int dl, int& path_c, int& n_out, int& n_clear) {
    int v = q >> 1;
    bool fresh = false, cur = false;
    if (act) {
        int lv = w.level[v];
        fresh = !w.seen[v] && lv > 0;
        cur = fresh && lv >= dl;
    }
    u64 fm = ballot(fresh), cm = ballot(cur);
    u64 lm = fm & ~cm;
    if (fresh) {
        w.seen[v] = 1;
        w.toclear[n_clear + popc64(fm & lanemask_lt(w.lane))] = v;
        if (!cur) w.learnt_buf[n_out + popc64(lm & lanemask_lt(w.lane))] = q;
    }
    n_clear += popc64(fm);
    n_out += popc64(lm);
    path_c += popc64(cm);
}

DEV Learnt analyze(Wk& w) {
    int path_c = 0, p = -1, n_out = 1, n_clear = 0;
    int index = w.trail_n - 1;
    const int dl = w.n_levels;
    int kind = w.confl_kind, cref = w.confl_cref, ba = w.confl_a, bb = w.confl_b;
    for (;;) {
        if (kind == 1) {
            const int32_t* cl;
            int size;
            clause_range(w, cref, cl, size);
            if ((uint32_t)cref >= w.n_orig && w.lane == 0) w.lc_lbd[cref - w.n_orig] |= 0x80000000u;  // used
            for (int k0 = 0; k0 < size; k0 += MS_WAVE) {
                int k = k0 + w.lane;
                int q = k < size ? cl[k] : 0;
                analyze_visit(w, k < size && q != p, q, dl, path_c, n_out, n_clear);
            }
        } else {
            int q = w.lane == 0 ? ba : bb;
            analyze_visit(w, w.lane < 2 && q != p, q, dl, path_c, n_out, n_clear);
        }
        wave_fence();
        // walk the trail back to the most recent literal marked seen
        for (;;) {
            int i = index - w.lane;
            int l = i >= 0 ? w.trail[i] : 0;
            bool ok = i >= 0 && w.seen[l >> 1];
            u64 m = ballot(ok);
            if (m) {
                int f = first_lane(m);
                index -= f;
                p = bcast(l, f);
                break;
            }
            index -= MS_WAVE;
            if (index < 0) { w.status = MS_ST_ERR_INTERNAL; return Learnt{0, 0, 0}; }
        }
        index--;
        const int v = p >> 1;
        const int r = uni(w.reason[v]);
        if (w.lane == 0) w.seen[v] = 0;
        wave_fence();
        path_c--;
        if (path_c <= 0) break;
        if (r >= 0) { kind = 1; cref = r; }
        else if (MS_IS_BIN_REASON(r)) { kind = 2; ba = p; bb = MS_BIN_REASON_LIT(r); }
        else { w.status = MS_ST_ERR_INTERNAL; return Learnt{0, 0, 0}; }
    }
    if (w.lane == 0) w.learnt_buf[0] = p ^ 1;
    wave_fence();
    // ---- local minimisation: drop a literal whose reason's other literals are all seen / level 0
    int j = 1;
    for (int i0 = 1; i0 < n_out; i0 += MS_WAVE) {
        int i = i0 + w.lane;
        bool act = i < n_out, keep = act;
        int q = act ? w.learnt_buf[i] : 0;
        if (act) {
            int r = w.reason[q >> 1];
            if (r >= 0) {
                const int32_t* cl;
                int size;
                clause_range(w, r, cl, size);
                bool red = true;
                for (int k = 0; k < size && red; k++) {
                    int l = cl[k];
                    if ((l >> 1) == (q >> 1)) continue;
                    red = w.seen[l >> 1] || w.level[l >> 1] == 0;
                }
                keep = !red;
            } else if (MS_IS_BIN_REASON(r)) {
                int l = MS_BIN_REASON_LIT(r);
                keep = !(w.seen[l >> 1] || w.level[l >> 1] == 0);
            }
        }
        u64 km = ballot(keep);
        if (keep) w.learnt_buf[j + popc64(km & lanemask_lt(w.lane))] = q;
        j += popc64(km);
        wave_fence();
    }
    n_out = j;
    // ---- backjump level = max level among learnt_buf[1..), moved to position 1
    int bt = 0;
    if (n_out > 1) {
        int best = -1, best_i = 0x7fffffff;
        for (int i = 1 + w.lane; i < n_out; i += MS_WAVE) {
            int lv = w.level[w.learnt_buf[i] >> 1];
            if (lv > best) { best = lv; best_i = i; }
        }
        int mx = wave_max(best);
        int cand = (best == mx) ? best_i : 0x7fffffff;
        int mi = -wave_max(-cand);
        bt = mx;
        if (w.lane == 0 && mi != 1) {
            int t = w.learnt_buf[mi];
            w.learnt_buf[mi] = w.learnt_buf[1];
            w.learnt_buf[1] = t;
        }
        wave_fence();
    }
    // ---- LBD: number of distinct decision levels
    uint32_t lbd = 0;
    {
        const uint32_t base = w.lvl_stamp_ctr;
        for (int i0 = 0; i0 < n_out; i0 += MS_WAVE) {
            int i = i0 + w.lane;
            bool act = i < n_out;
            int lv = act ? w.level[w.learnt_buf[i] >> 1] : 0;
            uint32_t id = base + 1 + (uint32_t)i;
            bool cand = act && w.lvl_stamp[lv] <= base;
            if (cand) w.lvl_stamp[lv] = id;
            wave_fence();
            bool won = cand && w.lvl_stamp[lv] == id;
            lbd += (uint32_t)popc64(ballot(won));
            wave_fence();
        }
        uint32_t nb = base + (uint32_t)n_out + 1;
        if (nb > 0xf0000000u) {  // stamp space exhausted: reset
            for (uint32_t i = (uint32_t)w.lane; i < w.n_vars + 2; i += MS_WAVE) w.lvl_stamp[i] = 0;
            nb = 0;
        }
        w.lvl_stamp_ctr = nb;
    }
    // ---- clear marks and bump the analysed variables to the front of the queue
    if (w.vm_end + n_clear > (int)w.vm_cap) vm_compact(w);
    for (int i = w.lane; i < n_clear; i += MS_WAVE) {
        int v = w.toclear[i];
        w.seen[v] = 0;
        w.vm_order[w.vm_end + i] = v;
        w.vm_pos[v] = w.vm_end + i;
    }
    w.vm_end += n_clear;
    wave_fence();
    return Learnt{n_out, bt, lbd};
}

// ---- watch pool garbage collection ---------------------------------------------------
// Lists that outgrow their slot are moved to the top of a bump pool and leave a
// hole behind.  The rebuild lays every list out again, densely, straight from the
// per-clause watched-literal pairs (no read of the old pool): count, exclusive scan
// over the 2*n_vars lists (wave prefix sums), fill.  Runs at a propagation fixpoint.
DEV void rebuild_watches(Wk& w) {
    const uint32_t nlist = 2 * w.n_vars;
    const uint32_t ncl = w.n_orig + w.n_learnts;
    for (uint32_t t = (uint32_t)w.lane; t < nlist; t += MS_WAVE) w.w_size[t] = 0;
    wave_fence();
    for (uint32_t c = (uint32_t)w.lane; c < ncl; c += MS_WAVE) {
        int2 ww = w.wl[c];
        atomicAdd(&w.w_size[ww.x ^ 1], 1u);
        atomicAdd(&w.w_size[ww.y ^ 1], 1u);
    }
    wave_fence();
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < nlist; t0 += MS_WAVE) {
        uint32_t t = t0 + (uint32_t)w.lane;
        uint32_t sz = t < nlist ? w.w_size[t] : 0;
        uint32_t cap = t < nlist ? sz + (sz >> 1) + 4 : 0;
        uint32_t incl = cap;
        for (int o = 1; o < MS_WAVE; o <<= 1) {
            uint32_t x = (uint32_t)__shfl_up((int)incl, o, 64);
            if (w.lane >= o) incl += x;
        }
        if (t < nlist) { w.w_base[t] = run + incl - cap; w.w_cap[t] = cap; w.w_size[t] = 0; }
        run += (uint32_t)bcast((int)incl, 63);
    }
    if (run > w.pool_cap) { w.status = MS_ST_ERR_POOL; return; }
    w.pool_top = run;
    wave_fence();
    for (uint32_t c = (uint32_t)w.lane; c < ncl; c += MS_WAVE) {
        int2 ww = w.wl[c];
        uint32_t pa = atomicAdd(&w.w_size[ww.x ^ 1], 1u);
        w.pool[w.w_base[ww.x ^ 1] + pa] = make_int2((int)c, ww.y);
        uint32_t pb = atomicAdd(&w.w_size[ww.y ^ 1], 1u);
        w.pool[w.w_base[ww.y ^ 1] + pb] = make_int2((int)c, ww.x);
    }
    wave_fence();
}

// ---- learnt clause database reduction ---------------------------------------
// Keep every clause with lbd <= 2, every locked clause and every clause used
// since the last reduction with lbd <= 6; of the rest drop the worse half by an
// LBD cut-off (histogram in LDS, no sort), breaking ties by age.
DEV void reduce_db(Wk& w, volatile uint32_t* hist /* 64 LDS words */) {
    const uint32_t n = w.n_learnts;
    if (w.lane < 64) hist[w.lane] = 0;
    lds_fence();
    for (uint32_t k = (uint32_t)w.lane; k < n; k += MS_WAVE) {
        uint32_t l = w.lc_lbd[k] & 0x7fffffffu;
        atomicAdd((uint32_t*)&hist[l > 63 ? 63 : l], 1u);
    }
    lds_fence();
    // cut: smallest c such that #(lbd > c) <= n/2
    uint32_t cut = 63, above = 0;
    for (int c = 63; c >= 2; c--) {
        uint32_t h = hist[c];
        if (above + h > n / 2) { cut = (uint32_t)c; break; }
        above += h;
        cut = (uint32_t)c - 1;
    }
    if (cut < 2) cut = 2;
    uint32_t quota = n / 2 > above ? n / 2 - above : 0;  // how many of lbd == cut may still go (oldest first)
    // pass 1: decide + build remap, compacting lits/off/lbd/wl in place
    uint32_t nk = 0, nlits = 0, cut_seen = 0;
    for (uint32_t k0 = 0; k0 < n; k0 += MS_WAVE) {
        uint32_t k = k0 + (uint32_t)w.lane;
        bool act = k < n;
        bool del = false;
        uint32_t lb = 0, o0 = 0, o1 = 0;
        int2 ww = make_int2(0, 0);
        bool at_cut = false;
        if (act) {
            uint32_t raw = w.lc_lbd[k];
            lb = raw & 0x7fffffffu;
            bool used = raw >> 31;
            o0 = w.lc_off[k];
            o1 = w.lc_off[k + 1];
            ww = w.wl[w.n_orig + k];
            int cref = (int)(w.n_orig + k);
            bool locked = (lit_value(w, ww.x) == MS_VAL_TRUE && w.reason[ww.x >> 1] == cref) ||
                          (lit_value(w, ww.y) == MS_VAL_TRUE && w.reason[ww.y >> 1] == cref);
            bool protect = locked || lb <= 2 || (used && lb <= 6) || (o1 - o0) <= 2;
            if (!protect) {
                if (lb > cut) del = true;
                else if (lb == cut) at_cut = true;
            }
        }
        u64 cm = ballot(at_cut);
        if (at_cut) {
            uint32_t r = cut_seen + (uint32_t)popc64(cm & lanemask_lt(w.lane));
            if (r < quota) del = true;
        }
        cut_seen += (uint32_t)popc64(cm);
        bool keep = act && !del;
        u64 km = ballot(keep);
        uint32_t nkeep = (uint32_t)popc64(km);
        // exclusive prefix of literal counts among kept clauses
        uint32_t len = keep ? (o1 - o0) : 0, pre = len;
        for (int o = 1; o < MS_WAVE; o <<= 1) {
            uint32_t t = (uint32_t)__shfl_up((int)pre, o, 64);
            if (w.lane >= o) pre += t;
        }
        uint32_t total = (uint32_t)bcast((int)pre, 63);
        pre -= len;
        uint32_t nkk = nk + (uint32_t)popc64(km & lanemask_lt(w.lane));
        if (act) w.remap[k] = keep ? nkk : 0xffffffffu;
        wave_fence();
        // move literals (dest <= source, clause by clause inside the chunk in lane order)
        for (int src = 0; src < MS_WAVE; src++) {
            if (!((km >> src) & 1)) continue;
            uint32_t so = (uint32_t)bcast((int)o0, src), sl = (uint32_t)bcast((int)len, src);
            uint32_t dd = nlits + (uint32_t)bcast((int)pre, src);
            if (dd != so)
                for (uint32_t t = 0; t < sl; t += MS_WAVE) {
                    uint32_t x = t + (uint32_t)w.lane;
                    int lv = x < sl ? w.lc_lits[so + x] : 0;
                    wave_fence();
                    if (x < sl) w.lc_lits[dd + x] = lv;
                    wave_fence();
                }
        }
        if (keep) {
            w.lc_off[nkk] = nlits + pre;
            w.lc_lbd[nkk] = lb;  // clears the used bit
            w.wl[w.n_orig + nkk] = ww;
        }
        nk += nkeep;
        nlits += total;
        wave_fence();
    }
    if (w.lane == 0) w.lc_off[nk] = nlits;
    // pass 2: lay the watch lists out again without the deleted clauses (also collects pool garbage)
    w.n_learnts = nk;
    w.lc_lits_n = nlits;
    wave_fence();
    rebuild_watches(w);
    // pass 3: reasons of assigned variables
    for (int i = w.lane; i < w.trail_n; i += MS_WAVE) {
        int v = w.trail[i] >> 1;
        int r = w.reason[v];
        if (r >= 0 && (uint32_t)r >= w.n_orig) w.reason[v] = (int)(w.n_orig + w.remap[(uint32_t)r - w.n_orig]);
    }
    w.n_learnts = nk;
    w.lc_lits_n = nlits;
    wave_fence();
}

// Store the clause in learnt_buf[0..n) and attach it.  Returns its cref (or -1).
DEV int add_learnt(Wk& w, int n, uint32_t lbd) {
    if (w.n_learnts >= w.learnt_cap || w.lc_lits_n + (uint32_t)n > w.learnt_lit_cap) {
        reduce_db(w, w.hist);  // store full before the scheduled reduction: reduce now (state is consistent here)
        if (w.n_learnts >= w.learnt_cap || w.lc_lits_n + (uint32_t)n > w.learnt_lit_cap) {
            w.status = MS_ST_ERR_LEARNT;
            return -1;
        }
    }
    const uint32_t k = w.n_learnts, o = w.lc_lits_n;
    for (int i = w.lane; i < n; i += MS_WAVE) w.lc_lits[o + i] = w.learnt_buf[i];
    const int l0 = uni(w.learnt_buf[0]), l1 = uni(w.learnt_buf[1]);
    const int cref = (int)(w.n_orig + k);
    if (w.lane == 0) {
        w.lc_off[k] = o;
        w.lc_off[k + 1] = o + (uint32_t)n;
        w.lc_lbd[k] = lbd;
        w.wl[cref] = make_int2(l0, l1);
    }
    w.n_learnts++;
    w.lc_lits_n += (uint32_t)n;
    wave_fence();
    if (!list_push_uniform(w, l0 ^ 1, cref, l1)) return -1;
    if (!list_push_uniform(w, l1 ^ 1, cref, l0)) return -1;
    return cref;
}

// ---- worker load / store -------------------------------------------------------
DEV void wk_bind(Wk& w, const MsShared& sh, const MsLayout& L, char* slab) {
    w.n_vars = sh.n_vars; w.n_orig = sh.n_orig;
    w.cl_off = sh.cl_off; w.cl_lits = sh.cl_lits; w.bin_off = sh.bin_off; w.bin_lits = sh.bin_lits;
    w.st = (MsState*)(slab + L.state);
    w.val = (uint8_t*)(slab + L.val); w.phase = (uint8_t*)(slab + L.phase); w.seen = (uint8_t*)(slab + L.seen);
    w.level = (int32_t*)(slab + L.level); w.reason = (int32_t*)(slab + L.reason);
    w.trail = (int32_t*)(slab + L.trail); w.trail_lim = (int32_t*)(slab + L.trail_lim);
    w.vm_pos = (int32_t*)(slab + L.vm_pos); w.vm_order = (int32_t*)(slab + L.vm_order);
    w.wl = (int2*)(slab + L.wl);
    w.w_base = (uint32_t*)(slab + L.w_base); w.w_size = (uint32_t*)(slab + L.w_size); w.w_cap = (uint32_t*)(slab + L.w_cap);
    w.pool = (int2*)(slab + L.pool);
    w.lc_off = (uint32_t*)(slab + L.lc_off); w.lc_lbd = (uint32_t*)(slab + L.lc_lbd); w.lc_lits = (int32_t*)(slab + L.lc_lits);
    w.learnt_buf = (int32_t*)(slab + L.learnt_buf); w.toclear = (int32_t*)(slab + L.toclear);
    w.lvl_stamp = (uint32_t*)(slab + L.lvl_stamp); w.remap = (uint32_t*)(slab + L.remap);
    w.overflow = (int32_t*)(slab + L.overflow); w.assumps = (int32_t*)(slab + L.assumps); w.script = (int32_t*)(slab + L.script);
    w.learnt_cap = L.learnt_cap; w.learnt_lit_cap = L.learnt_lit_cap; w.pool_cap = L.pool_cap; w.vm_cap = L.vm_cap;
    const MsState* s = w.st;
    w.trail_n = s->trail_n; w.qhead = s->qhead; w.n_levels = s->n_levels;
    w.vm_end = s->vm_end; w.vm_search = s->vm_search;
    w.n_learnts = s->n_learnts; w.lc_lits_n = s->lc_lits_n; w.pool_top = s->pool_top;
    w.status = s->status;
    w.lvl_stamp_ctr = s->lvl_stamp_ctr;
    w.ring_lo = w.trail_n;  // nothing staged yet: the queue suffix is re-read from HBM
    w.confl_kind = 0; w.confl_cref = 0; w.confl_a = 0; w.confl_b = 0;
    w.c_props = w.c_watch = w.c_move = w.c_enq = w.c_dec = 0; w.c_cl_lit = 0;
}

DEV void wk_store(Wk& w, u64 cycles) {
    u64 cl = wave_sum_u32(w.c_cl_lit);
    if (w.lane == 0) {
        MsState* s = w.st;
        s->trail_n = w.trail_n; s->qhead = w.qhead; s->n_levels = w.n_levels;
        s->vm_end = w.vm_end; s->vm_search = w.vm_search;
        s->n_learnts = w.n_learnts; s->lc_lits_n = w.lc_lits_n; s->pool_top = w.pool_top;
        s->status = w.status;
        s->lvl_stamp_ctr = w.lvl_stamp_ctr;
        s->propagations += w.c_props; s->decisions += w.c_dec;
        s->n_watch += w.c_watch; s->n_move += w.c_move; s->n_enq += w.c_enq; s->n_cl_lit += cl;
        s->slice_cycles += cycles;
    }
}

// ---- the search kernel -------------------------------------------------------------
// grid = n_workers blocks of 64 threads.  Runs each worker until it has a verdict,
// or has spent its slice (conflicts / propagations), or the host / another worker
// raised a stop flag.  All state is persisted in the slab, so the host simply
// relaunches the kernel to continue.
__global__ __launch_bounds__(MS_WAVE) void ms_search_kernel(MsShared sh, MsLayout L, char* slabs, MsParams prm) {
    __shared__ int32_t s_ring[MS_LDS_RING];
    __shared__ uint32_t s_claim[MS_CLAIM_SLOTS];
    __shared__ uint32_t s_hist[64];
    __shared__ uint32_t s_lbdq[64];
    __shared__ uint32_t s_ov;
    const uint32_t wid = blockIdx.x;
    if (wid >= prm.n_workers) return;
    Wk w;
    w.lane = (int)threadIdx.x;
    w.ring = s_ring; w.claim = s_claim; w.ov_cnt = &s_ov; w.hist = s_hist;
    if (w.lane == 0) s_ov = 0;
    wk_bind(w, sh, L, slabs + (size_t)wid * L.slab_bytes);
    MsState* st = w.st;
    volatile uint32_t* lbdq = s_lbdq;
    if (w.lane < MS_LBDQ) lbdq[w.lane] = st->lbdq[w.lane];
    lds_fence();
    const u64 t0 = __builtin_readcyclecounter();
    const int n_assumps = st->n_assumps;
    // restart / reduce state: uniform registers for the slice
    u64 conflicts = st->conflicts, restarts = st->restarts, reduce_dbs = st->reduce_dbs;
    u64 lbdq_sum = st->lbdq_sum, lbd_total = st->lbd_total, next_reduce = st->next_reduce;
    uint32_t lbdq_n = st->lbdq_n, lbdq_i = st->lbdq_i;
    double trail_avg = st->trail_avg;
    u64 learnt_total = st->learnt_total, learnt_lits_total = st->learnt_lits_total;
    uint32_t slice_confl = 0;
    const bool entered_running = w.status == MS_ST_RUNNING;
    while (w.status == MS_ST_RUNNING) {
        if (propagate(w)) {
            // ---------------- conflict
            slice_confl++;
            conflicts++;
            if (w.n_levels == 0) { w.status = MS_ST_UNSAT; break; }
            // Glucose restart blocking: a trail much longer than its running average
            // (an exponential average stands in for the 5000-entry queue)
            trail_avg += ((double)w.trail_n - trail_avg) * (1.0 / 5000.0);
            if (conflicts > 10000 && lbdq_n == MS_LBDQ && (double)w.trail_n > 1.4 * trail_avg) {
                lbdq_n = 0; lbdq_i = 0; lbdq_sum = 0;
            }
            Learnt lr = analyze(w);
            if (w.status != MS_ST_RUNNING) break;
            cancel_until(w, lr.bt_level);
            if (lr.n == 1) {
                int l0 = uni(w.learnt_buf[0]);  // unit learnt: bt_level is 0
                if (lit_value(w, l0) == MS_VAL_FALSE) { w.status = MS_ST_UNSAT; break; }
                enqueue_uniform(w, l0, MS_REASON_NONE);
            } else {
                int cref = add_learnt(w, lr.n, lr.lbd);
                if (cref < 0) break;
                enqueue_uniform(w, uni(w.learnt_buf[0]), cref);
            }
            learnt_total++;
            learnt_lits_total += (u64)lr.n;
            lbdq_sum += lr.lbd;
            if (lbdq_n == MS_LBDQ) lbdq_sum -= lbdq[lbdq_i]; else lbdq_n++;
            lds_fence();
            if (w.lane == 0) lbdq[lbdq_i] = lr.lbd;
            lds_fence();
            lbdq_i = (lbdq_i + 1) % MS_LBDQ;
            lbd_total += lr.lbd;
            if (slice_confl >= prm.slice_conflicts) break;
            if ((slice_confl & 63) == 0) {
                if (*prm.stop_flag) break;
                if (prm.stop_on_any && *(volatile int32_t*)prm.any_done) break;
            }
        } else {
            if (w.status != MS_ST_RUNNING) break;
            if (prm.slice_props && w.c_props >= prm.slice_props) break;
            // ---------------- no conflict: restart? reduce? decide
            if (lbdq_n == MS_LBDQ && ((double)lbdq_sum / MS_LBDQ) * 0.8 > (double)lbd_total / (double)conflicts) {
                lbdq_n = 0; lbdq_i = 0; lbdq_sum = 0;
                restarts++;
                cancel_until(w, 0);
            }
            if (conflicts >= next_reduce || w.n_learnts > w.learnt_cap - w.learnt_cap / 8 ||
                w.lc_lits_n > w.learnt_lit_cap - w.learnt_lit_cap / 8) {
                reduce_dbs++;
                next_reduce = conflicts + prm.reduce_first + (u64)prm.reduce_inc * reduce_dbs;
                reduce_db(w, s_hist);
            }
            if (w.pool_top > w.pool_cap - w.pool_cap / 4) rebuild_watches(w);  // pool running low: collect holes
            if (w.status != MS_ST_RUNNING) break;
            int next = -1;
            bool refuted = false;
            while (w.n_levels < n_assumps) {
                int a = uni(w.assumps[w.n_levels]);
                int va = lit_value(w, a);
                if (va == MS_VAL_TRUE) new_decision_level(w);      // dummy level
                else if (va == MS_VAL_FALSE) { refuted = true; break; }
                else { next = a; break; }
            }
            if (refuted) { w.status = MS_ST_UNSAT; break; }
            if (next < 0) {
                int v = pick_branch_var(w);
                if (v < 0) { w.status = MS_ST_SAT; break; }
                w.c_dec++;
                next = uni(2 * v + (int)w.phase[v]);
            }
            new_decision_level(w);
            enqueue_uniform(w, next, MS_REASON_NONE);
        }
    }
    if (entered_running && w.status != MS_ST_RUNNING && w.lane == 0 && prm.any_done) atomicExch(prm.any_done, 1);
    lds_fence();
    if (w.lane < MS_LBDQ) st->lbdq[w.lane] = lbdq[w.lane];
    if (w.lane == 0) {
        st->conflicts = conflicts; st->restarts = restarts; st->reduce_dbs = reduce_dbs;
        st->lbdq_sum = lbdq_sum; st->lbd_total = lbd_total; st->next_reduce = next_reduce;
        st->lbdq_n = lbdq_n; st->lbdq_i = lbdq_i; st->trail_avg = trail_avg;
        st->learnt_total = learnt_total; st->learnt_lits_total = learnt_lits_total;
    }
    wk_store(w, __builtin_readcyclecounter() - t0);
}

// ---- scripted BCP kernel (BASELINE.json configs[1]) -----------------------------------
// Each worker propagates the formula's own units, then takes its scripted
// decisions one decision level at a time.  status: MS_ST_SAT is (ab)used as "fixpoint
// reached without conflict", MS_ST_UNSAT as "conflict".
__global__ __launch_bounds__(MS_WAVE) void ms_bcp_kernel(MsShared sh, MsLayout L, char* slabs, MsParams prm) {
    __shared__ int32_t s_ring[MS_LDS_RING];
    __shared__ uint32_t s_claim[MS_CLAIM_SLOTS];
    __shared__ uint32_t s_ov;
    const uint32_t wid = blockIdx.x;
    if (wid >= prm.n_workers) return;
    Wk w;
    w.lane = (int)threadIdx.x;
    w.ring = s_ring; w.claim = s_claim; w.ov_cnt = &s_ov; w.hist = nullptr;
    if (w.lane == 0) s_ov = 0;
    wk_bind(w, sh, L, slabs + (size_t)wid * L.slab_bytes);
    lds_fence();
    const u64 t0 = __builtin_readcyclecounter();
    const int n_script = w.st->n_script;
    bool confl = propagate(w);
    for (int d = 0; d < n_script && !confl && w.status == MS_ST_RUNNING; d++) {
        int a = uni(w.script[d]);
        int va = lit_value(w, a);
        if (va == MS_VAL_TRUE) continue;
        if (va == MS_VAL_FALSE) { confl = true; break; }
        new_decision_level(w);
        enqueue_uniform(w, a, MS_REASON_NONE);
        confl = propagate(w);
    }
    if (w.status == MS_ST_RUNNING) w.status = confl ? MS_ST_UNSAT : MS_ST_SAT;
    wk_store(w, __builtin_readcyclecounter() - t0);
}
