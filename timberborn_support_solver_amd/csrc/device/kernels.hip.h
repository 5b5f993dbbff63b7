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
// shuffles and LDS:
//   * BCP propagates up to 8 queue literals per step: the wave splits into G lane
//     groups (G = 1/2/4/8 by queue length), one dequeued literal per group.  Each
//     group streams that literal's three lists with coalesced reads: binary
//     implications and ternary literal pairs from the SHARED read-only CSRs
//     (no watches, no writes), and the private two-watched-literal list of the
//     long and learnt clauses, compacting kept watchers in place with a
//     group-masked ballot + prefix popcount.
//   * the assignment lives in LDS, 2 bits per variable, when it fits (template
//     parameter LV); every value lookup of BCP is then an LDS gather.
//   * conflict detection is a ballot over the lanes' clause states; implied
//     literals are deduplicated through an LDS claim table (lanes may imply the
//     same or complementary literals in one step) and appended to the trail with a
//     prefix popcount.
//   * the propagation queue (the not-yet-propagated suffix of the trail) is staged
//     in an LDS ring; the trail in HBM is only re-read on ring overflow.
//   * analysis, minimisation, LBD, backjump, the move-to-front decision queue and
//     the learnt-clause reduction are wave-parallel over literals / trail segments.
// Integer / indexing work only: no MFMA.  The bound is memory latency and HBM
// bandwidth on the private slabs; the shared CSRs sit in L2 / Infinity Cache.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

typedef unsigned long long u64;

// Register budget of the search kernel: 4 waves per SIMD = 16 workers per CU = 4096 per GPU.
// (Measured on rect 64x64: 3.9e9 prop/s at 4/SIMD with ~50 spilled VGPRs vs 3.7e9 at 3/SIMD without spills.)
#ifndef MS_SEARCH_WAVES_PER_SIMD
#define MS_SEARCH_WAVES_PER_SIMD 4
#endif

// Optional per-phase cycle stamps (diagnostic build only: make prof -> libmi355sat_prof.so).
// Phase totals land in MsState::reserved[0..5] + n_prof[...] and are never read by the solver.
#ifdef MS_PROFILE
#define PROF_DECL u64 prof_t_ = __builtin_readcyclecounter();
#define PROF_MARK(slot)                                              \
    do {                                                             \
        u64 n_ = __builtin_readcyclecounter();                       \
        w.prof[slot] += (n_ - prof_t_);                    \
        prof_t_ = n_;                                                \
    } while (0)
#define PROF_RESET prof_t_ = __builtin_readcyclecounter();
#else
#define PROF_DECL
#define PROF_MARK(slot)
#define PROF_RESET
#endif
enum { PF_OFF = 0, PF_BIN, PF_TERN, PF_LONG, PF_CLOSE, PF_ANALYZE, PF_BACKJUMP, PF_DECIDE, PF_REDUCE, PF_N,
       // counts and sub-phases of conflict analysis (slots behind the phase shares)
       PF_RES_STEPS = PF_N, PF_MIN_DEEP, PF_MIN_LOCAL, PF_MIN_NODES, PF_MIN_CALLS, PF_ALL };

#define DEV __device__ __forceinline__
#ifndef MS_BATCH_WALK
#define MS_BATCH_WALK 1     // batched resolution also where the analysis marks are bytes in the variable records (0: one literal per round)
#endif
#ifndef MS_TAIL_UNROLL
#define MS_TAIL_UNROLL 2      // chunks of a clause tail examined per round trip (long_eval, phase B)
#endif
// cold paths are real calls: keeps them out of the hot loop's register allocation
#define DEV_COLD __device__ __noinline__

// ---- address spaces ----------------------------------------------------------------
// A HIP pointer is generic ("flat") unless its type says otherwise, and the compiler's address-space inference never
// rewrites a VOLATILE access: round 2's kernels made every LDS access (ring, claim set, assignment, marks: all through
// `volatile T*` fields of Wk) and most slab accesses (pointers rebuilt from integers, or handed through scratch copies
// of Wk) FLAT instructions - 472 flat_load / 377 flat_store and ONE ds_read in the one-worker-per-SIMD build.  A flat
// access to LDS goes the vector-memory way round, counts in vmcnt AND lgkmcnt and returns out of order, so every use
// waits for `vmcnt(0) lgkmcnt(0)`: each LDS look-up drained every global load in flight.  Here every pointer carries its
// address space in its type: Gp<T> = global (the worker slabs, the shared clause database, the exchange ring),
// LdsI32 / LdsU32 = LDS.  The host-side emulator build (tests/emu) has one address space: the qualifiers vanish.
#if defined(__HIP_DEVICE_COMPILE__)
#define MS_LDS __attribute__((address_space(3)))
#define MS_GLB __attribute__((address_space(1)))
#else
#define MS_LDS
#define MS_GLB
#endif
template <class T> using Gp = T MS_GLB*;
template <class T> DEV T gld(Gp<const T> p) { return *p; }   // (a load as a value: `c ? *p : x` would mix address spaces)
typedef volatile int32_t MS_LDS* LdsI32;
typedef volatile uint32_t MS_LDS* LdsU32;
// LDS read-modify-write (ds_or / ds_and / ds_add_rtn / ds_cmpst_rtn); the wave is the only agent that sees this memory
DEV void lds_or(LdsU32 p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
DEV void lds_and(LdsU32 p, uint32_t v) { (void)__hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
DEV uint32_t lds_or_rtn(LdsU32 p, uint32_t v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
DEV uint32_t lds_add(LdsU32 p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
DEV uint32_t lds_cas(LdsU32 p, uint32_t expect, uint32_t v) {   // returns what was there
    (void)__hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return expect;
}

struct Wk {
    Gp<char> slab;              // this worker's private slab; arrays are at slab + L.<field>
    // LDS
    LdsI32 ring;
    LdsU32 claim;
    LdsU32 ov_cnt;
    LdsU32 hist;   // 64 words, reduce_db; also the scratch of flat_setup (BCP)
    LdsU32 lval;   // packed assignment, 2 bits per variable (LV variants)
    LdsU32 lseen;  // conflict analysis' "seen" marks, 1 bit per variable (LV variants; all zero between analyses)
    LdsU32 lcur;   // (LV) variables assigned at the CURRENT decision level (level > 0)
    LdsU32 lzero;  // (LV) variables assigned at level 0
    LdsU32 lfail;  // (LV) recursive minimisation: variables the learnt clause does not imply (all zero between analyses)
    LdsU32 lq;     // (LV) ... variables already in the node list
    LdsU32 mcnt;   // (LV) ... its length
    LdsU32 sortbuf = nullptr; int sort_n = 0;   // queue positions of the analysed variables (bump order); 0 entries in the builds without it
    LdsI32 bfl;     // the false literal of each lane group of the current BCP step
    // hot uniform scalars
    int lane;
    int trail_n, qhead, n_levels, ring_lo;
    int vm_end, vm_search;
    uint32_t n_learnts, lc_lits_n, pool_top;
    int status;
    uint32_t lvl_stamp_ctr;
    int max_groups;
    // conflict: kind 1 long (cref), 2 binary (a,b), 3 ternary (a,b,c)
    int confl_kind, confl_cref, confl_a, confl_b, confl_c;
    // per-slice counters (flushed into the 64-bit totals of MsState at slice end)
    uint32_t c_props, c_watch, c_move, c_enq, c_dec, c_steps, c_redo;
    uint32_t c_cl_lit;  // per lane
#ifdef MS_PROFILE
    u64 prof[PF_ALL];          // phase cycles, then counts / sub-phases of conflict analysis
#endif
    LdsU32 tl;                 // owner lanes of the clause tails scanned together (64 words)
    LdsI32 jd;                 // j / done of every group while the remainder of the watch lists is visited (2 x MS_MAX_GROUPS)
};

// Arrays are addressed through the kernel arguments (scalar registers / scalar loads), not through
// pointers held per worker: `sh` = shared immutable CSRs, `L` = offsets inside the private slab.
#define WK_PTR(T, w, L, field) ((Gp<T>)((w).slab + (L).field))
#define WKA(T, field) ((Gp<T>)(w.slab + L.field))
#define VREC WKA(MsVarRec, vrec)
#define VMPOS WKA(int32_t, vm_pos)
#define HX(t) MS_HIDX(t, L.n_vars)      // slot of literal t's header in whdr

DEV u64 ballot(bool p) { return __ballot(p); }
DEV int popc64(u64 m) { return __popcll(m); }
DEV int first_lane(u64 m) { return __ffsll((long long)m) - 1; }
DEV u64 lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
DEV int bcast(int v, int src) { return __shfl(v, src, 64); }
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Compiler-level ordering between lanes of the same wave (no instruction: memory
// operations of one wave are issued and performed in order; DESIGN.md "intra-wave ordering").
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

// ---- assignment ------------------------------------------------------------------
// In the slab: one BYTE per variable (MS_ASG_*: bit1 assigned, bit0 sign), read and written with plain loads and stores.
// The wave is the only agent that ever touches its slab and a wave's memory operations are performed in order, so
// lanes see each other's assignments without atomics.  (Rounds 1-2 packed 2 bits per variable and updated the words
// with agent-scope fetch_or / fetch_and and read them with agent-scope loads: on this multi-XCD part a device-scope
// read-modify-write is not done in the XCD's L2 but leaves it as an uncached 64-byte request to the memory side, and
// there were two of them per propagated literal - set and clear - plus an L2-bypassing load per look-up.)
// LV: the assignment is staged in LDS for the slice, 2 bits per variable, 16 variables per word.
template <bool LV>
DEV int lit_value(const Wk& w, const MsShared& sh, const MsLayout& L, int lit) {  // MS_VAL_TRUE / FALSE / UNDEF
    const int v = lit >> 1;
    uint32_t x;
    if (LV) x = (w.lval[v >> 4] >> ((v & 15) * 2)) & 3u;
    else x = WKA(uint8_t, val)[v];
    return (x & 2u) ? (int)((x ^ (uint32_t)lit) & 1u) : MS_VAL_UNDEF;
}
// A look-up in two halves: the fetch (the LDS word holding the variable's two bits, or its byte in the slab) and the
// decoding of the literal's value.  Callers with several look-ups to make issue all the fetches UNCONDITIONALLY and
// back to back (lanes with nothing to look up fetch a harmless literal), so that a group of look-ups costs ONE wait:
// a look-up under a lane condition compiles to a branch with a load and a wait of its own - eight clause literals were
// eight dependent round trips (to LDS, or, with the assignment in the slab, to the L2 / HBM).
template <bool LV>
DEV uint32_t val_fetch(const Wk& w, const MsShared& sh, const MsLayout& L, int lit) {
    if (LV) return w.lval[lit >> 5];
    return (uint32_t)WKA(uint8_t, val)[lit >> 1];
}
template <bool LV>
DEV int val_decode(uint32_t raw, int lit) {
    const uint32_t x = LV ? raw >> (uint32_t)(lit & 30) : raw;
    return (x & 2u) ? (int)((x ^ (uint32_t)lit) & 1u) : MS_VAL_UNDEF;
}
template <bool LV>
DEV void asg_set(Wk& w, const MsShared& sh, const MsLayout& L, int lit) {  // variable currently unassigned
    const int v = lit >> 1;
    if (LV) lds_or(&w.lval[v >> 4], (2u | (uint32_t)(lit & 1)) << ((v & 15) * 2));
    else WKA(uint8_t, val)[v] = (uint8_t)(2u | (uint32_t)(lit & 1));
}
template <bool LV>
DEV void asg_clear(Wk& w, const MsShared& sh, const MsLayout& L, int v) {
    if (LV) lds_and(&w.lval[v >> 4], ~(3u << ((v & 15) * 2)));
    else WKA(uint8_t, val)[v] = MS_ASG_UNDEF;
}

// ---- "seen" marks of conflict analysis ----------------------------------------------------
// LV: one bit per variable in LDS (a trail walk tests 64 entries at once without touching HBM); otherwise the
// byte in the variable's record.
template <bool LV>
DEV bool seen_get(const Wk& w, const MsShared& sh, const MsLayout& L, int v) {
    if (LV) return (w.lseen[v >> 5] >> (v & 31)) & 1u;
    return VREC[v].seen != 0;
}
template <bool LV>
DEV void seen_set(Wk& w, const MsShared& sh, const MsLayout& L, int v) {
    if (LV) lds_or(&w.lseen[v >> 5], 1u << (v & 31));
    else VREC[v].seen = 1;
}
template <bool LV>
DEV void seen_clr(Wk& w, const MsShared& sh, const MsLayout& L, int v) {
    if (LV) lds_and(&w.lseen[v >> 5], ~(1u << (v & 31)));
    else VREC[v].seen = 0;
}

// where a clause's literals are: the shared store of the original long clauses, or the worker's learnt store
DEV Gp<const int32_t> lits_base(const Wk& w, const MsShared& sh, const MsLayout& L, int c) {
    return (uint32_t)c < sh.n_orig ? (Gp<const int32_t>)sh.cl_lits : (Gp<const int32_t>)WKA(int32_t, lc_lits);
}
DEV void clause_range(const Wk& w, const MsShared& sh, const MsLayout& L, int c, Gp<const int32_t>& lits, int& size) {
    const MsClauseRec h = WKA(MsClauseRec, wl)[c];
    lits = lits_base(w, sh, L, c) + h.start;
    size = (int)h.size;
}

// ---- trail -------------------------------------------------------------
DEV void ring_note_growth(Wk& w, const MsShared& sh, const MsLayout& L) {
    if (w.trail_n - w.ring_lo > MS_LDS_RING) w.ring_lo = w.trail_n - MS_LDS_RING;
}

// all lanes call with identical arguments
// (LV) which level a variable was assigned at, as far as conflict analysis cares: level 0, the current one, or lower
template <bool LV>
DEV void level_mark(Wk& w, int v) {
    if (!LV) return;
    if (w.n_levels == 0) lds_or(&w.lzero[v >> 5], 1u << (v & 31));
    else lds_or(&w.lcur[v >> 5], 1u << (v & 31));
}
DEV MsVarRec var_rec(int level, int reason, uint32_t start, uint32_t size, int lit) {
    return MsVarRec{level, reason, start, (uint16_t)(size > 0xffffu ? 0u : size), (uint8_t)(lit & 1), 0};
}

template <bool LV>
DEV void enqueue_uniform(Wk& w, const MsShared& sh, const MsLayout& L, int lit, int reason, uint32_t start = 0, uint32_t size = 0) {
    wave_fence();  // every lane has finished reading the old assignment
    if (w.lane == 0) {
        int v = lit >> 1;
        asg_set<LV>(w, sh, L, lit);
        level_mark<LV>(w, v);
        VREC[v] = var_rec(w.n_levels, reason, start, size, lit);
        WKA(int32_t, trail)[w.trail_n] = lit;
        w.ring[w.trail_n & (MS_LDS_RING - 1)] = lit;
    }
    w.trail_n++;
    ring_note_growth(w, sh, L);
    lds_fence();
}

// ---- implication arbitration ---------------------------------------------------------
// Lanes of one step may imply the same literal (keep one) or complementary literals
// (conflict).  An exact claim set in LDS (open addressing on the variable, one CAS per probe)
// decides; it is cleared at the start of every commit round.
enum { CLAIM_NONE = 0, CLAIM_WON = 1, CLAIM_DUP = 2, CLAIM_LOST = 3 };

DEV void claims_clear(Wk& w) {
#pragma unroll
    for (int k = 0; k < MS_CLAIM_SLOTS / MS_WAVE; k++) w.claim[w.lane + MS_WAVE * k] = 0;
    lds_fence();
}

DEV int claim_insert(Wk& w, bool want, int q) {
    if (!want) return CLAIM_NONE;
    const uint32_t key = ((uint32_t)(q + 1) << 6) | (uint32_t)w.lane;
    uint32_t slot = (((uint32_t)(q >> 1) * 2654435761u) >> 20) & (MS_CLAIM_SLOTS - 1);
    for (;;) {
        const uint32_t old = lds_cas(&w.claim[slot], 0u, key);
        if (old == 0) return CLAIM_WON;
        const uint32_t oq = (old >> 6) - 1;
        if ((oq >> 1) == (uint32_t)(q >> 1)) return oq == (uint32_t)q ? CLAIM_DUP : CLAIM_LOST;
        slot = (slot + 1) & (MS_CLAIM_SLOTS - 1);
    }
}

// Winners append their literal to the trail (prefix popcount) and record level / reason.
template <bool LV>
DEV void assign_winners(Wk& w, const MsShared& sh, const MsLayout& L, bool won, int q, int reason, uint32_t start = 0, uint32_t size = 0) {
    const u64 wm = ballot(won);
    if (wm == 0) return;
    if (won) {
        const int v = q >> 1;
        const int t = w.trail_n + popc64(wm & lanemask_lt(w.lane));
        asg_set<LV>(w, sh, L, q);
        level_mark<LV>(w, v);
        VREC[v] = var_rec(w.n_levels, reason, start, size, q);
        WKA(int32_t, trail)[t] = q;
        w.ring[t & (MS_LDS_RING - 1)] = q;
    }
    const int nw = popc64(wm);
    w.trail_n += nw;
    w.c_enq += (uint32_t)nw;
    ring_note_growth(w, sh, L);
}

// One candidate per lane (later chunks of a list).  `lost`: my implication is already falsified.
template <bool LV>
DEV void commit_implications(Wk& w, const MsShared& sh, const MsLayout& L, bool want, int q, int reason, bool& lost, uint32_t start = 0,
                             uint32_t size = 0) {
    lost = false;
    const u64 m = ballot(want);
    if (m == 0) return;
    if ((m & (m - 1)) == 0) {  // single implication: no arbitration needed
        assign_winners<LV>(w, sh, L, want, q, reason, start, size);
        lds_fence();
        return;
    }
    claims_clear(w);
    const int c = claim_insert(w, want, q);
    lds_fence();
    lost = c == CLAIM_LOST;
    assign_winners<LV>(w, sh, L, c == CLAIM_WON, q, reason, start, size);
    lds_fence();
}

DEV MsClauseHdr clause_hdr_of(const Wk& w, const MsShared& sh, const MsLayout& L, int c) {
    const MsClauseRec h = WKA(MsClauseRec, wl)[c];
    return MsClauseHdr{h.start, h.size};
}

// ---- watch lists -------------------------------------------------------
// A watcher is 16 bytes: {cref, blocker, start, size} - where the clause's literals are travels with it, so a
// visit whose blocker is not true fetches the clause's watched pair (wl[cref]) and its first literals in ONE round
// trip instead of two dependent ones.
// Append (cref, blocker) to the list of literal t (uniform call, rare path:
// learnt clause attach and overflow repair).  Grows the list from the bump pool.
DEV bool list_push_uniform(Wk& w, const MsShared& sh, const MsLayout& L, int t, int cref, int blocker, uint32_t start, uint32_t size) {
    Gp<MsWatchHdr> whdr = WKA(MsWatchHdr, whdr);
    Gp<int4> pool = WKA(int4, pool);
    uint32_t s = (uint32_t)uni((int)whdr[HX(t)].size);
    uint32_t cap = (uint32_t)uni((int)whdr[HX(t)].cap);
    if (s > cap) s = cap;  // overshoot left by failed atomic pushes
    uint32_t base = (uint32_t)uni((int)whdr[HX(t)].base);
    if (s == cap) {
        uint32_t ncap = cap < 4 ? 8 : cap * 2;
        if (w.pool_top + ncap > L.pool_cap) { w.status = MS_ST_ERR_POOL; return false; }
        uint32_t nb = w.pool_top;
        w.pool_top += ncap;
        for (uint32_t i = (uint32_t)w.lane; i < s; i += MS_WAVE) pool[nb + i] = pool[base + i];
        wave_fence();
        if (w.lane == 0) { whdr[HX(t)].base = nb; whdr[HX(t)].cap = ncap; }
        base = nb;
    }
    if (w.lane == 0) { pool[base + s] = make_int4(cref, blocker, (int)start, (int)size); whdr[HX(t)].size = s + 1; }
    wave_fence();
    return true;
}

DEV void repair_overflow(Wk& w, const MsShared& sh, const MsLayout& L) {
    uint32_t n = (uint32_t)uni((int)*w.ov_cnt);
    if (n == 0) return;
    wave_fence();
    Gp<const int32_t> ov = WK_PTR(int32_t, w, L, overflow);
    for (uint32_t e = 0; e < n && w.status == MS_ST_RUNNING; e++) {
        int t = uni(ov[3 * e]), c = uni(ov[3 * e + 1]), b = uni(ov[3 * e + 2]);
        const MsClauseHdr h = clause_hdr_of(w, sh, L, c);
        list_push_uniform(w, sh, L, t, c, b, (uint32_t)uni((int)h.start), (uint32_t)uni((int)h.size));
    }
    if (w.lane == 0) *w.ov_cnt = 0;
    lds_fence();
}

// ---- one watcher of a long / learnt clause ---------------------------------------------
#ifndef MS_LANE_SCAN
#define MS_LANE_SCAN 8   // clause literals the visiting lane examines itself (multiple of 4)
#endif
struct LongRes {
    int4 wt;            // watcher to keep (blocker possibly updated)
    bool live, keep, want, cf, deferred;
    int imp;            // implied literal if want
};

// Visit watcher `wt` of the false literal `fl`.  vbl / ww are the blocker's value and the clause's
// watched pair, loaded by the caller together with everything else the step needs.  All values are a
// snapshot taken before the step's commit.  Must be called by all 64 lanes (phase B is wave-cooperative).
template <bool LV>
// (LV) h0 / h1: the clause's first 8 literals, loaded by the caller together with the watched pair - the blocker's value is
// known at once there, so pair and literals are ONE round trip; with the assignment in the slab they are loaded here, only
// for the watchers whose blocker turned out not to be true.
DEV LongRes long_eval(Wk& w, const MsShared& sh, const MsLayout& L, int4 wt, bool live, int vbl, int2 ww, int fl, int g,
                      int4 h0 = make_int4(0, 0, 0, 0), int4 h1 = make_int4(0, 0, 0, 0)) {
    const MsClauseHdr ch = MsClauseHdr{(uint32_t)wt.z, (uint32_t)wt.w};
    LongRes R;
    R.wt = wt; R.live = live; R.keep = live; R.want = false; R.cf = false; R.deferred = false; R.imp = 0;
    bool scanning = false, need_tail = false;
    int other = 0, vo = MS_VAL_TRUE, r = -1;
    int size = (int)ch.size;
    Gp<const int32_t> cl = lits_base(w, sh, L, wt.x) + ch.start;
    uint32_t nl = 0;
    const bool eval = live && vbl != MS_VAL_TRUE;
    int r8 = -1;        // the first of the first MS_LANE_SCAN literals that is not false
    if (eval) {
        other = (ww.x == fl) ? ww.y : ww.x;
        // the other watch and the first MS_LANE_SCAN literals (16-byte loads) + their values, issued together
        int ls[MS_LANE_SCAN], vs[MS_LANE_SCAN];
#pragma unroll
        for (int k = 0; k < MS_LANE_SCAN; k += 4) {
            const int4 q = k < size ? (LV ? (k == 0 ? h0 : h1) : gld<int4>((Gp<const int4>)(cl + k))) : make_int4(fl, fl, fl, fl);
            ls[k] = q.x; ls[k + 1] = q.y; ls[k + 2] = q.z; ls[k + 3] = q.w;
        }
        {
            uint32_t wv[MS_LANE_SCAN];
#pragma unroll
            for (int u = 0; u < MS_LANE_SCAN; u++) { ls[u] = u < size ? ls[u] : fl; wv[u] = val_fetch<LV>(w, sh, L, ls[u]); }
            vo = val_decode<LV>(val_fetch<LV>(w, sh, L, other), other);
#pragma unroll
            for (int u = 0; u < MS_LANE_SCAN; u++) vs[u] = (ls[u] != fl && ls[u] != other) ? val_decode<LV>(wv[u], ls[u]) : MS_VAL_FALSE;
        }
#pragma unroll
        for (int u = MS_LANE_SCAN - 1; u >= 0; u--)
            if (vs[u] != MS_VAL_FALSE) r8 = ls[u];
    }
    // Both watches false and the other one is being propagated by another group in this very step: the lower group
    // handles the clause, the higher one re-queues.  Whole wave, one asking lane at a time (there are few): lane i < g holds
    // the literal group i propagates, a ballot says whether one of them is the asking lane's other watch.
    bool other_lower = false;
    {
        u64 am = ballot(eval && vo == MS_VAL_FALSE);
        if (am != 0) {
            const int bflv = w.lane < MS_MAX_GROUPS ? w.bfl[w.lane] : 0;
            for (; am != 0; am &= am - 1) {
                const int f = first_lane(am);
                const int of = bcast(other, f), gf = bcast(g, f);
                const u64 mm = ballot(w.lane < gf && bflv == of);
                if (w.lane == f) other_lower = mm != 0;
            }
        }
    }
    if (eval) {
        nl = 2;
        if (vo == MS_VAL_TRUE) R.wt.y = other;
        else if (other_lower) R.deferred = true;
        else {
            R.wt.y = other;
            scanning = true;
            r = r8;
            nl += (uint32_t)(size < MS_LANE_SCAN ? size : MS_LANE_SCAN);
            need_tail = r < 0 && size > MS_LANE_SCAN;
        }
    }
    // The header of the list a moved watcher goes to ({base, size, cap, -}) is requested as soon as the new watch is
    // known: for the lanes that found it among the first MS_LANE_SCAN literals that is before the tail scans, whose
    // round trips then hide this one.
    Gp<MsWatchHdr> whdr = WKA(MsWatchHdr, whdr);
    const bool push_a = scanning && r >= 0;
    int4 th = push_a ? gld<int4>((Gp<const int4>)&whdr[HX(r ^ 1)]) : make_int4(0, 0, 0, 0);
    // phase B: the tails of long clauses.  T clauses need one: the wave splits into segments of 64 >> ceil(log2 T) lanes,
    // one clause per segment, so that one round trip serves all of them (a learnt clause here has ~100 literals and two
    // thirds of the steps have a tail to scan: one clause at a time was 14 % of a lone worker's cycles).
    const u64 tm = ballot(need_tail);
    if (tm != 0) {
        const int T = popc64(tm);
        const int lgT = T > 1 ? 32 - __builtin_clz((unsigned)(T - 1)) : 0, SL = MS_WAVE >> lgT;
        const int myrank = popc64(tm & lanemask_lt(w.lane));
        lds_fence();
        if (need_tail) w.tl[myrank] = (uint32_t)w.lane;
        lds_fence();
        const int seg = w.lane >> (6 - lgT), sl2 = w.lane & (SL - 1);
        bool open = seg < T;
        const int own = open ? (int)w.tl[seg] : 0;
        const unsigned long long cp = (unsigned long long)cl;
        Gp<const int32_t> clf = (Gp<const int32_t>)(((unsigned long long)(uint32_t)bcast((int)(cp >> 32), own) << 32) |
                                                  (unsigned long long)(uint32_t)bcast((int)cp, own));
        const int szf = bcast(size, own), flf = bcast(fl, own), of = bcast(other, own);
        const u64 segmask = (SL == MS_WAVE ? ~0ull : ((1ull << SL) - 1ull)) << (seg * SL);
        int found = -1;
        // MS_TAIL_UNROLL chunks per round: their loads (and, with the assignment in LDS, their look-ups) are in flight
        // together, so a round costs one round trip whatever it covers
        for (int k0 = MS_LANE_SCAN; ballot(open) != 0; k0 += MS_TAIL_UNROLL * SL) {
            int lu[MS_TAIL_UNROLL];
            bool oku[MS_TAIL_UNROLL];
#pragma unroll
            for (int u = 0; u < MS_TAIL_UNROLL; u++) {
                const int k = k0 + u * SL + sl2;
                lu[u] = (open && k < szf) ? clf[k] : flf;
            }
            {
                uint32_t wv[MS_TAIL_UNROLL];
#pragma unroll
                for (int u = 0; u < MS_TAIL_UNROLL; u++) wv[u] = val_fetch<LV>(w, sh, L, lu[u]);
#pragma unroll
                for (int u = 0; u < MS_TAIL_UNROLL; u++) oku[u] = lu[u] != flf && lu[u] != of && val_decode<LV>(wv[u], lu[u]) != MS_VAL_FALSE;
            }
            int fnd = -1;
#pragma unroll
            for (int u = 0; u < MS_TAIL_UNROLL; u++) {
                const u64 om = ballot(oku[u]) & segmask;
                const int lf = bcast(lu[u], om ? first_lane(om) : w.lane);
                if (fnd < 0 && om) fnd = lf;
            }
            if (open) {
                if (fnd >= 0) { found = fnd; open = false; }
                else if (k0 + MS_TAIL_UNROLL * SL >= szf) open = false;
            }
        }
        const int res = bcast(found, myrank << (6 - lgT));
        if (need_tail) { r = res; nl += (uint32_t)(size - MS_LANE_SCAN); }
    }
    // phase C: move the watch, or report unit / conflict.  A moved watcher is appended to the list of its new literal;
    // lanes that append to the SAME list in this call rank themselves (a loop over the few distinct lists), so the
    // list's size is read and written with plain accesses: no read-modify-write leaves the wave.
    const bool push = scanning && r >= 0;
    const int t = r ^ 1;
    if (push && !push_a) th = gld<int4>((Gp<const int4>)&whdr[HX(t)]);
    int rank = 0, cnt = 0;
    for (u64 pm = ballot(push); pm != 0;) {
        const int tf = bcast(t, first_lane(pm));
        const u64 same = ballot(push && t == tf);
        if (push && t == tf) { rank = popc64(same & lanemask_lt(w.lane)); cnt = popc64(same); }
        pm &= ~same;
    }
    if (scanning) {
        if (r >= 0) {
            *(Gp<int2>)&WKA(MsClauseRec, wl)[wt.x] = make_int2(other, r);
            const uint32_t tsize = (uint32_t)th.y < (uint32_t)th.z ? (uint32_t)th.y : (uint32_t)th.z;   // (overshoot left by pushes that found the list full)
            const uint32_t pos = tsize + (uint32_t)rank;
            if (pos < (uint32_t)th.z) WKA(int4, pool)[(uint32_t)th.x + pos] = R.wt;
            else {
                uint32_t o = lds_add(w.ov_cnt, 1u);
                Gp<int32_t> ov = WK_PTR(int32_t, w, L, overflow);
                ov[3 * o] = t;
                ov[3 * o + 1] = wt.x;
                ov[3 * o + 2] = other;
            }
            if (rank == 0) whdr[HX(t)].size = tsize + (uint32_t)cnt;
            R.keep = false;
        } else if (vo == MS_VAL_FALSE) R.cf = true;
        else { R.want = true; R.imp = other; }
    }
    w.c_cl_lit += nl;
    return R;
}

// Flat work distribution for the REMAINDER of long lists: group g still has rem entries (after its first S), described
// by two per-group words a, b.  Leaders publish (rem, a, b) in LDS, lanes 0..MS_MAX_GROUPS-1 scan, and afterwards lane
// `item` finds its (group, index) by walking the scan.  Returns the total number of items.
// w.hist layout (M = MS_MAX_GROUPS): [0..M) rem, [M..2M) a, [2M..3M) b, [3M..4M) inclusive scan.
DEV int flat_setup(Wk& w, int G, int g, int sl, int rem, int a, int b) {
    LdsU32 fs = w.hist;
    lds_fence();
    if (sl == 0) { fs[g] = (uint32_t)rem; fs[MS_MAX_GROUPS + g] = (uint32_t)a; fs[2 * MS_MAX_GROUPS + g] = (uint32_t)b; }
    lds_fence();
    int x = w.lane < G ? (int)fs[w.lane] : 0;
    for (int o = 1; o < MS_MAX_GROUPS; o <<= 1) {
        int t = __shfl_up(x, o, 64);
        if (w.lane >= o) x += t;
    }
    if (w.lane < MS_MAX_GROUPS) fs[3 * MS_MAX_GROUPS + w.lane] = (uint32_t)x;
    lds_fence();
    return (int)fs[3 * MS_MAX_GROUPS + G - 1];
}
// prev / next: the items before this list and up to its end (the list's item range is [prev, next)).
// REGS: the whole scan is read in one go (slots of groups >= G hold the total, which no item reaches) and counted in
// registers - a walk with one dependent LDS read per group cost a lone worker 5 % of its cycles; the 16-waves-per-CU
// builds (128 VGPRs, already spilling) keep the walk: the 32 extra registers cost them 4 %.
template <bool REGS>
DEV void flat_item(const Wk& w, int G, int item, int& gg, int& idx, int& a, int& b, int& prev, int& next) {
    LdsU32 fs = w.hist;
    gg = 0;
    if (REGS) {
        int sc[MS_MAX_GROUPS];
#pragma unroll
        for (int k = 0; k < MS_MAX_GROUPS; k++) sc[k] = (int)fs[3 * MS_MAX_GROUPS + k];
#pragma unroll
        for (int k = 0; k < MS_MAX_GROUPS - 1; k++) gg += sc[k] <= item ? 1 : 0;
    } else {
        while (gg < G - 1 && (int)fs[3 * MS_MAX_GROUPS + gg] <= item) gg++;
    }
    prev = gg ? (int)fs[3 * MS_MAX_GROUPS + gg - 1] : 0;
    next = (int)fs[3 * MS_MAX_GROUPS + gg];
    idx = item - prev;
    a = (int)fs[MS_MAX_GROUPS + gg];
    b = (int)fs[2 * MS_MAX_GROUPS + gg];
}
template <bool REGS>
DEV void flat_item(const Wk& w, int G, int item, int& gg, int& idx, int& a, int& b) {
    int prev, next;
    flat_item<REGS>(w, G, item, gg, idx, a, b, prev, next);
}

// Unit propagation to fixpoint.  Returns true on conflict (w.confl_*).
template <bool LV>
DEV bool propagate(Wk& w, const MsShared& sh, const MsLayout& L) {
    w.confl_kind = 0;
    Gp<MsWatchHdr> whdr = WKA(MsWatchHdr, whdr);
    Gp<int4> pool = WKA(int4, pool);
    Gp<const MsClauseRec> wl = WKA(MsClauseRec, wl);
    Gp<const int32_t> trail = WKA(int32_t, trail);
    Gp<const int32_t> bin_lits = (Gp<const int32_t>)sh.bin_lits;
    Gp<const int2> tern_pairs = (Gp<const int2>)sh.tern_pairs;
    while (w.qhead < w.trail_n && w.status == MS_ST_RUNNING) {
        PROF_DECL
        // ---- split the wave into G groups of S lanes, one queue literal per group
        const int qlen = w.trail_n - w.qhead;
        int lg = qlen >= 32 ? 5 : qlen >= 16 ? 4 : (qlen >= 8 ? 3 : (qlen >= 4 ? 2 : (qlen >= 2 ? 1 : 0)));
        while ((1 << lg) > w.max_groups) lg--;
        const int G = 1 << lg, S = MS_WAVE >> lg;
        const int g = w.lane >> (6 - lg), sl = w.lane & (S - 1);
        const u64 gmask = (S == MS_WAVE ? ~0ull : ((1ull << S) - 1ull)) << (g * S);
        const int qbase = w.qhead;
        const int idx = qbase + g;
        const int p = (idx >= w.ring_lo) ? w.ring[idx & (MS_LDS_RING - 1)] : trail[idx];
        const MsWatchHdr wh = whdr[HX(p)];   // watch list + binary / ternary list headers: one line
        const int fl = p ^ 1;
        const uint32_t b0 = wh.bin_off, nb = wh.bin_n, t0 = wh.tern_off, nt = wh.tern_n;
        const uint32_t wb = wh.base;
        const int n = (int)wh.size;
        // round trip 1: the first chunk of all three lists
        const bool act_b = (uint32_t)sl < nb, act_t = (uint32_t)sl < nt;
        // (three unconditional loads - lanes beyond the end of a list read entry 0 of the array and drop it: under lane
        // conditions the compiler waited for the first two before it issued the third, a round trip of its own)
        const int q0r = bin_lits[act_b ? b0 + (uint32_t)sl : 0u];
        const int2 pr0r = gld<int2>(tern_pairs + (act_t ? t0 + (uint32_t)sl : 0u));
        const int4 wt0r = gld<int4>(pool + (sl < n ? wb + (uint32_t)sl : 0u));
        const int q0 = act_b ? q0r : 0;
        const int2 pr0 = act_t ? pr0r : make_int2(0, 0);
        const int4 wt0 = sl < n ? wt0r : make_int4(-1, 0, 0, 0);
        w.qhead += G;
        w.c_props += (uint32_t)G;
        w.c_steps++;
        // the false literals of all groups of this step (LDS), for the clash test in long_eval
        if (sl == 0) w.bfl[g] = fl;
        lds_fence();
        // round trip 2: ONE snapshot of every value the first chunks need, plus the watched pair of the clauses
        // whose blocker is not true (assignment in LDS: the blockers' values are known at once) or, with the
        // assignment in HBM, speculatively of every live watcher's clause
        const bool live0 = sl < n && wt0.x >= 0;
        // (inactive lanes look literal 0 up: four fetches, one wait)
        const int bl0 = live0 ? wt0.y : 0;
        const uint32_t xq = val_fetch<LV>(w, sh, L, q0), xb = val_fetch<LV>(w, sh, L, pr0.x), xc = val_fetch<LV>(w, sh, L, pr0.y),
                       xl = val_fetch<LV>(w, sh, L, bl0);
        const int vq = act_b ? val_decode<LV>(xq, q0) : MS_VAL_TRUE;
        const int vb = act_t ? val_decode<LV>(xb, pr0.x) : MS_VAL_TRUE;
        const int vc = act_t ? val_decode<LV>(xc, pr0.y) : MS_VAL_TRUE;
        const int vbl0 = live0 ? val_decode<LV>(xl, bl0) : MS_VAL_TRUE;
        const int2 ww0 = (live0 && (!LV || vbl0 != MS_VAL_TRUE)) ? gld<int2>((Gp<const int2>)&wl[wt0.x]) : make_int2(0, 0);
        int4 h00 = make_int4(0, 0, 0, 0), h01 = make_int4(0, 0, 0, 0);
        if (LV && live0 && vbl0 != MS_VAL_TRUE) {
            Gp<const int32_t> cl0 = lits_base(w, sh, L, wt0.x) + (uint32_t)wt0.z;
            h00 = *(Gp<const int4>)cl0;
            if (wt0.w > 4) h01 = *(Gp<const int4>)(cl0 + 4);
        }
        PROF_MARK(PF_OFF);
        // evaluate binary + ternary entries on the snapshot
        const bool cf_b = vq == MS_VAL_FALSE, want_b = vq == MS_VAL_UNDEF;
        const bool sat_t = vb == MS_VAL_TRUE || vc == MS_VAL_TRUE;
        const bool cf_t = !sat_t && vb == MS_VAL_FALSE && vc == MS_VAL_FALSE;
        const bool want_t = !sat_t && !cf_t && (vb == MS_VAL_FALSE || vc == MS_VAL_FALSE);
        const int imp_t = vb == MS_VAL_FALSE ? pr0.y : pr0.x;
        w.c_watch += (uint32_t)(popc64(ballot(act_b)) + popc64(ballot(act_t)) + popc64(ballot(live0)));
        PROF_MARK(PF_BIN);
        // first chunk of the watch lists (round trips 3..: other watch + clause literals, pushes)
        int j = 0, done = 0, defer_g = MS_MAX_GROUPS;
        LongRes R0 = long_eval<LV>(w, sh, L, wt0, live0, vbl0, ww0, fl, g, h00, h01);
        PROF_MARK(PF_LONG);
        // ONE commit for the whole first chunk: binary, ternary and watched-clause implications
        bool any_cf;
        {
            claims_clear(w);
            const int cb = claim_insert(w, want_b, q0);
            const int ct = claim_insert(w, want_t, imp_t);
            const int cl3 = claim_insert(w, R0.want, R0.imp);
            lds_fence();
            assign_winners<LV>(w, sh, L, cb == CLAIM_WON, q0, MS_REASON_BIN(fl));
            assign_winners<LV>(w, sh, L, ct == CLAIM_WON, imp_t, MS_REASON_TERN(t0 + sl));
            assign_winners<LV>(w, sh, L, cl3 == CLAIM_WON, R0.imp, wt0.x, (uint32_t)wt0.z, (uint32_t)wt0.w);
            lds_fence();
            const bool xb = cf_b || cb == CLAIM_LOST, xt = cf_t || ct == CLAIM_LOST, xl = R0.cf || cl3 == CLAIM_LOST;
            const u64 cm = ballot(xb || xt || xl);
            any_cf = cm != 0;
            if (any_cf) {
                const int f = first_lane(cm);
                const int kind = bcast(xb ? 2 : (xt ? 3 : 1), f);
                w.confl_kind = kind;
                w.confl_cref = bcast(wt0.x, f);
                w.confl_a = bcast(fl, f);
                // (kind 1: the clause's literal range instead of the unused literals b, c)
                w.confl_b = bcast(kind == 2 ? q0 : (kind == 1 ? wt0.z : pr0.x), f);
                w.confl_c = bcast(kind == 1 ? wt0.w : pr0.y, f);
            }
        }
        {   // in-place compaction of the first chunk of each group's watch list
            w.c_move += (uint32_t)popc64(ballot(R0.live && !R0.keep));
            const u64 km = ballot(R0.keep);
            wave_fence();
            const int d = popc64(km & gmask & lanemask_lt(w.lane));
            if (R0.keep && (d != sl || R0.wt.y != wt0.y)) pool[wb + d] = R0.wt;
            j = popc64(km & gmask);
            done = min(n, S);
            const u64 dm = ballot(R0.deferred);
            if (dm) defer_g = first_lane(dm) >> (6 - lg);
            repair_overflow(w, sh, L);
        }
        PROF_MARK(PF_TERN);
        bool lost;
        // ---- the rest of long binary lists, spread flat over all 64 lanes ----
        {
            const int rem = nb > (uint32_t)S ? (int)nb - S : 0;
            if (!any_cf && ballot(rem > 0) != 0) {
                const int total = flat_setup(w, G, g, sl, rem, (int)b0, fl);
                for (int base = 0; !any_cf && base < total; base += MS_WAVE) {
                    const int item = base + w.lane;
                    const bool act = item < total;
                    int gg = 0, i = 0, fb0 = 0, ffl = 0;
                    if (act) flat_item<LV>(w, G, item, gg, i, fb0, ffl);
                    const int q = act ? bin_lits[(uint32_t)fb0 + (uint32_t)S + (uint32_t)i] : 0;
                    const int v = act ? lit_value<LV>(w, sh, L, q) : MS_VAL_TRUE;
                    w.c_watch += (uint32_t)popc64(ballot(act));
                    commit_implications<LV>(w, sh, L, v == MS_VAL_UNDEF, q, MS_REASON_BIN(ffl), lost);
                    const u64 cm = ballot(v == MS_VAL_FALSE || lost);
                    if (cm) {
                        const int f = first_lane(cm);
                        any_cf = true;
                        w.confl_kind = 2;
                        w.confl_a = bcast(ffl, f);
                        w.confl_b = bcast(q, f);
                    }
                }
            }
        }
        // ---- the rest of long ternary lists, likewise ----------------------
        {
            const int rem = nt > (uint32_t)S ? (int)nt - S : 0;
            if (!any_cf && ballot(rem > 0) != 0) {
                const int total = flat_setup(w, G, g, sl, rem, (int)t0, fl);
                for (int base = 0; !any_cf && base < total; base += MS_WAVE) {
                    const int item = base + w.lane;
                    const bool act = item < total;
                    int gg = 0, i = 0, ft0 = 0, ffl = 0;
                    if (act) flat_item<LV>(w, G, item, gg, i, ft0, ffl);
                    const uint32_t e = (uint32_t)ft0 + (uint32_t)S + (uint32_t)i;
                    const int2 pr = act ? gld<int2>(tern_pairs + e) : make_int2(0, 0);
                    const uint32_t rb = val_fetch<LV>(w, sh, L, pr.x), rc = val_fetch<LV>(w, sh, L, pr.y);   // (inactive lanes: literal 0)
                    const int xb = act ? val_decode<LV>(rb, pr.x) : MS_VAL_TRUE;
                    const int xc = act ? val_decode<LV>(rc, pr.y) : MS_VAL_TRUE;
                    w.c_watch += (uint32_t)popc64(ballot(act));
                    const bool sat = xb == MS_VAL_TRUE || xc == MS_VAL_TRUE;
                    const bool cf = !sat && xb == MS_VAL_FALSE && xc == MS_VAL_FALSE;
                    const bool want = !sat && !cf && (xb == MS_VAL_FALSE || xc == MS_VAL_FALSE);
                    const int imp = xb == MS_VAL_FALSE ? pr.y : pr.x;
                    commit_implications<LV>(w, sh, L, want, imp, MS_REASON_TERN(e), lost);
                    const u64 cm = ballot(cf || lost);
                    if (cm) {
                        const int f = first_lane(cm);
                        any_cf = true;
                        w.confl_kind = 3;
                        w.confl_a = bcast(ffl, f);
                        w.confl_b = bcast(pr.x, f);
                        w.confl_c = bcast(pr.y, f);
                    }
                }
            }
        }
        // ---- the rest of long watch lists, spread flat over all 64 lanes like the binary and ternary remainders: ONE
        // chain of round trips (watchers -> watched pairs and clause heads -> tails -> target headers) per 64 watchers
        // of any list instead of one per list (a lone worker spent a quarter of its steps here, 1.8 chains each).  Kept
        // watchers are compacted in place per list - the lanes of a list are contiguous in a batch - and j / done of
        // every group live in LDS while the loop runs.
        {
            const int rem = n > S ? n - S : 0;
            if (!any_cf && w.status == MS_ST_RUNNING && ballot(rem > 0) != 0) {
                const int total = flat_setup(w, G, g, sl, rem, (int)wb, fl);
                if (sl == 0) { w.jd[g] = j; w.jd[MS_MAX_GROUPS + g] = done; }
                lds_fence();
                for (int base = 0; base < total && !any_cf && w.status == MS_ST_RUNNING; base += MS_WAVE) {
                    const int item = base + w.lane;
                    const bool act = item < total;
                    int gg = 0, ix = 0, fwb = 0, ffl = 0, prev = 0, next = 0;
                    if (act) flat_item<LV>(w, G, item, gg, ix, fwb, ffl, prev, next);
                    const int i = S + ix;
                    const int4 wt = act ? gld<int4>(pool + ((uint32_t)fwb + (uint32_t)i)) : make_int4(-1, 0, 0, 0);
                    const bool live = act && wt.x >= 0;
                    const int vbl = live ? lit_value<LV>(w, sh, L, wt.y) : MS_VAL_TRUE;
                    const int2 ww = (live && (!LV || vbl != MS_VAL_TRUE)) ? gld<int2>((Gp<const int2>)&wl[wt.x]) : make_int2(0, 0);
                    int4 h0 = make_int4(0, 0, 0, 0), h1 = make_int4(0, 0, 0, 0);
                    if (LV && live && vbl != MS_VAL_TRUE) {
                        Gp<const int32_t> clh = lits_base(w, sh, L, wt.x) + (uint32_t)wt.z;
                        h0 = *(Gp<const int4>)clh;
                        if (wt.w > 4) h1 = *(Gp<const int4>)(clh + 4);
                    }
                    w.c_watch += (uint32_t)popc64(ballot(live));
                    LongRes R = long_eval<LV>(w, sh, L, wt, live, vbl, ww, ffl, gg, h0, h1);
                    w.c_move += (uint32_t)popc64(ballot(R.live && !R.keep));
                    const u64 km = ballot(R.keep);
                    wave_fence();
                    // this lane's list occupies lanes [s0, s1) of the batch
                    const int s0 = prev > base ? prev - base : 0, s1 = next - base < MS_WAVE ? next - base : MS_WAVE;
                    const u64 segm = act ? (s1 >= MS_WAVE ? ~0ull : ((1ull << s1) - 1ull)) & ~((1ull << s0) - 1ull) : 0ull;
                    const int jold = act ? w.jd[gg] : 0;
                    {
                        const int d = jold + popc64(km & segm & lanemask_lt(w.lane));
                        if (R.keep && (d != i || R.wt.y != wt.y)) pool[(uint32_t)fwb + (uint32_t)d] = R.wt;
                    }
                    lds_fence();
                    if (act && w.lane == s0) {
                        w.jd[gg] = jold + popc64(km & segm);
                        w.jd[MS_MAX_GROUPS + gg] = S + ((next < base + MS_WAVE ? next : base + MS_WAVE) - prev);
                    }
                    lds_fence();
                    const u64 dm = ballot(R.deferred);
                    if (dm) defer_g = min(defer_g, bcast(gg, first_lane(dm)));
                    repair_overflow(w, sh, L);
                    commit_implications<LV>(w, sh, L, R.want, R.imp, wt.x, lost, (uint32_t)wt.z, (uint32_t)wt.w);
                    const u64 cm = ballot(R.cf || lost);
                    if (cm) {
                        const int f = first_lane(cm);
                        any_cf = true;
                        w.confl_kind = 1;
                        w.confl_cref = bcast(wt.x, f);
                        w.confl_a = bcast(ffl, f);
                        w.confl_b = bcast(wt.z, f);
                        w.confl_c = bcast(wt.w, f);
                    }
                }
                lds_fence();
                j = w.jd[g];
                done = w.jd[MS_MAX_GROUPS + g];
            }
        }
        // close each group's list: fully visited -> new size; interrupted -> tombstone the gap
        // between the compacted prefix and the first unvisited entry
        wave_fence();
        if (done == n) {
            if (sl == 0 && n > 0 && j != n) whdr[HX(p)].size = (uint32_t)j;
        } else {
            for (int x = j + sl; x < done; x += S) pool[wb + x] = make_int4(-1, 0, 0, 0);
        }
        wave_fence();
        PROF_MARK(PF_CLOSE);
        if (w.confl_kind) { w.qhead = w.trail_n; return true; }
        if (defer_g < G) {  // re-queue the deferred group's literal and everything after it
            w.qhead = qbase + defer_g;
            w.c_props -= (uint32_t)(G - defer_g);
            w.c_redo += (uint32_t)(G - defer_g);
        }
    }
    return false;
}

// ---- backtracking --------------------------------------------------------
// (LV) The bitmap of the variables assigned at the current decision level, from the trail: after a backjump the
// level we land on already has literals.  One coalesced pass over that level's segment.
DEV void cur_level_rebuild(Wk& w, const MsShared& sh, const MsLayout& L) {
    for (uint32_t i = (uint32_t)w.lane; i < ((sh.n_vars + 31) >> 5); i += MS_WAVE) w.lcur[i] = 0;
    lds_fence();
    if (w.n_levels > 0) {
        const int from = uni(WK_PTR(int32_t, w, L, trail_lim)[w.n_levels - 1]);
        for (int i = from + w.lane; i < w.trail_n; i += MS_WAVE) {
            const int v = WKA(int32_t, trail)[i] >> 1;
            lds_or(&w.lcur[v >> 5], 1u << (v & 31));
        }
        lds_fence();
    }
}
DEV void level_zero_rebuild(Wk& w, const MsShared& sh, const MsLayout& L) {
    for (uint32_t i = (uint32_t)w.lane; i < ((sh.n_vars + 31) >> 5); i += MS_WAVE) w.lzero[i] = 0;
    lds_fence();
    const int to = w.n_levels > 0 ? uni(WK_PTR(int32_t, w, L, trail_lim)[0]) : w.trail_n;
    for (int i = w.lane; i < to; i += MS_WAVE) {
        const int v = WKA(int32_t, trail)[i] >> 1;
        lds_or(&w.lzero[v >> 5], 1u << (v & 31));
    }
    lds_fence();
}

template <bool LV>
DEV void cancel_until(Wk& w, const MsShared& sh, const MsLayout& L, int lvl) {
    if (w.n_levels <= lvl) return;
    const int lim = uni(WK_PTR(int32_t, w, L, trail_lim)[lvl]);
    for (int i = lim + w.lane; i < w.trail_n; i += MS_WAVE) {
        const int l = WKA(int32_t, trail)[i];
        asg_clear<LV>(w, sh, L, l >> 1);   // (the saved phase was written with the assignment)
    }
    // Decision queue: "every live entry above vm_search is assigned" holds trivially for the front of the
    // queue.  The exact bound (the highest queue position among the variables just unassigned) would cost one
    // random variable-record line per backtracked literal; starting the next search at the front instead only
    // makes pick_branch_var skip the few analysed variables that stay assigned (they were just bumped there).
    w.vm_search = w.vm_end - 1;
    w.trail_n = lim;
    w.qhead = lim;
    w.n_levels = lvl;
    if (w.ring_lo > lim) w.ring_lo = lim;
    lds_fence();
    if (LV) cur_level_rebuild(w, sh, L);
}

template <bool LV>
DEV void new_decision_level(Wk& w, const MsShared& sh, const MsLayout& L) {
    if (w.lane == 0) WK_PTR(int32_t, w, L, trail_lim)[w.n_levels] = w.trail_n;
    w.n_levels++;
    if (LV) {   // nothing is assigned at the new level yet
        for (uint32_t i = (uint32_t)w.lane; i < ((sh.n_vars + 31) >> 5); i += MS_WAVE) w.lcur[i] = 0;
        lds_fence();
    }
}

// ---- decision queue (move-to-front as an append-only array) ---------------
// vm_order[0..vm_end) holds variables; the entry of v is live iff vm_pos[v] is its
// index.  Later index = more recently bumped.  vm_search: every live entry above
// it is assigned.
DEV void vm_compact(Wk& w, const MsShared& sh, const MsLayout& L) {
    Gp<int32_t> vm_order = WK_PTR(int32_t, w, L, vm_order);
    int j = 0;
    for (int i0 = 0; i0 < w.vm_end; i0 += MS_WAVE) {
        int i = i0 + w.lane;
        int v = -1;
        bool live = false;
        if (i < w.vm_end) { v = vm_order[i]; live = VMPOS[v] == i; }
        u64 m = ballot(live);
        wave_fence();
        if (live) {
            int d = j + popc64(m & lanemask_lt(w.lane));
            vm_order[d] = v;
            VMPOS[v] = d;
        }
        j += popc64(m);
        wave_fence();
    }
    w.vm_end = j;
    w.vm_search = j - 1;
}

template <bool LV>
DEV int pick_branch_var(Wk& w, const MsShared& sh, const MsLayout& L) {
    Gp<const int32_t> vm_order = WK_PTR(int32_t, w, L, vm_order);
    // Every candidate costs two random lines (its record for the liveness check, its assignment word).  The next
    // unassigned variable is usually among the first few entries (the search position follows the queue front),
    // so the first probe looks at 16 candidates only and only a miss widens to the whole wave.
    int width = 16;
    for (;;) {
        if (w.vm_search < 0) return -1;
        int idx = w.vm_search - w.lane;
        int v = -1;
        bool ok = false;
        if (w.lane < width && idx >= 0) {
            v = vm_order[idx];
            const int pos = VMPOS[v];                                  // (both look-ups in flight together)
            const uint32_t raw = val_fetch<LV>(w, sh, L, 2 * v);
            ok = pos == idx && val_decode<LV>(raw, 2 * v) == MS_VAL_UNDEF;
        }
        u64 m = ballot(ok);
        if (m) {
            int f = first_lane(m);
            w.vm_search -= f;
            return bcast(v, f);
        }
        w.vm_search -= width;
        width = MS_WAVE;
    }
}

// ---- recursive clause minimisation (LDS builds) -------------------------------------------------------
// A literal of the learnt clause is redundant if every other literal of its reason is in the clause, fixed at level 0,
// or itself implied by the clause in this sense (MiniSat's litRedundant).  Wave form: a node list (in the slab) takes every
// variable outside the clause that a reason leads to (each once: bitmap `lq`); the list is expanded 64 nodes at a time,
// breadth first - a node whose reason's other variables are all marked (clause / level 0 / proven) is marked in `lseen`
// like a clause literal, a decision, a variable of a level the clause does not touch (MiniSat's abstract levels) or one
// with a failed child goes to `lfail`; the rest waits for its children and is settled by a few passes over the list,
// youngest nodes first (reasons point backwards on the trail, so there are no cycles).  What is not settled by then
// counts as not implied: the result is a clause between the locally and the fully minimised one, always a consequence
// of the formula by the same resolution steps.  Bounded: MS_MIN_NODES nodes; a clause literal hands its reason's open
// variables to the list if that reason has at most MS_MIN_REASON literals (it is TESTED against reasons of any length), an
// inner node needs a reason of at most 8 (min_scan8).
#ifndef MS_DEEP_MIN
#define MS_DEEP_MIN 1      // 0: local minimisation only (A/B)
#endif
#ifndef MS_MIN_NODES
#define MS_MIN_NODES 4096   // (the list lives in the slab, in the array the clause-database reduction uses for its renumbering)
#endif
#ifndef MS_MIN_REASON
#define MS_MIN_REASON 32
#endif
#ifndef MS_BUMP_REASON_SIDE
#define MS_BUMP_REASON_SIDE 1
#endif
#ifndef MS_SORT_N
#define MS_SORT_N 2048      // analysed variables sorted by queue position before they are bumped (LDS words per wave)
#endif
#ifndef MS_MIN_PASSES
#define MS_MIN_PASSES 6
#endif
// state of variable y given its record: 0 = every other variable of its reason is marked, 1 = some are still open,
// 2 = cannot be implied.  `queue`: open children join the node list.
DEV uint32_t min_nodes_cap(const MsLayout& L) { return L.learnt_cap / 2 < MS_MIN_NODES ? L.learnt_cap / 2 : MS_MIN_NODES; }   // (node list + waiting list share the array)
DEV int min_scan(Wk& w, const MsShared& sh, const MsLayout& L, int y, const MsVarRec& vr, bool queue, int size_cap) {
    const int r = vr.reason;
    int st = 0;
    auto child = [&](int l) {
        const int c = l >> 1;
        const uint32_t bit = 1u << (c & 31);
        if (c == y || ((w.lseen[c >> 5] | w.lzero[c >> 5]) & bit)) return;
        if (w.lfail[c >> 5] & bit) { st = 2; return; }
        if (st < 1) st = 1;
        if (queue && !(w.lq[c >> 5] & bit)) {
            if (!(lds_or_rtn(&w.lq[c >> 5], bit) & bit)) {          // the lane that flips the bit appends the node
                const uint32_t k = lds_add(w.mcnt, 1u);
                if (k < min_nodes_cap(L)) WKA(int32_t, remap)[k] = c;
                else lds_and(&w.lq[c >> 5], ~bit);                  // list full: the child stays open for ever
            }
        }
    };
    if (r >= 0) {
        Gp<const int32_t> cl;
        int size = (int)vr.size;
        if (size > 0) cl = lits_base(w, sh, L, r) + vr.start;
        else clause_range(w, sh, L, r, cl, size);
        if (size > size_cap) return 2;
        for (int k = 0; k < size && st != 2; k += 4) {
            const int4 q4 = *(Gp<const int4>)(cl + k);
            child(q4.x);
            if (k + 1 < size) child(q4.y);
            if (k + 2 < size) child(q4.z);
            if (k + 3 < size) child(q4.w);
        }
    } else if (MS_IS_TERN_REASON(r)) {
        const int e = MS_TERN_REASON_ENTRY(r);
        const int2 pr = ((Gp<const int2>)sh.tern_pairs)[e];
        child(((Gp<const int32_t>)sh.tern_owner)[e] ^ 1); child(pr.x); child(pr.y);
    } else if (MS_IS_BIN_REASON(r)) child(MS_BIN_REASON_LIT(r));
    else return 2;      // a decision
    return st;
}
// The same for a NODE: only reasons of at most 8 literals (binary, ternary, short clauses - on the CPU restatement a cap of
// 8 / 16 / 24 literals on inner nodes costs nothing: the chains run through the totalizer's and the overlap families'
// short clauses), both halves loaded at once: one round trip per node instead of one per four literals.
DEV int min_scan8(Wk& w, const MsShared& sh, const MsLayout& L, int y, const MsVarRec& vr, bool queue) {
    const int r = vr.reason;
    int4 a = make_int4(0, 0, 0, 0), b = make_int4(0, 0, 0, 0);
    int n = 0;
    if (r >= 0) {
        if (vr.size == 0 || vr.size > 8) return 2;
        Gp<const int32_t> cl = lits_base(w, sh, L, r) + vr.start;
        a = *(Gp<const int4>)cl;
        if (vr.size > 4) b = *(Gp<const int4>)(cl + 4);
        n = (int)vr.size;
    } else if (MS_IS_TERN_REASON(r)) {
        const int e = MS_TERN_REASON_ENTRY(r);
        const int2 pr = ((Gp<const int2>)sh.tern_pairs)[e];
        a = make_int4(((Gp<const int32_t>)sh.tern_owner)[e] ^ 1, pr.x, pr.y, 0);
        n = 3;
    } else if (MS_IS_BIN_REASON(r)) { a.x = MS_BIN_REASON_LIT(r); n = 1; }
    else return 2;
    int st = 0;
    auto child = [&](int l) {
        const int c = l >> 1;
        const uint32_t bit = 1u << (c & 31);
        if (c == y || ((w.lseen[c >> 5] | w.lzero[c >> 5]) & bit)) return;
        if (w.lfail[c >> 5] & bit) { st = 2; return; }
        if (st < 1) st = 1;
        if (queue && !(w.lq[c >> 5] & bit)) {
            if (!(lds_or_rtn(&w.lq[c >> 5], bit) & bit)) {
                const uint32_t k = lds_add(w.mcnt, 1u);
                if (k < min_nodes_cap(L)) WKA(int32_t, remap)[k] = c;
                else lds_and(&w.lq[c >> 5], ~bit);
            }
        }
    };
    child(a.x);
    if (n > 1) child(a.y);
    if (n > 2) child(a.z);
    if (n > 3) child(a.w);
    if (n > 4) child(b.x);
    if (n > 5) child(b.y);
    if (n > 6) child(b.z);
    if (n > 7) child(b.w);
    return st;
}
DEV void deep_minimize_marks(Wk& w, const MsShared& sh, const MsLayout& L, Gp<const int32_t> learnt_buf, int n_out) {
    Gp<const int32_t> nodes = WKA(int32_t, remap);
    if (w.lane == 0) *w.mcnt = 0;
    lds_fence();
    // seeds: the open children of the clause literals' (short) reasons; the levels the clause touches
    uint32_t abs_levels = 0;
    for (int i0 = 0; i0 < n_out; i0 += MS_WAVE) {
        const int i = i0 + w.lane;
        if (i < n_out) {
            const int qv = learnt_buf[i] >> 1;
            const MsVarRec qr = VREC[qv];
            abs_levels |= 1u << (qr.level & 31);
            if (i > 0 && qr.reason != MS_REASON_NONE && !(qr.reason >= 0 && (qr.size == 0 || qr.size > MS_MIN_REASON)))
                (void)min_scan(w, sh, L, qv, qr, true, MS_MIN_REASON);
        }
    }
    for (int o = 32; o > 0; o >>= 1) abs_levels |= (uint32_t)__shfl_xor((int)abs_levels, o, 64);
    lds_fence();
    // expansion, breadth first; the nodes that have to wait for a child go to a second list
    Gp<int32_t> waiting = WKA(int32_t, remap) + min_nodes_cap(L);
    int head = 0, n_wait = 0;
    for (;;) {
        const int n = min((int)uni((int)*w.mcnt), (int)min_nodes_cap(L));
        if (head >= n) break;
        const int idx = head + w.lane;
        int z = 0, st = 0;
        if (idx < n) {
            z = nodes[idx];
            const MsVarRec zr = VREC[z];
            const bool dead = zr.reason == MS_REASON_NONE || !((abs_levels >> (zr.level & 31)) & 1u);
            st = dead ? 2 : min_scan8(w, sh, L, z, zr, true);
            if (st == 0) lds_or(&w.lseen[z >> 5], 1u << (z & 31));
            else if (st == 2) lds_or(&w.lfail[z >> 5], 1u << (z & 31));
        }
        const u64 wm = ballot(idx < n && st == 1);
        if (idx < n && st == 1) waiting[n_wait + popc64(wm & lanemask_lt(w.lane))] = z;
        n_wait += popc64(wm);
        head = min(head + MS_WAVE, n);
        lds_fence();
    }
    // settle the nodes that waited for their children: youngest first, until a pass changes nothing
    const int n = n_wait;
    for (int pass = 0; pass < MS_MIN_PASSES && n > 0; pass++) {
        bool changed = false;
        for (int i0 = ((n - 1) / MS_WAVE) * MS_WAVE; i0 >= 0; i0 -= MS_WAVE) {
            const int idx = i0 + w.lane;
            bool ch = false;
            if (idx < n) {
                const int z = waiting[idx];
                const uint32_t bit = 1u << (z & 31);
                if (!((w.lseen[z >> 5] | w.lfail[z >> 5]) & bit)) {
                    const int st = min_scan8(w, sh, L, z, VREC[z], false);
                    if (st == 0) { lds_or(&w.lseen[z >> 5], bit); ch = true; }
                    else if (st == 2) { lds_or(&w.lfail[z >> 5], bit); ch = true; }
                }
            }
            changed = changed || ballot(ch) != 0;
            lds_fence();
        }
        if (!changed) break;
    }
}
// the marks of the node list go again (clause literals are not in it; theirs are cleared with the analysis marks)
DEV void deep_minimize_clear(Wk& w, const MsLayout& L) {
    Gp<const int32_t> nodes = WKA(int32_t, remap);
    const int n = min((int)uni((int)*w.mcnt), (int)min_nodes_cap(L));
    for (int idx = w.lane; idx < n; idx += MS_WAVE) {
        const int z = nodes[idx];
        const uint32_t m = ~(1u << (z & 31));
        lds_and(&w.lseen[z >> 5], m); lds_and(&w.lfail[z >> 5], m); lds_and(&w.lq[z >> 5], m);
    }
    if (w.lane == 0) *w.mcnt = 0;
    lds_fence();
}

// ---- conflict analysis (first UIP) -----------------------------------------
struct Learnt { int n, bt_level; uint32_t lbd; };

// CLAIM: the lanes may hold literals of DIFFERENT clauses (batched resolution), so two of them may present the same
// variable: the mark is set with a fetch-or and only the lane that flipped the bit counts it.
template <bool LV, bool CLAIM = false>
DEV void analyze_visit(Wk& w, const MsShared& sh, const MsLayout& L, Gp<MsVarRec> vrec, Gp<int32_t> toclear, Gp<int32_t> learnt_buf, bool act, int q, int dl,
                       int& path_c, int& n_out, int& n_clear) {
    int v = q >> 1;
    bool fresh = false, cur = false;
    if (LV) {   // everything analysis asks about a literal is a bit in LDS: marked? level 0? current level?
        if (act) {
            const uint32_t bit = 1u << (v & 31);
            if (CLAIM) fresh = !(w.lzero[v >> 5] & bit) && !(w.lseen[v >> 5] & bit) && !(lds_or_rtn(&w.lseen[v >> 5], bit) & bit);
            else fresh = !(w.lseen[v >> 5] & bit) && !(w.lzero[v >> 5] & bit);
            cur = fresh && (w.lcur[v >> 5] & bit);
        }
    } else {
        bool atdl = false;
        if (act) {
            const MsVarRec vr = vrec[v];
            fresh = !vr.seen && vr.level > 0;
            atdl = vr.level >= dl;
        }
        if (CLAIM) {    // the marks are bytes in the records here: an exact claim set (the one BCP's commit uses) keeps one of the lanes that present the same variable
            claims_clear(w);
            fresh = claim_insert(w, fresh, q) == CLAIM_WON;
            lds_fence();
        }
        cur = fresh && atdl;
    }
    u64 fm = ballot(fresh), cm = ballot(cur);
    u64 lm = fm & ~cm;
    if (fresh) {
        if (!(LV && CLAIM)) seen_set<LV>(w, sh, L, v);
        toclear[n_clear + popc64(fm & lanemask_lt(w.lane))] = v;
        if (!cur) learnt_buf[n_out + popc64(lm & lanemask_lt(w.lane))] = q;
    }
    n_clear += popc64(fm);
    n_out += popc64(lm);
    path_c += popc64(cm);
}

// One long clause, 64 literals per round (wave-cooperative).
template <bool LV>
DEV void analyze_visit_clause(Wk& w, const MsShared& sh, const MsLayout& L, Gp<MsVarRec> vrec, Gp<int32_t> toclear, Gp<int32_t> learnt_buf,
                              Gp<const int32_t> cl, int size, int skip_var, int dl, int& path_c, int& n_out, int& n_clear) {
    for (int k0 = 0; k0 < size; k0 += MS_WAVE) {
        const int k = k0 + w.lane;
        const int q = k < size ? cl[k] : 0;
        analyze_visit<LV>(w, sh, L, vrec, toclear, learnt_buf, k < size && (q >> 1) != skip_var, q, dl, path_c, n_out, n_clear);
    }
}

// (LDS builds) The backward walk of first-UIP analysis over the trail, 64 entries at a time (one coalesced load, lane
// i holds position chunk_hi - i; the marks are bits in LDS, so the whole chunk is tested at once).  BATCHED: as long
// as more marked literals of the current level are open than this chunk holds (path_c > marked-in-chunk), none of the
// chunk's marked literals can be the first UIP - at least one open literal lies further down the trail - so ALL of them
// are resolved in one round: every lane marks the literals of its own literal's reason (records and reason heads of all
// of them fetched together; binary, ternary and short long reasons are then in registers).  The marks are a set union,
// so the order of the resolutions does not matter; reasons point backwards on the trail, so what a round newly marks
// inside the chunk is picked up by the next round.  Once the chunk holds every open literal, the walk is sequential
// (the last open one is the UIP and must not be resolved).  Round 2 resolved one literal per iteration (~160 per
// conflict at ~1 us each).  Returns the UIP literal, or -1 (internal error).
template <bool LV>
DEV int analyze_walk(Wk& w, const MsShared& sh, const MsLayout& L, Gp<MsVarRec> vrec, Gp<int32_t> toclear, Gp<int32_t> learnt_buf,
                         Gp<uint32_t> lc_lbd, int dl, int& path_c, int& n_out, int& n_clear) {
    int index = w.trail_n - 1;
    for (;;) {
        if (index < 0) { w.status = MS_ST_ERR_INTERNAL; return -1; }
        const int chunk_hi = index;
        const int pos = chunk_hi - w.lane;
        const int my = pos >= 0 ? WKA(int32_t, trail)[pos] : 0;
        const int myv = my >> 1;
        bool have = false;          // this lane's record and reason head are in registers
        MsVarRec rec = MsVarRec{0, MS_REASON_NONE, 0, 0, 0, 0};
        int4 l0 = make_int4(0, 0, 0, 0), l1 = make_int4(0, 0, 0, 0);
        for (;;) {
            lds_fence();
            bool mine;
            if (LV) mine = pos >= 0 && pos <= index && ((w.lseen[myv >> 5] >> (myv & 31)) & 1u);
            else {      // marks in the records: every lane of the chunk still in play re-reads its own (one round trip per round)
                wave_fence();
                const bool inr = pos >= 0 && pos <= index;
                if (inr) rec = vrec[myv];
                mine = inr && rec.seen;
            }
            const u64 sm = ballot(mine);
            if (sm == 0) break;
            const int cnt = popc64(sm);
            if (mine && !have) {    // records, then reason heads, of every marked literal of the chunk not fetched yet
                if (LV) rec = vrec[myv];
                const int rr = rec.reason;
                if (rr >= 0 && rec.size > 0) {
                    Gp<const int32_t> cl = lits_base(w, sh, L, rr) + rec.start;
                    l0 = *(Gp<const int4>)cl;
                    if (rec.size > 4) l1 = *(Gp<const int4>)(cl + 4);
                } else if (rr < 0 && MS_IS_TERN_REASON(rr)) {
                    const int e = MS_TERN_REASON_ENTRY(rr);
                    const int2 tp = ((Gp<const int2>)sh.tern_pairs)[e];
                    l0 = make_int4(((Gp<const int32_t>)sh.tern_owner)[e] ^ 1, tp.x, tp.y, 0);
                }
                have = true;
            }
            const bool batch = path_c > cnt;
            const int f = first_lane(sm);
            const bool me = batch ? mine : w.lane == f;      // the lanes whose literal is resolved in this round
#ifdef MS_PROFILE
            w.prof[PF_RES_STEPS] += (u64)(batch ? cnt : 1);
#endif
            if (batch) path_c -= cnt;
            else {
                index = chunk_hi - f - 1;
                path_c--;
                if (path_c <= 0) {      // the first UIP: not resolved
                    if (me) seen_clr<LV>(w, sh, L, myv);
                    lds_fence();
                    wave_fence();
                    return bcast(my, f);
                }
            }
            // the reasons: up to 8 literals per lane from registers, longer clauses afterwards one at a time
            const int r = rec.reason;
            int nl = 0;
            bool big = false;
            if (me) {
                if (r >= 0) { if (rec.size > 0 && rec.size <= 8) nl = (int)rec.size; else big = true; }
                else if (MS_IS_TERN_REASON(r)) nl = 3;
                else if (MS_IS_BIN_REASON(r)) { nl = 1; l0.x = MS_BIN_REASON_LIT(r); }
                else { big = false; nl = -1; }      // a marked decision above the UIP cannot be
                if (r >= 0 && (uint32_t)r >= sh.n_orig) lc_lbd[(uint32_t)r - sh.n_orig] |= 0x80000000u;   // used
            }
            if (ballot(me && nl < 0)) { w.status = MS_ST_ERR_INTERNAL; return -1; }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int q = j < 4 ? (j == 0 ? l0.x : (j == 1 ? l0.y : (j == 2 ? l0.z : l0.w)))
                                    : (j == 4 ? l1.x : (j == 5 ? l1.y : (j == 6 ? l1.z : l1.w)));
                const bool act = me && j < nl && (q >> 1) != myv;
                if (ballot(me && j < nl) == 0) break;
                analyze_visit<LV, true>(w, sh, L, vrec, toclear, learnt_buf, act, q, dl, path_c, n_out, n_clear);
            }
            for (u64 bm = ballot(me && big); bm != 0; bm &= bm - 1) {
                const int fb = first_lane(bm);
                const int cref = bcast(r, fb);
                Gp<const int32_t> cl;
                int size = bcast((int)rec.size, fb);
                if (size > 0) cl = lits_base(w, sh, L, cref) + (uint32_t)bcast((int)rec.start, fb);
                else clause_range(w, sh, L, cref, cl, size);
                analyze_visit_clause<LV>(w, sh, L, vrec, toclear, learnt_buf, cl, size, bcast(myv, fb), dl, path_c, n_out, n_clear);
            }
            // the resolved literals' own marks go LAST: while they stand, no reason of this round can mark them anew
            lds_fence();
            wave_fence();
            if (me) seen_clr<LV>(w, sh, L, myv);
        }
        index = chunk_hi - MS_WAVE;     // nothing marked is left in this chunk at or below `index`
    }
}

template <bool LV>
DEV Learnt analyze(Wk& w, const MsShared& sh, const MsLayout& L, uint32_t stamp) {
    Gp<MsVarRec> vrec = VREC;
    Gp<int32_t> toclear = WK_PTR(int32_t, w, L, toclear);
    Gp<int32_t> learnt_buf = WK_PTR(int32_t, w, L, learnt_buf);
    Gp<uint32_t> lc_lbd = WK_PTR(uint32_t, w, L, lc_lbd);
    int path_c = 0, p = -1, n_out = 1, n_clear = 0;
    const int dl = w.n_levels;
    int kind = w.confl_kind, cref = w.confl_cref, ba = w.confl_a, bb = w.confl_b, bc = w.confl_c;
    if (LV || MS_BATCH_WALK) {       // the conflict clause, then the batched walk (analyze_walk)
        if (kind == 1) {
            Gp<const int32_t> cl;
            int size = bc;     // kind 1: (bb, bc) = the clause's literal range when known (size 0: look it up)
            if (size > 0) cl = lits_base(w, sh, L, cref) + (uint32_t)bb;
            else clause_range(w, sh, L, cref, cl, size);
            if ((uint32_t)cref >= sh.n_orig && w.lane == 0) lc_lbd[cref - sh.n_orig] |= 0x80000000u;  // used
            analyze_visit_clause<LV>(w, sh, L, vrec, toclear, learnt_buf, cl, size, -1, dl, path_c, n_out, n_clear);
        } else {
            const int q = w.lane == 0 ? ba : (w.lane == 1 ? bb : bc);
            analyze_visit<LV>(w, sh, L, vrec, toclear, learnt_buf, w.lane < kind, q, dl, path_c, n_out, n_clear);
        }
        p = analyze_walk<LV>(w, sh, L, vrec, toclear, learnt_buf, lc_lbd, dl, path_c, n_out, n_clear);
        if (p < 0) return Learnt{0, 0, 0};
    } else {
        // MS_BATCH_WALK = 0 (A/B reference, marks in the records): one literal resolved per iteration - its reason's
        // literals visited, then the trail walked back to the next marked literal.  A chunk of 64 trail entries stays in
        // registers across iterations (lane i holds position chunk_hi - i); the 16 positions below `index` are tested first
        // (each costs a random record line and the next marked literal is usually close), the rest of the chunk only when
        // those miss.  The lane that finds p has loaded p's whole record - its reason and where the reason's literals are.
        int index = w.trail_n - 1, chunk_hi = -1, chunk_l = 0;
        MsVarRec c_rec = MsVarRec{0, MS_REASON_NONE, 0, 0, 0, 0};
        for (;;) {
            if (kind == 1) {
                Gp<const int32_t> cl;
                int size = bc;
                if (size > 0) cl = lits_base(w, sh, L, cref) + (uint32_t)bb;
                else clause_range(w, sh, L, cref, cl, size);
                if ((uint32_t)cref >= sh.n_orig && w.lane == 0) lc_lbd[cref - sh.n_orig] |= 0x80000000u;  // used
                for (int k0 = 0; k0 < size; k0 += MS_WAVE) {
                    const int k = k0 + w.lane;
                    const int q = k < size ? cl[k] : 0;
                    analyze_visit<LV>(w, sh, L, vrec, toclear, learnt_buf, k < size && q != p, q, dl, path_c, n_out, n_clear);
                }
            } else {
                const int q = w.lane == 0 ? ba : (w.lane == 1 ? bb : bc);
                analyze_visit<LV>(w, sh, L, vrec, toclear, learnt_buf, w.lane < kind && q != p, q, dl, path_c, n_out, n_clear);
            }
            wave_fence();
            bool wide = false;
            for (;;) {
                if (chunk_hi < 0 || index > chunk_hi || index <= chunk_hi - MS_WAVE) {
                    chunk_hi = index;
                    const int i = chunk_hi - w.lane;
                    chunk_l = i >= 0 ? WKA(int32_t, trail)[i] : 0;
                    wide = false;
                }
                const int pos = chunk_hi - w.lane;
                const bool in = pos >= 0 && pos <= index && (wide || pos > index - 16);
                if (in) c_rec = vrec[chunk_l >> 1];
                const u64 m = ballot(in && c_rec.seen);
                if (m) {
                    const int f = first_lane(m);
                    index = chunk_hi - f;
                    p = bcast(chunk_l, f);
                    break;
                }
                if (!wide && index - 16 > chunk_hi - MS_WAVE) { wide = true; continue; }    // the rest of this chunk
                index = chunk_hi - MS_WAVE;
                if (index < 0) { w.status = MS_ST_ERR_INTERNAL; return Learnt{0, 0, 0}; }
            }
            index--;
#ifdef MS_PROFILE
            w.prof[PF_N]++;                    // resolution steps (diagnostic build)
#endif
            const int v = p >> 1;
            const int fp = chunk_hi - (index + 1);      // the lane that holds p and its record
            const int r = bcast(c_rec.reason, fp);
            wave_fence();
            if (w.lane == 0) seen_clr<LV>(w, sh, L, v);
            lds_fence();
            path_c--;
            if (path_c <= 0) break;
            if (r >= 0) { kind = 1; cref = r; bb = bcast((int)c_rec.start, fp); bc = bcast((int)c_rec.size, fp); }
            else if (MS_IS_TERN_REASON(r)) {
                const int e = MS_TERN_REASON_ENTRY(r);
                const int2 pr = ((Gp<const int2>)sh.tern_pairs)[e];
                kind = 3; ba = uni(((Gp<const int32_t>)sh.tern_owner)[e]) ^ 1; bb = uni(pr.x); bc = uni(pr.y);
            }
            else if (MS_IS_BIN_REASON(r)) { kind = 2; ba = p; bb = MS_BIN_REASON_LIT(r); }
            else { w.status = MS_ST_ERR_INTERNAL; return Learnt{0, 0, 0}; }
        }
    }
    wave_fence();
    if (w.lane == 0) learnt_buf[0] = p ^ 1;
    wave_fence();
    // ---- recursive minimisation (MiniSat's litRedundant), LDS builds: first the variables OUTSIDE the clause whose
    // assignment the clause's own literals (and level 0) imply get marked like clause literals - then the local test
    // below is the recursive one.  Measured on the CPU restatement (rect 26x26 k = 10): 4.2e5 conflicts and learnt
    // clauses of 36 literals with it, 8.0e5 and 177 with the local test alone.
#ifdef MS_PROFILE
    u64 tmin_ = __builtin_readcyclecounter();
#endif
    if (LV && MS_DEEP_MIN && n_out > 2) deep_minimize_marks(w, sh, L, learnt_buf, n_out);
#ifdef MS_PROFILE
    if (LV && MS_DEEP_MIN && n_out > 2) { w.prof[PF_MIN_NODES] += (u64)uni((int)*w.mcnt); w.prof[PF_MIN_CALLS]++; }
    { const u64 t_ = __builtin_readcyclecounter(); w.prof[PF_MIN_DEEP] += t_ - tmin_; tmin_ = t_; }
#endif
    // ---- local minimisation: drop a literal whose reason's other literals are all seen / level 0.
    // One learnt literal per lane; a long reason is read four literals per load (clauses start 16-byte aligned)
    // and a literal's level is only fetched when it is not marked.
    int j = 1;
    for (int i0 = 1; i0 < n_out; i0 += MS_WAVE) {
        int i = i0 + w.lane;
        bool act = i < n_out, keep = act;
        int q = act ? learnt_buf[i] : 0;
        if (act) {
            const int qv = q >> 1;
            const MsVarRec qr = VREC[qv];
            const int r = qr.reason;
            auto implied = [&](int l) {
                const int lv = l >> 1;
                if (lv == qv) return true;
                if (LV) return ((w.lseen[lv >> 5] | w.lzero[lv >> 5]) >> (lv & 31) & 1u) != 0;
                return seen_get<LV>(w, sh, L, lv) || VREC[lv].level == 0;
            };
            if (r >= 0) {
                Gp<const int32_t> cl;
                int size = (int)qr.size;
                if (size > 0) cl = lits_base(w, sh, L, r) + qr.start;
                else clause_range(w, sh, L, r, cl, size);
                bool red = true;
                for (int k = 0; k < size && red; k += 4) {
                    const int4 q4 = *(Gp<const int4>)(cl + k);
                    red = implied(q4.x) && (k + 1 >= size || implied(q4.y)) && (k + 2 >= size || implied(q4.z)) &&
                          (k + 3 >= size || implied(q4.w));
                }
                keep = !red;
            } else if (MS_IS_TERN_REASON(r)) {
                const int e = MS_TERN_REASON_ENTRY(r);
                const int2 pr = ((Gp<const int2>)sh.tern_pairs)[e];
                keep = !(implied(((Gp<const int32_t>)sh.tern_owner)[e] ^ 1) && implied(pr.x) && implied(pr.y));
            } else if (MS_IS_BIN_REASON(r)) {
                keep = !implied(MS_BIN_REASON_LIT(r));
            }
        }
        u64 km = ballot(keep);
        wave_fence();
        if (keep) learnt_buf[j + popc64(km & lanemask_lt(w.lane))] = q;
        j += popc64(km);
        wave_fence();
    }
    n_out = j;
    if (LV && MS_DEEP_MIN) deep_minimize_clear(w, L);
#ifdef MS_PROFILE
    w.prof[PF_MIN_LOCAL] += __builtin_readcyclecounter() - tmin_;
#endif
    // ---- backjump level = max level among learnt_buf[1..), moved to position 1
    int bt = 0;
    if (n_out > 1) {
        int best = -1, best_i = 0x7fffffff;
        for (int i = 1 + w.lane; i < n_out; i += MS_WAVE) {
            int lv = VREC[learnt_buf[i] >> 1].level;
            if (lv > best) { best = lv; best_i = i; }
        }
        int mx = wave_max(best);
        int cand = (best == mx) ? best_i : 0x7fffffff;
        int mi = -wave_max(-cand);
        bt = mx;
        if (w.lane == 0 && mi != 1) {
            int t = learnt_buf[mi];
            learnt_buf[mi] = learnt_buf[1];
            learnt_buf[1] = t;
        }
        wave_fence();
    }
    // ---- LBD: number of distinct decision levels
    uint32_t lbd = 0;
    {
        Gp<uint32_t> lvl_stamp = WK_PTR(uint32_t, w, L, lvl_stamp);
        const uint32_t base = w.lvl_stamp_ctr;
        for (int i0 = 0; i0 < n_out; i0 += MS_WAVE) {
            int i = i0 + w.lane;
            bool act = i < n_out;
            int lv = act ? VREC[learnt_buf[i] >> 1].level : 0;
            uint32_t id = base + 1 + (uint32_t)i;
            bool cand = act && lvl_stamp[lv] <= base;
            wave_fence();
            if (cand) lvl_stamp[lv] = id;
            wave_fence();
            bool won = cand && lvl_stamp[lv] == id;
            lbd += (uint32_t)popc64(ballot(won));
            wave_fence();
        }
        uint32_t nb = base + (uint32_t)n_out + 1;
        if (nb > 0xf0000000u) {  // stamp space exhausted: reset
            for (uint32_t i = (uint32_t)w.lane; i < sh.n_vars + 2; i += MS_WAVE) lvl_stamp[i] = 0;
            nb = 0;
        }
        w.lvl_stamp_ctr = nb;
    }
    // ---- reason side (LDS builds with the sorted bump): the other variables of the short reasons of the learnt clause's
    // literals are bumped too (CaDiCaL's "bump reason side", one level deep).  On the CPU restatement with the sorted
    // move-to-front queue: rect 26 k = 10 3.5e5 against 3.9e5 conflicts, rect 28 k = 11 4.1e5 against 5.4e5.
    if (LV && MS_BUMP_REASON_SIDE && w.sort_n > 0) {
        // (`lq`, all zero outside the recursive minimisation, says here what is in `toclear` already: the resolved
        // literals' analysis marks are gone by now)
        for (int i = w.lane; i < n_clear; i += MS_WAVE) { const int v = toclear[i]; lds_or(&w.lq[v >> 5], 1u << (v & 31)); }
        lds_fence();
        const int n_clear0 = n_clear;
        for (int i0 = 1; i0 < n_out; i0 += MS_WAVE) {
            const int i = i0 + w.lane;
            int4 a = make_int4(0, 0, 0, 0), b = make_int4(0, 0, 0, 0);
            int n = 0, qv = -1;
            if (i < n_out) {
                qv = learnt_buf[i] >> 1;
                const MsVarRec vr = VREC[qv];
                const int r = vr.reason;
                if (r >= 0) {
                    if (vr.size > 0 && vr.size <= 8) {
                        Gp<const int32_t> cl = lits_base(w, sh, L, r) + vr.start;
                        a = *(Gp<const int4>)cl;
                        if (vr.size > 4) b = *(Gp<const int4>)(cl + 4);
                        n = (int)vr.size;
                    }
                } else if (MS_IS_TERN_REASON(r)) {
                    const int e = MS_TERN_REASON_ENTRY(r);
                    const int2 pr = ((Gp<const int2>)sh.tern_pairs)[e];
                    a = make_int4(((Gp<const int32_t>)sh.tern_owner)[e] ^ 1, pr.x, pr.y, 0);
                    n = 3;
                } else if (MS_IS_BIN_REASON(r)) { a.x = MS_BIN_REASON_LIT(r); n = 1; }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (ballot(j < n) == 0) break;
                const int c = (j < 4 ? (j == 0 ? a.x : (j == 1 ? a.y : (j == 2 ? a.z : a.w)))
                                     : (j == 4 ? b.x : (j == 5 ? b.y : (j == 6 ? b.z : b.w)))) >> 1;
                bool fresh = false;
                if (j < n && c != qv) {
                    const uint32_t bit = 1u << (c & 31);
                    fresh = !((w.lq[c >> 5] | w.lzero[c >> 5]) & bit) && !(lds_or_rtn(&w.lq[c >> 5], bit) & bit);
                }
                const u64 fm = ballot(fresh);
                if (fresh) toclear[n_clear + popc64(fm & lanemask_lt(w.lane))] = c;
                n_clear += popc64(fm);
            }
        }
        lds_fence();
        for (int i = w.lane; i < n_clear; i += MS_WAVE) { const int v = toclear[i]; lds_and(&w.lq[v >> 5], ~(1u << (v & 31))); }
        (void)n_clear0;
        lds_fence();
    }
    // ---- clear marks and bump the analysed variables to the front of the queue.  In WHICH ORDER they go there decides how
    // good the search is on the hard bounds: sorted by their previous queue position (CaDiCaL's order: the analysed
    // variables keep their relative order) against the order analysis met them, on the CPU restatement with a
    // move-to-front queue in place of its VSIDS heap: rect 26 k = 10 3.9e5 against 6.8e5 conflicts, rect 28 k = 11 5.4e5
    // against 1.67e6 (VSIDS: 4.2e5 / 4.4e5).  Round 1 compared the two on rect 20 / 24 only and saw no difference.  So:
    // positions to LDS, bitonic sort (wave-wide, 64 pairs per pass), re-read the variables in that order.  More than
    // MS_SORT_N analysed variables (rare) or a build without the sort buffer: the order analysis met them.
    if (w.vm_end + n_clear > (int)L.vm_cap) vm_compact(w, sh, L);
    {
        Gp<int32_t> vm_order = WK_PTR(int32_t, w, L, vm_order);
        if (w.sort_n >= 64 && n_clear > 1 && n_clear <= w.sort_n) {
            int N = 64;
            while (N < n_clear) N <<= 1;
            for (int i = w.lane; i < N; i += MS_WAVE) {
                uint32_t key = 0xffffffffu;
                if (i < n_clear) { const int v = toclear[i]; seen_clr<LV>(w, sh, L, v); key = (uint32_t)VMPOS[v]; }
                w.sortbuf[i] = key;
            }
            lds_fence();
            for (int k = 2; k <= N; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = w.lane; t < N / 2; t += MS_WAVE) {
                        const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), b = a + j;      // (j is a power of two)
                        const uint32_t x = w.sortbuf[a], y = w.sortbuf[b];
                        const bool asc = (a & k) == 0;
                        if ((x > y) == asc) { w.sortbuf[a] = y; w.sortbuf[b] = x; }
                    }
                    lds_fence();
                }
            for (int i = w.lane; i < n_clear; i += MS_WAVE) {
                const int v = vm_order[w.sortbuf[i]];      // (live entry: vm_order[vm_pos[v]] == v)
                vm_order[w.vm_end + i] = v;
                VMPOS[v] = w.vm_end + i;
            }
        } else {
            for (int i = w.lane; i < n_clear; i += MS_WAVE) {
                int v = toclear[i];
                seen_clr<LV>(w, sh, L, v);
                vm_order[w.vm_end + i] = v;
                VMPOS[v] = w.vm_end + i;
            }
        }
        w.vm_end += n_clear;
    }
    lds_fence();
    wave_fence();
    return Learnt{n_out, bt, lbd};
}

// ---- watch pool garbage collection ---------------------------------------------------
// Lists that outgrow their slot are moved to the top of a bump pool and leave a
// hole behind.  The rebuild lays every list out again, densely, straight from the
// per-clause watched-literal pairs (no read of the old pool): count, exclusive scan
// over the 2*n_vars lists (wave prefix sums), fill.  Runs at a propagation fixpoint.
DEV void rebuild_watches(Wk& w, const MsShared& sh, const MsLayout& L) {
    Gp<MsWatchHdr> whdr = WKA(MsWatchHdr, whdr);
    Gp<int4> pool = WKA(int4, pool);
    Gp<const MsClauseRec> wl = WKA(MsClauseRec, wl);
    const uint32_t nlist = 2 * sh.n_vars;
    const uint32_t ncl = sh.n_orig + w.n_learnts;
    for (uint32_t t = (uint32_t)w.lane; t < nlist; t += MS_WAVE) whdr[t].size = 0;
    wave_fence();
    for (uint32_t c = (uint32_t)w.lane; c < ncl; c += MS_WAVE) {
        const int2 ww = make_int2(wl[c].w0, wl[c].w1);
        atomicAdd(&whdr[HX(ww.x ^ 1)].size, 1u);
        atomicAdd(&whdr[HX(ww.y ^ 1)].size, 1u);
    }
    wave_fence();
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < nlist; t0 += MS_WAVE) {
        uint32_t t = t0 + (uint32_t)w.lane;
        uint32_t sz = t < nlist ? whdr[t].size : 0;
        uint32_t cap = t < nlist ? sz + (sz >> 1) + 2 : 0;
        uint32_t incl = cap;
        for (int o = 1; o < MS_WAVE; o <<= 1) {
            uint32_t x = (uint32_t)__shfl_up((int)incl, o, 64);
            if (w.lane >= o) incl += x;
        }
        wave_fence();
        if (t < nlist) { whdr[t].base = run + incl - cap; whdr[t].size = 0; whdr[t].cap = cap; }
        run += (uint32_t)bcast((int)incl, 63);
    }
    if (run > L.pool_cap) { w.status = MS_ST_ERR_POOL; return; }
    w.pool_top = run;
    wave_fence();
    for (uint32_t c = (uint32_t)w.lane; c < ncl; c += MS_WAVE) {
        const MsClauseRec cr = wl[c];
        uint32_t pa = atomicAdd(&whdr[HX(cr.w0 ^ 1)].size, 1u);
        pool[whdr[HX(cr.w0 ^ 1)].base + pa] = make_int4((int)c, cr.w1, (int)cr.start, (int)cr.size);
        uint32_t pb = atomicAdd(&whdr[HX(cr.w1 ^ 1)].size, 1u);
        pool[whdr[HX(cr.w1 ^ 1)].base + pb] = make_int4((int)c, cr.w0, (int)cr.start, (int)cr.size);
    }
    wave_fence();
}

struct LoopState;
// lc_lbd word of a learnt clause: LBD | MS_LBD_NOLOG (another worker may hold a copy: exchanged or imported - its deletion
// is not logged in the proof) | used << 31
#define MS_LBD_MASK 0x1fffffffu
#define MS_LBD_NOLOG 0x40000000u
#define MS_LBD_VIVIFIED 0x20000000u   // the clause went through vivify_pass once
DEV void proof_log_deletions(Wk& w, const MsLayout& L, LoopState* pls, bool dl, uint32_t o0, uint32_t len);

// ---- learnt clause database reduction ---------------------------------------
// Keep every clause with lbd <= 2, every locked clause and every clause used
// since the last reduction with lbd <= 6; of the rest drop the worse half by an
// LBD cut-off (histogram in LDS, no sort), breaking ties by age.
template <bool LV>
DEV void reduce_db(Wk& w, const MsShared& sh, const MsLayout& L, LoopState* pls = nullptr) {
    LdsU32 hist = w.hist;
    Gp<MsClauseRec> lrec = WKA(MsClauseRec, wl) + sh.n_orig;   // records of the learnt clauses
    Gp<uint32_t> lc_lbd = WK_PTR(uint32_t, w, L, lc_lbd);
    Gp<int32_t> lc_lits = WK_PTR(int32_t, w, L, lc_lits);
    Gp<uint32_t> remap = WK_PTR(uint32_t, w, L, remap);
    const uint32_t n = w.n_learnts;
    hist[w.lane] = 0;
    lds_fence();
    for (uint32_t k = (uint32_t)w.lane; k < n; k += MS_WAVE) {
        uint32_t l = lc_lbd[k] & MS_LBD_MASK;
        lds_add(&hist[l > 63 ? 63 : l], 1u);
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
        uint32_t lb = 0, o0 = 0, o1 = 0, raw = 0;
        int2 ww = make_int2(0, 0);
        bool at_cut = false;
        if (act) {
            raw = lc_lbd[k];
            lb = raw & MS_LBD_MASK;
            bool used = raw >> 31;
            const MsClauseRec ch = lrec[k];
            o0 = ch.start;
            o1 = ch.start + ch.size;
            ww = make_int2(ch.w0, ch.w1);
            int cref = (int)(sh.n_orig + k);
            bool locked = (lit_value<LV>(w, sh, L, ww.x) == MS_VAL_TRUE && VREC[ww.x >> 1].reason == cref) ||
                          (lit_value<LV>(w, sh, L, ww.y) == MS_VAL_TRUE && VREC[ww.y >> 1].reason == cref);
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
        // DRUP: deletion lines for the dropped clauses nobody else can hold (before this chunk's literals move)
        if (pls) proof_log_deletions(w, L, pls, act && del && !(raw & MS_LBD_NOLOG), o0, o1 - o0);
        bool keep = act && !del;
        u64 km = ballot(keep);
        uint32_t nkeep = (uint32_t)popc64(km);
        // exclusive prefix of (16-byte aligned) literal counts among kept clauses
        uint32_t len = keep ? (o1 - o0) : 0, pre = (len + 3u) & ~3u;
        for (int o = 1; o < MS_WAVE; o <<= 1) {
            uint32_t t = (uint32_t)__shfl_up((int)pre, o, 64);
            if (w.lane >= o) pre += t;
        }
        uint32_t total = (uint32_t)bcast((int)pre, 63);
        pre -= (len + 3u) & ~3u;
        uint32_t nkk = nk + (uint32_t)popc64(km & lanemask_lt(w.lane));
        if (act) remap[k] = keep ? nkk : 0xffffffffu;
        wave_fence();
        // move literals (dest <= source, clause by clause inside the chunk in lane order)
        for (int src = 0; src < MS_WAVE; src++) {
            if (!((km >> src) & 1)) continue;
            uint32_t so = (uint32_t)bcast((int)o0, src), slen = (uint32_t)bcast((int)len, src);
            uint32_t dd = nlits + (uint32_t)bcast((int)pre, src);
            if (dd != so)
                for (uint32_t t = 0; t < slen; t += MS_WAVE) {
                    uint32_t x = t + (uint32_t)w.lane;
                    int lv = x < slen ? lc_lits[so + x] : 0;
                    wave_fence();
                    if (x < slen) lc_lits[dd + x] = lv;
                    wave_fence();
                }
        }
        if (keep) {
            lrec[nkk] = MsClauseRec{ww.x, ww.y, nlits + pre, len};
            lc_lbd[nkk] = lb | (raw & (MS_LBD_NOLOG | MS_LBD_VIVIFIED));  // clears the used bit
        }
        nk += nkeep;
        nlits += total;
        wave_fence();
    }
    // pass 2: lay the watch lists out again without the deleted clauses (also collects pool garbage)
    w.n_learnts = nk;
    w.lc_lits_n = nlits;
    wave_fence();
    rebuild_watches(w, sh, L);
    // pass 3: reasons of assigned variables
    for (int i = w.lane; i < w.trail_n; i += MS_WAVE) {
        int v = WKA(int32_t, trail)[i] >> 1;
        int r = VREC[v].reason;
        if (r >= 0 && (uint32_t)r >= sh.n_orig) {
            const uint32_t nk2 = remap[(uint32_t)r - sh.n_orig];
            VREC[v].reason = (int)(sh.n_orig + nk2);
            VREC[v].start = lrec[nk2].start;     // (the reduction moved the literals of kept clauses)
        }
    }
    wave_fence();
}

// Store the clause in learnt_buf[0..n) and attach it.  Returns its cref (or -1).
template <bool LV>
DEV int add_learnt(Wk& w, const MsShared& sh, const MsLayout& L, int n, uint32_t lbd, LoopState* pls = nullptr) {
    if (w.n_learnts >= L.learnt_cap || w.lc_lits_n + (uint32_t)n + 8 > L.learnt_lit_cap) {
        reduce_db<LV>(w, sh, L, pls);  // store full before the scheduled reduction: reduce now (state is consistent here)
        if (w.status != MS_ST_RUNNING) return -1;
        if (w.n_learnts >= L.learnt_cap || w.lc_lits_n + (uint32_t)n + 8 > L.learnt_lit_cap) {
            w.status = MS_ST_ERR_LEARNT;
            return -1;
        }
    }
    Gp<const int32_t> learnt_buf = WK_PTR(int32_t, w, L, learnt_buf);
    Gp<int32_t> lc_lits = WK_PTR(int32_t, w, L, lc_lits);
    const uint32_t k = w.n_learnts, o = w.lc_lits_n;
    for (int i = w.lane; i < n; i += MS_WAVE) lc_lits[o + i] = learnt_buf[i];
    const int l0 = uni(learnt_buf[0]), l1 = uni(learnt_buf[1]);
    const int cref = (int)(sh.n_orig + k);
    if (w.lane == 0) {
        WKA(MsClauseRec, wl)[cref] = MsClauseRec{l0, l1, o, (uint32_t)n};
        WK_PTR(uint32_t, w, L, lc_lbd)[k] = lbd;
    }
    w.n_learnts++;
    w.lc_lits_n += ((uint32_t)n + 3u) & ~3u;   // next clause starts 16-byte aligned
    wave_fence();
    if (!list_push_uniform(w, sh, L, l0 ^ 1, cref, l1, o, (uint32_t)n)) return -1;
    if (!list_push_uniform(w, sh, L, l1 ^ 1, cref, l0, o, (uint32_t)n)) return -1;
    return cref;
}

// Every field of Wk except `lane` and `c_cl_lit` has the same value in all lanes, but after a
// vector load (HBM state, scratch copy of a cold call) the compiler cannot know that and would keep
// ~35 VGPRs for them.  readfirstlane makes them scalar values again.
DEV unsigned long long uni64(unsigned long long v) {
    return ((unsigned long long)(uint32_t)uni((int)(v >> 32)) << 32) | (unsigned long long)(uint32_t)uni((int)v);
}
DEV void wk_uniformize(Wk& w) {
    w.slab = (Gp<char>)uni64((unsigned long long)w.slab);
    w.trail_n = uni(w.trail_n); w.qhead = uni(w.qhead); w.n_levels = uni(w.n_levels); w.ring_lo = uni(w.ring_lo);
    w.vm_end = uni(w.vm_end); w.vm_search = uni(w.vm_search); w.sort_n = uni(w.sort_n);
    w.n_learnts = (uint32_t)uni((int)w.n_learnts); w.lc_lits_n = (uint32_t)uni((int)w.lc_lits_n);
    w.pool_top = (uint32_t)uni((int)w.pool_top);
    w.status = uni(w.status); w.lvl_stamp_ctr = (uint32_t)uni((int)w.lvl_stamp_ctr); w.max_groups = uni(w.max_groups);
    w.confl_kind = uni(w.confl_kind); w.confl_cref = uni(w.confl_cref); w.confl_a = uni(w.confl_a);
    w.confl_b = uni(w.confl_b); w.confl_c = uni(w.confl_c);
    w.c_props = (uint32_t)uni((int)w.c_props); w.c_watch = (uint32_t)uni((int)w.c_watch);
    w.c_move = (uint32_t)uni((int)w.c_move); w.c_enq = (uint32_t)uni((int)w.c_enq); w.c_dec = (uint32_t)uni((int)w.c_dec);
    w.c_steps = (uint32_t)uni((int)w.c_steps); w.c_redo = (uint32_t)uni((int)w.c_redo);
}

// ---- worker load / store -------------------------------------------------------
// 16 assignment bytes of the slab <-> one LDS word of sixteen 2-bit fields
DEV uint32_t asg_pack4(uint32_t x) {   // bytes b0..b3 (each 0..3) -> b0 | b1 << 2 | b2 << 4 | b3 << 6
    x = (x | (x >> 6)) & 0x000f000fu;
    return (x | (x >> 12)) & 0xffu;
}
DEV uint32_t asg_unpack4(uint32_t x) {  // the inverse, on the low 8 bits
    x = (x | (x << 12)) & 0x000f000fu;
    return (x | (x << 6)) & 0x03030303u;
}
DEV uint32_t asg_pack(uint4 q) { return asg_pack4(q.x) | asg_pack4(q.y) << 8 | asg_pack4(q.z) << 16 | asg_pack4(q.w) << 24; }
DEV uint4 asg_unpack(uint32_t x) {
    uint4 q;
    q.x = asg_unpack4(x & 0xffu); q.y = asg_unpack4((x >> 8) & 0xffu); q.z = asg_unpack4((x >> 16) & 0xffu); q.w = asg_unpack4(x >> 24);
    return q;
}
template <bool LV>
DEV void wk_bind(Wk& w, const MsShared& sh, const MsLayout& L, char* slab, const MsParams& prm) {
    w.slab = (Gp<char>)slab;
    Gp<const MsState> s = WKA(MsState, state);
    w.trail_n = s->trail_n; w.qhead = s->qhead; w.n_levels = s->n_levels;
    w.vm_end = s->vm_end; w.vm_search = s->vm_search;
    w.n_learnts = s->n_learnts; w.lc_lits_n = s->lc_lits_n; w.pool_top = s->pool_top;
    w.status = s->status;
    w.lvl_stamp_ctr = s->lvl_stamp_ctr;
    w.max_groups = prm.max_groups < 1 ? 1 : (prm.max_groups > MS_MAX_GROUPS ? MS_MAX_GROUPS : prm.max_groups);
    w.ring_lo = w.trail_n;  // nothing staged yet: the queue suffix is re-read from HBM
    w.confl_kind = 0; w.confl_cref = 0; w.confl_a = 0; w.confl_b = 0; w.confl_c = 0;
    w.c_props = w.c_watch = w.c_move = w.c_enq = w.c_dec = w.c_steps = w.c_redo = 0; w.c_cl_lit = 0;
#ifdef MS_PROFILE
    for (int i = 0; i < PF_ALL; i++) w.prof[i] = 0;
#endif
    if (LV) {  // stage the assignment in LDS for this slice: 16 bytes of the slab -> one word of 2-bit fields
        Gp<const uint4> gv = (Gp<const uint4>)WKA(uint8_t, val);
        const uint32_t words = (sh.n_vars + 15) >> 4;
        for (uint32_t i = (uint32_t)w.lane; i < words; i += MS_WAVE) w.lval[i] = asg_pack(gv[i]);
    }
}

template <bool LV>
DEV void wk_store(Wk& w, const MsShared& sh, const MsLayout& L, u64 cycles) {
    u64 cl = wave_sum_u32(w.c_cl_lit);
    lds_fence();
    if (LV) {
        Gp<uint4> gv = (Gp<uint4>)WKA(uint8_t, val);
        const uint32_t words = (sh.n_vars + 15) >> 4;
        for (uint32_t i = (uint32_t)w.lane; i < words; i += MS_WAVE) gv[i] = asg_unpack(w.lval[i]);
    }
    if (w.lane == 0) {
        Gp<MsState> s = WKA(MsState, state);
        s->trail_n = w.trail_n; s->qhead = w.qhead; s->n_levels = w.n_levels;
        s->vm_end = w.vm_end; s->vm_search = w.vm_search;
        s->n_learnts = w.n_learnts; s->lc_lits_n = w.lc_lits_n; s->pool_top = w.pool_top;
        s->status = w.status;
        s->lvl_stamp_ctr = w.lvl_stamp_ctr;
        s->propagations += w.c_props; s->decisions += w.c_dec;
        s->n_watch += w.c_watch; s->n_move += w.c_move; s->n_enq += w.c_enq; s->n_cl_lit += cl;
        s->slice_cycles += cycles;
        s->n_steps += w.c_steps; s->n_redo += w.c_redo;
#ifdef MS_PROFILE
        for (int i = 0; i < PF_ALL; i++) s->prof[i] += w.prof[i];
#endif
    }
}

// ---- the search kernel -------------------------------------------------------------
// Restart / reduction bookkeeping of one worker during a slice.  It is only touched by the two
// cold-path functions below, so it can live in scratch memory.
struct LoopState {
    u64 conflicts, restarts, reduce_dbs, lbdq_sum, lbd_total, next_reduce, learnt_total, learnt_lits_total;
    uint32_t lbdq_n, lbdq_i;
    double trail_avg;
    int n_assumps;
    LdsU32 lbdq;               // LDS ring of the last MS_LBDQ learnt-clause LBDs
    Gp<int32_t> proof_buf;     // DRUP log of this worker, or nullptr
    Gp<uint32_t> proof_len;
    uint32_t proof_cap;
    // clause exchange (share_pool == nullptr: off)
    Gp<const int4> share_pool;
    u64 share_n, share_pos, n_exported, n_imported, n_imported_units, last_import_confl;
    uint32_t share_slots, share_max_lbd, share_max_len, share_interval, exp_n, wid;
    // best-phase rephasing (CaDiCaL's "rephase to best"): the polarities of the longest conflict-free assignment seen
    // become the saved phases every so often - a strong heuristic on satisfiable bounds near the optimum
    int best_trail;
    uint32_t n_rephase;
    u64 next_rephase;
    bool rephase;
    uint32_t import_pct;
    uint32_t vivify;           // clauses per vivification pass (0 = off)
    u64 next_vivify, n_vivified, n_viv_lits;
    double restart_k;          // Glucose's K: restart when the recent LBD average times K exceeds the global one
};

// DRUP deletion lines (-2, literals, -1) for the lanes' dropped clauses, appended to the worker's log.
DEV void proof_log_deletions(Wk& w, const MsLayout& L, LoopState* pls, bool dl, uint32_t o0, uint32_t len) {
    if (!pls->proof_buf || ballot(dl) == 0) return;
    const uint32_t need = dl ? len + 2 : 0;
    uint32_t incl = need;
    for (int o = 1; o < MS_WAVE; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if (w.lane >= o) incl += t;
    }
    const uint32_t total = (uint32_t)bcast((int)incl, 63), base = (uint32_t)uni((int)*pls->proof_len);
    if (base + total > pls->proof_cap) return;      // no room: deletion lines are optional (a lost LEMMA fails the solve)
    if (dl) {
        Gp<int32_t> out = pls->proof_buf + base + (incl - need);
        Gp<const int32_t> lits = WK_PTR(int32_t, w, L, lc_lits) + o0;
        out[0] = -2;
        for (uint32_t j = 0; j < len; j++) out[1 + j] = lits[j];
        out[1 + len] = -1;
    }
    wave_fence();
    if (w.lane == 0) *pls->proof_len = base + total;
    wave_fence();
}

// Every field of LoopState is wave-uniform; after a load from the caller's scratch copy the compiler cannot know that.
DEV void ls_uniformize(LoopState& ls) {
    ls.conflicts = uni64(ls.conflicts); ls.restarts = uni64(ls.restarts); ls.reduce_dbs = uni64(ls.reduce_dbs);
    ls.lbdq_sum = uni64(ls.lbdq_sum); ls.lbd_total = uni64(ls.lbd_total); ls.next_reduce = uni64(ls.next_reduce);
    ls.learnt_total = uni64(ls.learnt_total); ls.learnt_lits_total = uni64(ls.learnt_lits_total);
    ls.lbdq_n = (uint32_t)uni((int)ls.lbdq_n); ls.lbdq_i = (uint32_t)uni((int)ls.lbdq_i);
    ls.trail_avg = __longlong_as_double((long long)uni64((u64)__double_as_longlong(ls.trail_avg)));
    ls.n_assumps = uni(ls.n_assumps);
    ls.lbdq = (LdsU32)uni64((u64)ls.lbdq);
    ls.proof_buf = (Gp<int32_t>)uni64((u64)ls.proof_buf); ls.proof_len = (Gp<uint32_t>)uni64((u64)ls.proof_len);
    ls.proof_cap = (uint32_t)uni((int)ls.proof_cap);
    ls.share_pool = (Gp<const int4>)uni64((u64)ls.share_pool);
    ls.share_n = uni64(ls.share_n); ls.share_pos = uni64(ls.share_pos); ls.n_exported = uni64(ls.n_exported);
    ls.n_imported = uni64(ls.n_imported); ls.n_imported_units = uni64(ls.n_imported_units);
    ls.last_import_confl = uni64(ls.last_import_confl);
    ls.share_slots = (uint32_t)uni((int)ls.share_slots); ls.share_max_lbd = (uint32_t)uni((int)ls.share_max_lbd);
    ls.share_max_len = (uint32_t)uni((int)ls.share_max_len); ls.share_interval = (uint32_t)uni((int)ls.share_interval);
    ls.exp_n = (uint32_t)uni((int)ls.exp_n); ls.wid = (uint32_t)uni((int)ls.wid);
    ls.best_trail = uni(ls.best_trail); ls.n_rephase = (uint32_t)uni((int)ls.n_rephase); ls.next_rephase = uni64(ls.next_rephase);
    ls.rephase = uni((int)ls.rephase) != 0;
    ls.import_pct = (uint32_t)uni((int)ls.import_pct);
    ls.vivify = (uint32_t)uni((int)ls.vivify); ls.next_vivify = uni64(ls.next_vivify); ls.n_vivified = uni64(ls.n_vivified); ls.n_viv_lits = uni64(ls.n_viv_lits);
    ls.restart_k = __longlong_as_double((long long)uni64((u64)__double_as_longlong(ls.restart_k)));
}

// Attach the records of the global ring this worker has not seen yet.  Called at decision level 0
// with the trail at its fixpoint.  Every record is a consequence of the formula alone (learnt clauses
// never depend on assumptions: those are decisions), so it may be added to any worker of any instance
// of the sweep.  A record that is a unit under the level-0 assignment is enqueued; the caller must run
// BCP before its next decision when the queue is not empty afterwards.
template <bool LV>
DEV void import_shared(Wk& w, const MsShared& sh, const MsLayout& L, LoopState& ls) {
    u64 pos = ls.share_pos;
    const u64 end = ls.share_n;
    if (end - pos > ls.share_slots) pos = end - ls.share_slots;   // the ring overwrote what we never read
    Gp<int32_t> learnt_buf = WK_PTR(int32_t, w, L, learnt_buf);
    Gp<const int32_t> pool = (Gp<const int32_t>)ls.share_pool;
    int budget = 4096;   // records per call; the rest waits for the next restart
    for (; pos < end && budget > 0 && w.status == MS_ST_RUNNING; pos++, budget--) {
        // one record per step, one literal per lane (records are 128 bytes: one coalesced load)
        const int word = pool[(pos % ls.share_slots) * MS_SHARE_REC + (u64)(w.lane & (MS_SHARE_REC - 1))];
        const int hdr = bcast(word, 0);
        const int rn = hdr & 63, rl = (hdr >> 6) & 255;
        if ((uint32_t)(hdr >> 14) == ls.wid || rn < 1 || rn > MS_SHARE_MAXLEN) continue;
        // import_pct < 100: a worker attaches only that share of the clauses of 3 and more literals (each worker another one)
        if (rn > 2 && ls.import_pct < 100 && (uint32_t)((((uint32_t)pos * 2654435761u) ^ (ls.wid * 40503u)) >> 13) % 100u >= ls.import_pct) continue;
        const bool in = w.lane >= 1 && w.lane <= rn;
        const int v = in ? lit_value<LV>(w, sh, L, word) : MS_VAL_FALSE;   // units attached before are visible
        if (ballot(in && v == MS_VAL_TRUE)) continue;
        const u64 free_m = ballot(in && v == MS_VAL_UNDEF);
        const int cnt = popc64(free_m);
        if (cnt == 0) { w.status = MS_ST_UNSAT; break; }        // falsified at level 0
        if (cnt == 1) {
            enqueue_uniform<LV>(w, sh, L, bcast(word, first_lane(free_m)), MS_REASON_NONE);
            ls.n_imported_units++;
        } else {
            if (w.n_learnts > L.learnt_cap / 2 && cnt > 2) continue;   // store half full: only binaries
            if ((free_m >> w.lane) & 1) learnt_buf[popc64(free_m & lanemask_lt(w.lane))] = word;
            wave_fence();
            // glue <= 2 would pin it for ever; an imported clause has to earn that here
            if (add_learnt<LV>(w, sh, L, cnt, (uint32_t)(rl < 3 ? 3 : rl) | MS_LBD_NOLOG, &ls) < 0) break;
        }
        ls.n_imported++;
    }
    ls.share_pos = pos;
}

// A conflict was found by propagate(): learn, backjump, assert (Glucose `search` conflict branch).
// Returns true when a restart or a learnt-clause reduction is due at the next fixpoint.
template <bool LV>
DEV bool on_conflict_body(Wk& w, const MsShared& sh, const MsLayout& L, LoopState& ls) {
    PROF_DECL
    ls.conflicts++;
    if (w.n_levels == 0) { w.status = MS_ST_UNSAT; return false; }
    // Glucose restart blocking: a trail much longer than its running average
    // (an exponential average stands in for the 5000-entry queue)
    ls.trail_avg += ((double)w.trail_n - ls.trail_avg) * (1.0 / 5000.0);
    if (ls.conflicts > 10000 && ls.lbdq_n == MS_LBDQ && (double)w.trail_n > 1.4 * ls.trail_avg) {
        ls.lbdq_n = 0; ls.lbdq_i = 0; ls.lbdq_sum = 0;
    }
    if (ls.rephase) {   // the assignment before the last decision was a conflict-free fixpoint: remember the longest
        const int lim = uni(WK_PTR(int32_t, w, L, trail_lim)[w.n_levels - 1]);
        if (lim > ls.best_trail) {
            ls.best_trail = lim;
            Gp<uint8_t> best = WK_PTR(uint8_t, w, L, best);
            for (int i = w.lane; i < lim; i += MS_WAVE) {
                const int l = WKA(int32_t, trail)[i];
                best[l >> 1] = (uint8_t)(l & 1);
            }
        }
    }
    Learnt lr = analyze<LV>(w, sh, L, (uint32_t)(ls.conflicts & 0x3fffu));
    PROF_MARK(PF_ANALYZE);
    if (w.status != MS_ST_RUNNING) return false;
    Gp<const int32_t> learnt_buf = WK_PTR(int32_t, w, L, learnt_buf);
    if (ls.proof_buf) {   // DRUP: every learnt clause, in derivation order
        const uint32_t o = *ls.proof_len;
        if (o + (uint32_t)lr.n + 1 <= ls.proof_cap) {
            for (int i = w.lane; i < lr.n; i += MS_WAVE) ls.proof_buf[o + i] = learnt_buf[i];
            if (w.lane == 0) ls.proof_buf[o + lr.n] = -1;
        }
        wave_fence();
        if (w.lane == 0) *ls.proof_len = o + (uint32_t)lr.n + 1;
        wave_fence();
    }
#ifndef MS_SHARE_SMALL
#define MS_SHARE_SMALL 2     // clauses up to this size are exchanged whatever their LBD
#endif
    bool exported = false;
    if (ls.share_pool && lr.n <= (int)ls.share_max_len && (lr.n <= MS_SHARE_SMALL || lr.lbd <= ls.share_max_lbd) && ls.exp_n < MS_EXPORT_RECS) {
        exported = true;
        Gp<int32_t> rec = WK_PTR(int32_t, w, L, exp) + ls.exp_n * MS_SHARE_REC;
        if (w.lane <= lr.n)
            rec[w.lane] = w.lane == 0 ? (int)((uint32_t)lr.n | ((lr.lbd > 255u ? 255u : lr.lbd) << 6) | (ls.wid << 14))
                                      : learnt_buf[w.lane - 1];
        ls.exp_n++;
        ls.n_exported++;
    }
    cancel_until<LV>(w, sh, L, lr.bt_level);
    if (lr.n == 1) {
        int l0 = uni(learnt_buf[0]);  // unit learnt: bt_level is 0
        if (lit_value<LV>(w, sh, L, l0) == MS_VAL_FALSE) { w.status = MS_ST_UNSAT; return false; }
        enqueue_uniform<LV>(w, sh, L, l0, MS_REASON_NONE);
    } else {
        int cref = add_learnt<LV>(w, sh, L, lr.n, lr.lbd | (exported ? MS_LBD_NOLOG : 0u), &ls);
        if (cref < 0) return false;
        const MsClauseHdr lh = clause_hdr_of(w, sh, L, cref);
        enqueue_uniform<LV>(w, sh, L, uni(learnt_buf[0]), cref, (uint32_t)uni((int)lh.start), (uint32_t)uni((int)lh.size));
    }
    PROF_MARK(PF_BACKJUMP);
    ls.learnt_total++;
    ls.learnt_lits_total += (u64)lr.n;
    ls.lbdq_sum += lr.lbd;
    if (ls.lbdq_n == MS_LBDQ) ls.lbdq_sum -= ls.lbdq[ls.lbdq_i]; else ls.lbdq_n++;
    lds_fence();
    if (w.lane == 0) ls.lbdq[ls.lbdq_i] = lr.lbd;
    lds_fence();
    ls.lbdq_i = (ls.lbdq_i + 1) % MS_LBDQ;
    ls.lbd_total += lr.lbd;
    const bool restart_due = ls.lbdq_n == MS_LBDQ &&
                             ((double)ls.lbdq_sum / MS_LBDQ) * ls.restart_k > (double)ls.lbd_total / (double)ls.conflicts;
    const bool reduce_due = ls.conflicts >= ls.next_reduce || w.n_learnts > L.learnt_cap - L.learnt_cap / 8 ||
                            w.lc_lits_n > L.learnt_lit_cap - L.learnt_lit_cap / 8;
    const bool import_due = ls.share_pool && ls.share_n > ls.share_pos && ls.conflicts - ls.last_import_confl >= ls.share_interval;
    return restart_due || reduce_due || import_due;
}
// The call form (builds for more than one wave per SIMD).  The caller's context arrives by reference, i.e. in scratch
// memory; working on it there makes every field access of the analysis a scratch round trip (measured: 40 % fewer
// conflicts/s than the inlined build at the same worker count), so the function takes copies, works on those - they are
// its own registers - and writes them back.  The immutable descriptors come from the kernel argument segment (scalar
// loads) instead of the caller's scratch copies.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const char __attribute__((address_space(4)))* MsKernarg;     // the kernel's argument segment (constant address space)
#define MS_KERNARG() ((MsKernarg)__builtin_amdgcn_kernarg_segment_ptr())   // only valid inside a __global__ function
#define MS_ARGS_FROM_KERNARG(sh, L, ka, shr, Lr)                                  \
    const MsShared& sh = *(const MsShared*)(ka);                                   \
    const MsLayout& L = *(const MsLayout*)((ka) + ((sizeof(MsShared) + 7) & ~7ul)); \
    (void)shr; (void)Lr;
#else
typedef const char* MsKernarg;
#define MS_KERNARG() ((MsKernarg) nullptr)
#define MS_ARGS_FROM_KERNARG(sh, L, ka, shr, Lr) const MsShared& sh = shr; const MsLayout& L = Lr; (void)ka;
#endif
template <bool LV, bool COPY>
DEV_COLD bool on_conflict(Wk& wr, const MsShared& shr, const MsLayout& Lr, LoopState& lsr, MsKernarg ka) {
    // (4 waves per SIMD: 128 registers do not hold the copies - measured 2.9 vs 3.9e9 prop/s on the 64x64 sweep with them)
    if (!COPY) return on_conflict_body<LV>(wr, shr, Lr, lsr);
    MS_ARGS_FROM_KERNARG(sh, L, ka, shr, Lr)
    Wk w = wr;
    wk_uniformize(w);
    LoopState ls = lsr;
    ls_uniformize(ls);
    const bool r = on_conflict_body<LV>(w, sh, L, ls);
    wr = w;
    lsr = ls;
    return r;
}

// ---- vivification of learnt clauses (Luo et al. 2017, "learnt clause minimisation"; CaDiCaL's vivify) ------------------
// At decision level 0, for a recent learnt clause of small LBD that has not been through this yet: falsify its literals
// one after the other (a decision + BCP each).  A literal that is already false by then is redundant; if one becomes
// true, or BCP runs into a conflict, the literals decided so far (plus that one) already form an implied clause.  The
// shorter clause replaces the old one in place; it is a RUP lemma (logged) and is exported like a freshly learnt clause.
// One literal per lane, so clauses of up to 64 literals.  Returns true if a clause changed.
template <bool LV>
DEV bool vivify_pass(Wk& w, const MsShared& sh, const MsLayout& L, LoopState& ls) {
    Gp<MsClauseRec> lrec = WKA(MsClauseRec, wl) + sh.n_orig;
    Gp<uint32_t> lc_lbd = WK_PTR(uint32_t, w, L, lc_lbd);
    Gp<int32_t> lc_lits = WK_PTR(int32_t, w, L, lc_lits);
    bool changed = false;
    uint32_t done = 0;
    for (int k = (int)w.n_learnts - 1; k >= 0 && done < ls.vivify && w.status == MS_ST_RUNNING && w.qhead == w.trail_n; k--) {
        const uint32_t raw = (uint32_t)uni((int)lc_lbd[k]);
        if ((raw & MS_LBD_VIVIFIED) || (raw & MS_LBD_MASK) > 6u) continue;
        const MsClauseRec cr = lrec[k];
        const int size = uni((int)cr.size);
        const uint32_t start = (uint32_t)uni((int)cr.start);
        if (w.lane == 0) lc_lbd[k] = raw | MS_LBD_VIVIFIED;
        if (size < 3 || size > MS_WAVE) continue;
        done++;
        const bool in = w.lane < size;
        const int lit = in ? lc_lits[start + w.lane] : 0;
        const int v0 = in ? lit_value<LV>(w, sh, L, lit) : MS_VAL_FALSE;
        if (ballot(in && v0 == MS_VAL_TRUE)) continue;                  // satisfied at level 0: nothing to gain
        const u64 alive = ballot(in && v0 == MS_VAL_UNDEF);
        if (popc64(alive) < 2) continue;                                // (a unit or empty clause under level 0: BCP's business)
        u64 kept = 0;
        for (u64 m = alive; m != 0; m &= m - 1) {
            const int i = first_lane(m);
            const int li = bcast(lit, i);
            const int vi = lit_value<LV>(w, sh, L, li);
            if (vi == MS_VAL_TRUE) { kept |= 1ull << i; break; }        // implied by the negations so far
            if (vi == MS_VAL_FALSE) continue;                           // falsified by them: redundant
            kept |= 1ull << i;
            if ((m & (m - 1)) == 0) break;                              // the last literal needs no decision
            new_decision_level<LV>(w, sh, L);
            enqueue_uniform<LV>(w, sh, L, li ^ 1, MS_REASON_NONE);
            if (propagate<LV>(w, sh, L)) break;                         // conflict: the literals decided so far are a clause
        }
        cancel_until<LV>(w, sh, L, 0);
        w.confl_kind = 0;
        const int n_new = popc64(kept);
        if (n_new >= size || n_new < 1) continue;
        changed = true;
        ls.n_vivified++;
        ls.n_viv_lits += (u64)(size - n_new);
        const bool mine = (kept >> w.lane) & 1ull;
        const int rank = popc64(kept & lanemask_lt(w.lane));
        if (ls.proof_buf) {   // DRUP: the shorter clause is a lemma
            const uint32_t o = *ls.proof_len;
            if (o + (uint32_t)n_new + 1 <= ls.proof_cap) {
                if (mine) ls.proof_buf[o + rank] = lit;
                if (w.lane == 0) ls.proof_buf[o + n_new] = -1;
            }
            wave_fence();
            if (w.lane == 0) *ls.proof_len = o + (uint32_t)n_new + 1;
            wave_fence();
        }
        if (n_new == 1) {     // a new level-0 fact; the old clause is satisfied by it
            enqueue_uniform<LV>(w, sh, L, bcast(lit, first_lane(kept)), MS_REASON_NONE);
            break;            // BCP first
        }
        wave_fence();
        if (mine) lc_lits[start + rank] = lit;
        const int w0 = bcast(lit, first_lane(kept)), w1 = bcast(lit, first_lane(kept & (kept - 1)));
        uint32_t lbd = raw & MS_LBD_MASK;
        if (lbd > (uint32_t)n_new - 1) lbd = (uint32_t)n_new - 1;
        bool exported = (raw & MS_LBD_NOLOG) != 0;
        if (!exported && ls.share_pool && n_new <= (int)ls.share_max_len && (n_new <= MS_SHARE_SMALL || lbd <= ls.share_max_lbd) && ls.exp_n < MS_EXPORT_RECS) {
            Gp<int32_t> rec = WK_PTR(int32_t, w, L, exp) + ls.exp_n * MS_SHARE_REC;
            if (w.lane == 0) rec[0] = (int)((uint32_t)n_new | ((lbd > 255u ? 255u : lbd) << 6) | (ls.wid << 14));
            if (mine) rec[1 + rank] = lit;
            ls.exp_n++;
            ls.n_exported++;
            exported = true;
        }
        if (w.lane == 0) {
            lrec[k] = MsClauseRec{w0, w1, start, (uint32_t)n_new};
            lc_lbd[k] = lbd | MS_LBD_VIVIFIED | (raw & 0x80000000u) | (exported ? MS_LBD_NOLOG : 0u);
        }
        wave_fence();
    }
    return changed;
}

// BCP reached a fixpoint without conflict: restart? reduce? then assumptions / next decision.
// VIV = with vivification.  It is left out of the full-fleet build: its code (a second copy of the BCP loop) in the
// cold call costs that build's hot loop registers (81 -> 131 spilled VGPRs, -7 % propagations/s on the 64x64 sweep), and
// a worker of 4096 makes too few conflicts for it to matter there.
template <bool LV, bool VIV = true>
DEV void on_fixpoint_body(Wk& w, const MsShared& sh, const MsLayout& L, LoopState& ls, uint32_t reduce_first,
                          uint32_t reduce_inc) {
    PROF_DECL
    if (ls.lbdq_n == MS_LBDQ && ((double)ls.lbdq_sum / MS_LBDQ) * ls.restart_k > (double)ls.lbd_total / (double)ls.conflicts) {
        ls.lbdq_n = 0; ls.lbdq_i = 0; ls.lbdq_sum = 0;
        ls.restarts++;
        cancel_until<LV>(w, sh, L, 0);
        if (ls.rephase && ls.conflicts >= ls.next_rephase) {
            Gp<const uint8_t> best = WK_PTR(uint8_t, w, L, best);
            for (uint32_t v = (uint32_t)w.lane; v < sh.n_vars; v += MS_WAVE) {
                const uint8_t b = best[v];
                if (b != 255) VREC[v].phase = b;
            }
            ls.n_rephase++;
            ls.next_rephase = ls.conflicts + 2000ull * (ls.n_rephase + 1);
            ls.best_trail = 0;      // the next era records its own best
            wave_fence();
        }
    }
    if (VIV && ls.vivify && w.n_levels == 0 && ls.conflicts >= ls.next_vivify) {
        ls.next_vivify = ls.conflicts + 400;
        if (vivify_pass<LV>(w, sh, L, ls)) rebuild_watches(w, sh, L);   // shrunk clauses watch their new first two literals
        if (w.status != MS_ST_RUNNING || w.qhead < w.trail_n) return;   // a new unit: BCP first
    }
    if (ls.conflicts >= ls.next_reduce || w.n_learnts > L.learnt_cap - L.learnt_cap / 8 ||
        w.lc_lits_n > L.learnt_lit_cap - L.learnt_lit_cap / 8) {
        ls.reduce_dbs++;
        ls.next_reduce = ls.conflicts + reduce_first + (u64)reduce_inc * ls.reduce_dbs;
        reduce_db<LV>(w, sh, L, &ls);
    }
    if (w.pool_top > L.pool_cap - L.pool_cap / 4) rebuild_watches(w, sh, L);  // pool running low: collect holes
    PROF_MARK(PF_REDUCE);
    if (w.status != MS_ST_RUNNING) return;
    if (ls.share_pool && ls.share_n > ls.share_pos &&
        (w.n_levels == 0 || ls.conflicts - ls.last_import_confl >= ls.share_interval)) {
        if (w.n_levels > 0) cancel_until<LV>(w, sh, L, 0);   // other workers' clauses are waiting: take them at level 0
        import_shared<LV>(w, sh, L, ls);
        ls.last_import_confl = ls.conflicts;
        if (w.status != MS_ST_RUNNING || w.qhead < w.trail_n) return;   // imported units: BCP first
    }
    Gp<const int32_t> assumps = WK_PTR(int32_t, w, L, assumps);
    int next = -1;
    while (w.n_levels < ls.n_assumps) {
        int a = uni(assumps[w.n_levels]);
        int va = lit_value<LV>(w, sh, L, a);
        if (va == MS_VAL_TRUE) new_decision_level<LV>(w, sh, L);      // dummy level
        else if (va == MS_VAL_FALSE) { w.status = MS_ST_REFUTED; return; }   // the cube is refuted
        else { next = a; break; }
    }
    if (next < 0) {
        int v = pick_branch_var<LV>(w, sh, L);
        if (v < 0) { w.status = MS_ST_SAT; return; }
        w.c_dec++;
        next = uni(2 * v + (int)VREC[v].phase);
    }
    new_decision_level<LV>(w, sh, L);
    enqueue_uniform<LV>(w, sh, L, next, MS_REASON_NONE);
    PROF_MARK(PF_DECIDE);
}
template <bool LV, bool COPY, bool VIV>
DEV_COLD void on_fixpoint(Wk& wr, const MsShared& shr, const MsLayout& Lr, LoopState& lsr, uint32_t reduce_first, uint32_t reduce_inc,
                          MsKernarg ka) {
    if (!COPY) { on_fixpoint_body<LV, VIV>(wr, shr, Lr, lsr, reduce_first, reduce_inc); return; }
    MS_ARGS_FROM_KERNARG(sh, L, ka, shr, Lr)
    Wk w = wr;
    wk_uniformize(w);
    LoopState ls = lsr;
    ls_uniformize(ls);
    on_fixpoint_body<LV, VIV>(w, sh, L, ls, reduce_first, reduce_inc);
    wr = w;
    lsr = ls;
}

// grid = n_workers blocks of 64 threads.  Runs each worker until it has a verdict,
// or has spent its slice (conflicts / propagations), or the host / another worker
// raised a stop flag.  All state is persisted in the slab, so the host simply
// relaunches the kernel to continue.  LV = assignment staged in (dynamic) LDS.
// The loop body is propagate() plus two cold calls working on a scratch copy of the
// worker context, so the register allocation is that of the BCP loop.
// ONE = the build for a fleet of at most one worker per SIMD (<= 1024 workers): the whole register file of the
// SIMD is the worker's (no spills, no scratch copies: the per-conflict and per-fixpoint code is inlined), which is
// what counts when nothing else hides a wave's latencies.
// WPS = waves per SIMD the build is compiled for: 1 (the ONE build above), 2 (<= 2048 workers: 256 registers per wave, no
// spills, the per-conflict code still a call) or MS_SEARCH_WAVES_PER_SIMD (the full fleet).
template <bool LV, int WPS>
__global__ __launch_bounds__(MS_WAVE, WPS) void ms_search_kernel(MsShared sh, MsLayout L, char* slabs, MsParams prm) {
    constexpr bool ONE = WPS == 1;
    __shared__ int32_t s_ring[MS_LDS_RING];
    __shared__ uint32_t s_claim[MS_CLAIM_SLOTS];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[4 * MS_MAX_GROUPS < 64 ? 64 : 4 * MS_MAX_GROUPS];
    __shared__ uint32_t s_tl[MS_WAVE];
    __shared__ int32_t s_jd[2 * MS_MAX_GROUPS];
    __shared__ uint32_t s_lbdq[64];
    __shared__ int32_t s_bfl[MS_MAX_GROUPS];
    __shared__ uint32_t s_ov;
    __shared__ uint32_t s_mcnt;
    __shared__ uint32_t s_sort[WPS <= 2 ? MS_SORT_N : 1];   // (16 waves per CU have no room for it)
    HIP_DYNAMIC_SHARED(uint32_t, s_lval)
    const uint32_t wid = blockIdx.x;
    if (wid >= prm.n_workers) return;
    Wk w;
    w.lane = (int)threadIdx.x;
    w.sortbuf = (LdsU32)s_sort; w.sort_n = WPS <= 2 ? MS_SORT_N : 0;
    w.ring = (LdsI32)s_ring; w.claim = (LdsU32)s_claim; w.ov_cnt = (LdsU32)&s_ov; w.hist = (LdsU32)s_hist; w.lval = (LdsU32)s_lval; w.bfl = (LdsI32)s_bfl; w.tl = (LdsU32)s_tl; w.jd = (LdsI32)s_jd;
    w.lseen = w.lval + ((sh.n_vars + 15) >> 4);   // (LV) three bitmaps behind the assignment words
    w.lcur = w.lseen + ((sh.n_vars + 31) >> 5);
    w.lzero = w.lcur + ((sh.n_vars + 31) >> 5);
    w.lfail = w.lzero + ((sh.n_vars + 31) >> 5);
    w.lq = w.lfail + ((sh.n_vars + 31) >> 5);
    w.mcnt = (LdsU32)&s_mcnt;
    if (w.lane == 0) { s_ov = 0; s_mcnt = 0; }
    if (LV) for (uint32_t i = (uint32_t)w.lane; i < ((sh.n_vars + 31) >> 5); i += MS_WAVE) { w.lseen[i] = 0; w.lfail[i] = 0; w.lq[i] = 0; }
    wk_bind<LV>(w, sh, L, slabs + (size_t)wid * L.slab_bytes, prm);
    wk_uniformize(w);
    if (LV) { level_zero_rebuild(w, sh, L); cur_level_rebuild(w, sh, L); }
    Gp<MsState> st = WKA(MsState, state);
    if (w.lane < MS_LBDQ) s_lbdq[w.lane] = st->lbdq[w.lane];
    lds_fence();
    const u64 t0 = __builtin_readcyclecounter();
    LoopState ls;
    ls.conflicts = st->conflicts; ls.restarts = st->restarts; ls.reduce_dbs = st->reduce_dbs;
    ls.lbdq_sum = st->lbdq_sum; ls.lbd_total = st->lbd_total; ls.next_reduce = st->next_reduce;
    ls.learnt_total = st->learnt_total; ls.learnt_lits_total = st->learnt_lits_total;
    ls.lbdq_n = st->lbdq_n; ls.lbdq_i = st->lbdq_i; ls.trail_avg = st->trail_avg;
    ls.n_assumps = st->n_assumps; ls.lbdq = (LdsU32)s_lbdq;
    // DRUP: every worker logs the clauses it learns into its own buffer; the host drains all of them after each slice
    ls.proof_buf = prm.proof_buf ? (Gp<int32_t>)prm.proof_buf + (size_t)wid * prm.proof_cap : nullptr;
    ls.proof_len = (Gp<uint32_t>)prm.proof_len + wid; ls.proof_cap = prm.proof_cap;
    ls.share_pool = (Gp<const int4>)prm.share_pool;
    ls.share_n = prm.share_pool ? *(Gp<const unsigned long long>)prm.share_n : 0;
    ls.share_pos = st->share_pos; ls.n_exported = st->n_exported; ls.n_imported = st->n_imported;
    ls.n_imported_units = st->n_imported_units; ls.last_import_confl = st->last_import_confl;
    ls.share_slots = prm.share_slots; ls.share_max_lbd = prm.share_max_lbd; ls.share_interval = prm.share_interval;
    ls.share_max_len = prm.share_max_len < MS_SHARE_MAXLEN ? prm.share_max_len : MS_SHARE_MAXLEN;
    ls.exp_n = st->exp_n; ls.wid = wid;
    ls.best_trail = st->best_trail; ls.n_rephase = st->n_rephase; ls.next_rephase = st->next_rephase;
    ls.rephase = prm.rephase == 1 || (prm.rephase == 2 && (wid & 1u));
    ls.import_pct = prm.import_pct > 0 ? (uint32_t)prm.import_pct : 50u;
    ls.vivify = prm.vivify > 0 ? (uint32_t)prm.vivify : 0u;      // (off by default since round 3)
    ls.next_vivify = st->next_vivify; ls.n_vivified = st->n_vivified; ls.n_viv_lits = st->n_viv_lits;
    // restart_k2_pct: every second worker uses this K instead (a portfolio of restart policies)
    ls.restart_k = 0.01 * (double)(((wid & 1u) && prm.restart_k2_pct > 0) ? prm.restart_k2_pct : (prm.restart_k_pct > 0 ? prm.restart_k_pct : 100));
    const int n_assumps_reg = ls.n_assumps;
    MsShared sc = sh;     // private copies for the cold calls (their address is taken)
    MsLayout lc = L;
    const MsKernarg ka = MS_KERNARG();
#if defined(__HIP_DEVICE_COMPILE__)
    if (WPS == 2) {   // the 2-waves build reads sh / L through the argument segment: make sure they are where it looks
        const MsShared& s2 = *(const MsShared*)ka;
        const MsLayout& l2 = *(const MsLayout*)(ka + ((sizeof(MsShared) + 7) & ~7ul));
        if (s2.n_vars != sh.n_vars || s2.cl_lits != sh.cl_lits || l2.slab_bytes != L.slab_bytes || l2.pool != L.pool || l2.n_vars != L.n_vars) {
            if (w.lane == 0) st->status = MS_ST_ERR_INTERNAL;
            return;
        }
    }
#endif
    uint32_t slice_confl = 0;
    const bool entered_running = w.status == MS_ST_RUNNING;
    if (entered_running && st->restart_req) {   // a new cube was assigned: drop the old search path
        cancel_until<LV>(w, sh, L, 0);
        if (w.lane == 0) st->restart_req = 0;
    }
    const u64 tick0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
    bool maintenance_due = true;   // first fixpoint of the slice goes through the full path once
    while (w.status == MS_ST_RUNNING) {
        if (prm.slice_ticks && __builtin_amdgcn_s_memrealtime() - tick0 >= prm.slice_ticks) break;
        if (propagate<LV>(w, sh, L)) {
            if (ONE) maintenance_due = on_conflict_body<LV>(w, sh, L, ls);
            else {
                Wk t = w;
                maintenance_due = on_conflict<LV, WPS == 2>(t, sc, lc, ls, ka);
                w = t;
                wk_uniformize(w);
            }
            slice_confl++;
            if (slice_confl >= prm.slice_conflicts) break;
            if ((slice_confl & 63) == 0) {
                // one answer per wave (another thread / workgroup may write these while the wave reads them)
                if (uni(*(Gp<const volatile int32_t>)prm.stop_flag)) break;
                if (prm.stop_on_any && uni(*(Gp<const volatile int32_t>)prm.any_done)) break;
            }
        } else {
            if (w.status != MS_ST_RUNNING) break;
            if (prm.slice_props && w.c_props >= prm.slice_props) break;
            if (maintenance_due || w.n_levels < n_assumps_reg || w.pool_top > L.pool_cap - L.pool_cap / 4) {
                // restart / reduce / watch GC / assumptions: the full (cold) path
                if (ONE) on_fixpoint_body<LV>(w, sh, L, ls, prm.reduce_first, prm.reduce_inc);
                else {
                    Wk t = w;
                    on_fixpoint<LV, WPS == 2, WPS == 2>(t, sc, lc, ls, prm.reduce_first, prm.reduce_inc, ka);
                    w = t;
                    wk_uniformize(w);
                }
                maintenance_due = false;
            } else {        // common case: just the next decision
                PROF_DECL
                const int v = pick_branch_var<LV>(w, sh, L);
                if (v < 0) { w.status = MS_ST_SAT; break; }
                w.c_dec++;
                const int next = uni(2 * v + (int)VREC[v].phase);
                new_decision_level<LV>(w, sh, L);
                enqueue_uniform<LV>(w, sh, L, next, MS_REASON_NONE);
                PROF_MARK(PF_DECIDE);
            }
        }
    }
    if (entered_running && w.lane == 0 && prm.any_done &&
        (w.status == MS_ST_SAT || w.status == MS_ST_UNSAT || (w.status == MS_ST_REFUTED && prm.done_on_refuted)))
        atomicExch(prm.any_done, 1);
    lds_fence();
    if (w.lane < MS_LBDQ) st->lbdq[w.lane] = s_lbdq[w.lane];
    if (w.lane == 0) {
        // offer the oldest free decisions for cube splitting (host-side work stealing)
        int ns = 0;
        if (w.status == MS_ST_RUNNING) {
            Gp<const int32_t> tl = WK_PTR(int32_t, w, L, trail_lim);
            Gp<const int32_t> tr = WKA(int32_t, trail);
            for (int lv = ls.n_assumps; lv < w.n_levels && ns < MS_SPLIT_MAX; lv++) st->split[ns++] = tr[tl[lv]];
        }
        st->n_split = ns;
        st->conflicts = ls.conflicts; st->restarts = ls.restarts; st->reduce_dbs = ls.reduce_dbs;
        st->lbdq_sum = ls.lbdq_sum; st->lbd_total = ls.lbd_total; st->next_reduce = ls.next_reduce;
        st->lbdq_n = ls.lbdq_n; st->lbdq_i = ls.lbdq_i; st->trail_avg = ls.trail_avg;
        st->learnt_total = ls.learnt_total; st->learnt_lits_total = ls.learnt_lits_total;
        st->share_pos = ls.share_pos; st->n_exported = ls.n_exported; st->n_imported = ls.n_imported;
        st->n_imported_units = ls.n_imported_units; st->last_import_confl = ls.last_import_confl;
        st->exp_n = ls.exp_n;
        st->best_trail = ls.best_trail; st->n_rephase = ls.n_rephase; st->next_rephase = ls.next_rephase;
        st->next_vivify = ls.next_vivify; st->n_vivified = ls.n_vivified; st->n_viv_lits = ls.n_viv_lits;
    }
    wk_store<LV>(w, sh, L, __builtin_readcyclecounter() - t0);
}

// ---- scripted BCP kernel (BASELINE.json configs[1]) -----------------------------------
// Each worker propagates the formula's own units, then takes its scripted
// decisions one decision level at a time.  status: MS_ST_SAT is (ab)used as "fixpoint
// reached without conflict", MS_ST_UNSAT as "conflict".
template <bool LV>
__global__ __launch_bounds__(MS_WAVE) void ms_bcp_kernel(MsShared sh, MsLayout L, char* slabs, MsParams prm) {
    __shared__ int32_t s_ring[MS_LDS_RING];
    __shared__ uint32_t s_claim[MS_CLAIM_SLOTS];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[4 * MS_MAX_GROUPS < 64 ? 64 : 4 * MS_MAX_GROUPS];
    __shared__ uint32_t s_tl[MS_WAVE];
    __shared__ int32_t s_jd[2 * MS_MAX_GROUPS];
    __shared__ int32_t s_bfl[MS_MAX_GROUPS];
    __shared__ uint32_t s_ov;
    __shared__ uint32_t s_mcnt;
    HIP_DYNAMIC_SHARED(uint32_t, s_lval)
    const uint32_t wid = blockIdx.x;
    if (wid >= prm.n_workers) return;
    Wk w;
    w.lane = (int)threadIdx.x;
    w.ring = (LdsI32)s_ring; w.claim = (LdsU32)s_claim; w.ov_cnt = (LdsU32)&s_ov; w.hist = (LdsU32)s_hist; w.lval = (LdsU32)s_lval; w.bfl = (LdsI32)s_bfl; w.tl = (LdsU32)s_tl; w.jd = (LdsI32)s_jd;
    w.lseen = w.lval + ((sh.n_vars + 15) >> 4);   // (no analysis in this kernel; the level bitmaps are still maintained)
    w.lcur = w.lseen + ((sh.n_vars + 31) >> 5);
    w.lzero = w.lcur + ((sh.n_vars + 31) >> 5);
    w.lfail = w.lzero + ((sh.n_vars + 31) >> 5);
    w.lq = w.lfail + ((sh.n_vars + 31) >> 5);
    w.mcnt = (LdsU32)&s_mcnt;
    if (w.lane == 0) { s_ov = 0; s_mcnt = 0; }
    wk_bind<LV>(w, sh, L, slabs + (size_t)wid * L.slab_bytes, prm);
    wk_uniformize(w);
    if (LV) { level_zero_rebuild(w, sh, L); cur_level_rebuild(w, sh, L); }
    lds_fence();
    const u64 t0 = __builtin_readcyclecounter();
    const int n_script = WKA(MsState, state)->n_script;
    Gp<const int32_t> script = WK_PTR(int32_t, w, L, script);
    bool confl = propagate<LV>(w, sh, L);
    for (int d = 0; d < n_script && !confl && w.status == MS_ST_RUNNING; d++) {
        int a = uni(script[d]);
        int va = lit_value<LV>(w, sh, L, a);
        if (va == MS_VAL_TRUE) continue;
        if (va == MS_VAL_FALSE) { confl = true; break; }
        new_decision_level<LV>(w, sh, L);
        enqueue_uniform<LV>(w, sh, L, a, MS_REASON_NONE);
        confl = propagate<LV>(w, sh, L);
    }
    if (w.status == MS_ST_RUNNING) w.status = confl ? MS_ST_UNSAT : MS_ST_SAT;
    wk_store<LV>(w, sh, L, __builtin_readcyclecounter() - t0);
}

// ---- failed-literal probing (formula simplification before search) -----------------------------------
// The reference's backend is `simp::Glucose` (crates/repl/src/main.rs:17): it simplifies the formula before it
// searches.  Probing is the part of that which is unit propagation, i.e. this kernel's hot loop: worker w takes the
// probe literals of its script one after the other - assume the literal at decision level 1, propagate to fixpoint,
// note the outcome, backtrack - and both polarities of a variable go to the same worker back to back, so that it can
// compare what they imply:
//   p -> conflict                      : ~p is a consequence of the formula (failed literal)
//   p -> m  and  ~p -> m               : m is a consequence (necessary assignment)
//   p -> m  and  ~p -> ~m              : m == p (equivalent literals)
// Results: script[d] is overwritten with the number of literals the probe implied, or -1 for a failed literal, or -2
// if it was not probed (already assigned); facts go to `toclear` as (kind, a, b) triples: kind 1 = unit a,
// kind 2 = a == b; their number to learnt_buf[0].
template <bool LV>
__global__ __launch_bounds__(MS_WAVE) void ms_probe_kernel(MsShared sh, MsLayout L, char* slabs, MsParams prm) {
    __shared__ int32_t s_ring[MS_LDS_RING];
    __shared__ uint32_t s_claim[MS_CLAIM_SLOTS];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[4 * MS_MAX_GROUPS < 64 ? 64 : 4 * MS_MAX_GROUPS];
    __shared__ uint32_t s_tl[MS_WAVE];
    __shared__ int32_t s_jd[2 * MS_MAX_GROUPS];
    __shared__ int32_t s_bfl[MS_MAX_GROUPS];
    __shared__ uint32_t s_ov;
    __shared__ uint32_t s_mcnt;
    HIP_DYNAMIC_SHARED(uint32_t, s_lval)
    const uint32_t wid = blockIdx.x;
    if (wid >= prm.n_workers) return;
    Wk w;
    w.lane = (int)threadIdx.x;
    w.ring = (LdsI32)s_ring; w.claim = (LdsU32)s_claim; w.ov_cnt = (LdsU32)&s_ov; w.hist = (LdsU32)s_hist; w.lval = (LdsU32)s_lval; w.bfl = (LdsI32)s_bfl; w.tl = (LdsU32)s_tl; w.jd = (LdsI32)s_jd;
    w.lseen = w.lval + ((sh.n_vars + 15) >> 4);
    w.lcur = w.lseen + ((sh.n_vars + 31) >> 5);
    w.lzero = w.lcur + ((sh.n_vars + 31) >> 5);
    w.lfail = w.lzero + ((sh.n_vars + 31) >> 5);
    w.lq = w.lfail + ((sh.n_vars + 31) >> 5);
    w.mcnt = (LdsU32)&s_mcnt;
    if (w.lane == 0) { s_ov = 0; s_mcnt = 0; }
    wk_bind<LV>(w, sh, L, slabs + (size_t)wid * L.slab_bytes, prm);
    wk_uniformize(w);
    if (LV) { level_zero_rebuild(w, sh, L); cur_level_rebuild(w, sh, L); }
    lds_fence();
    const u64 t0 = __builtin_readcyclecounter();
    const int n_script = WKA(MsState, state)->n_script;
    Gp<int32_t> script = WK_PTR(int32_t, w, L, script);
    Gp<uint32_t> stamp = WK_PTR(uint32_t, w, L, lvl_stamp);     // per variable: (probe index + 1) << 1 | sign of the implied literal
    Gp<int32_t> facts = WK_PTR(int32_t, w, L, toclear);
    const int fact_cap = ((int)sh.n_vars + 1) / 3;
    int n_facts = 0;
    bool confl = propagate<LV>(w, sh, L);      // the formula's own units (normally already at their fixpoint)
    int prev_a = -1;                           // literal of the previous probe if it ran to a fixpoint, else -1
    for (int d = 0; d < n_script && !confl && w.status == MS_ST_RUNNING; d++) {
        const int a = uni(script[d]);
        int res = -2;
        bool ran = false;
        if (lit_value<LV>(w, sh, L, a) == MS_VAL_UNDEF) {
            const int base = w.trail_n;
            new_decision_level<LV>(w, sh, L);
            enqueue_uniform<LV>(w, sh, L, a, MS_REASON_NONE);
            const bool failed = propagate<LV>(w, sh, L);
            res = failed ? -1 : w.trail_n - base - 1;
            // the partner probe (same variable, other polarity) is the previous script entry
            const bool second = prev_a == (a ^ 1);
            ran = !failed;
            if (!failed) {
                Gp<const int32_t> trail = WKA(int32_t, trail);
                const uint32_t mine = (uint32_t)(d + 1) << 1, prev = (uint32_t)d << 1;
                for (int i0 = base + 1; i0 < w.trail_n; i0 += MS_WAVE) {
                    const int i = i0 + w.lane;
                    const bool in = i < w.trail_n;
                    const int m = in ? trail[i] : 0;
                    int kind = 0;
                    if (in) {
                        const uint32_t st = stamp[m >> 1];
                        if (second && (st >> 1) == (prev >> 1)) kind = (st & 1u) == (uint32_t)(m & 1) ? 1 : 2;
                        stamp[m >> 1] = mine | (uint32_t)(m & 1);
                    }
                    const u64 km = ballot(kind != 0);
                    if (kind != 0) {
                        const int o = n_facts + popc64(km & lanemask_lt(w.lane));
                        // kind 2: the previous probe (~a) implied ~m and this one (a) implies m: m == a
                        if (o < fact_cap) { facts[3 * o] = kind; facts[3 * o + 1] = m; facts[3 * o + 2] = a; }
                    }
                    n_facts += popc64(km);
                    wave_fence();
                }
            }
            cancel_until<LV>(w, sh, L, 0);
            w.confl_kind = 0;
        }
        prev_a = ran ? a : -1;
        wave_fence();
        if (w.lane == 0) script[d] = res;
        wave_fence();
    }
    if (w.lane == 0) WK_PTR(int32_t, w, L, learnt_buf)[0] = n_facts < fact_cap ? n_facts : fact_cap;
    if (w.status == MS_ST_RUNNING) w.status = confl ? MS_ST_UNSAT : MS_ST_SAT;
    wk_store<LV>(w, sh, L, __builtin_readcyclecounter() - t0);
}
