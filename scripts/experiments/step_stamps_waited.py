#!/usr/bin/env python3
"""Cycle stamps with a forced s_waitcnt before each (which round trip is waited for where) + step counters: patches a COPY of
timberborn_support_solver_amd/csrc given as argv[1] (verbose=1 prints [xp]).  Diagnostic experiment builds only."""
import sys, os
d = sys.argv[1]
p=os.path.join(d,'device/layout.h'); s=open(p).read()
s=s.replace("uint64_t prof[16];","uint64_t prof[64];"); open(p,'w').write(s)
p=os.path.join(d,'device/kernels.hip.h'); s=open(p).read()
def rep(a,b,cnt=1,opt=False):
    global s
    if s.count(a)!=cnt:
        if opt: print("skip", a[:60]); return
        raise SystemExit(("anchor", s.count(a), a[:80]))
    s=s.replace(a,b)
rep('''#define DEV __device__ __forceinline__''','''#define DEV __device__ __forceinline__
#define XP_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define XP_MARK(slot) do { XP_WAIT(); u64 n_ = __builtin_readcyclecounter(); if (w.lane == 0) w.xp[slot] += n_ - w.xpt; w.xpt = n_; } while (0)
#define XP_CNT(slot, v) do { if (w.lane == 0) w.xp[slot] += (u64)(v); } while (0)
#define XP_RESET() do { XP_WAIT(); w.xpt = __builtin_readcyclecounter(); } while (0)''')
rep('''    u64 prof[PF_ALL];          // phase cycles, then counts / sub-phases of conflict analysis
#endif''','''    u64 prof[PF_ALL];          // phase cycles, then counts / sub-phases of conflict analysis
#endif
    u64 MS_LDS* xp; u64 xpt;''')
rep('''    HIP_DYNAMIC_SHARED(uint32_t, s_lval)
    const uint32_t wid = blockIdx.x;
    if (wid >= prm.n_workers) return;
    Wk w;
    w.lane = (int)threadIdx.x;
    w.sortbuf''','''    __shared__ u64 s_xp[48];
    HIP_DYNAMIC_SHARED(uint32_t, s_lval)
    const uint32_t wid = blockIdx.x;
    if (wid >= prm.n_workers) return;
    Wk w;
    w.lane = (int)threadIdx.x;
    w.xp = (u64 MS_LDS*)s_xp; w.xpt = 0;
    if (w.lane < 48) s_xp[w.lane] = 0;
    w.sortbuf''')
lines=s.split('\n')
for i,l in enumerate(lines):
    if l.strip()=="Wk w;" and "s_xp" not in lines[i+2]:
        lines[i]="    __shared__ u64 s_xp_aux[48]; Wk w; w.xp = (u64 MS_LDS*)s_xp_aux; w.xpt = 0; if (threadIdx.x < 48) s_xp_aux[threadIdx.x] = 0;"
s='\n'.join(lines)
rep('''        for (int i = 0; i < PF_ALL; i++) s->prof[i] += w.prof[i];
#endif''','''        for (int i = 0; i < PF_ALL; i++) s->prof[i] += w.prof[i];
#endif
        for (int i = 0; i < 48; i++) s->prof[16 + i] += w.xp[i];''')
rep('''        const int qlen = w.trail_n - w.qhead;
        int lg =''','''        const int qlen = w.trail_n - w.qhead;
        XP_RESET();
        XP_CNT(qlen >= 64 ? 6 : qlen >= 32 ? 5 : qlen >= 16 ? 4 : qlen >= 8 ? 3 : qlen >= 4 ? 2 : qlen >= 2 ? 1 : 0, 1);
        XP_CNT(27, 1);
        int lg =''')
rep('''        const int fl = p ^ 1;
        const uint32_t b0 = wh.bin_off,''','''        const int fl = p ^ 1;
        XP_MARK(7);
        const uint32_t b0 = wh.bin_off,''')
rep('''        w.qhead += G;
        w.c_props += (uint32_t)G;
        w.c_steps++;''','''        XP_MARK(8);
        w.qhead += G;
        w.c_props += (uint32_t)G;
        w.c_steps++;''')
rep('''        PROF_MARK(PF_OFF);
        // evaluate binary''','''        PROF_MARK(PF_OFF);
        XP_MARK(9);
        // evaluate binary''')
rep('''        PROF_MARK(PF_BIN);
        // first chunk''','''        PROF_MARK(PF_BIN);
        XP_MARK(10);
        // first chunk''')
rep('''        PROF_MARK(PF_LONG);
        // ONE commit''','''        PROF_MARK(PF_LONG);
        XP_RESET();
        // ONE commit''')
rep('''        {   // in-place compaction of the first chunk of each group's watch list''','''        XP_MARK(14);
        {   // in-place compaction of the first chunk of each group's watch list''')
rep('''        PROF_MARK(PF_TERN);
        bool lost;''','''        PROF_MARK(PF_TERN);
        XP_MARK(15);
        bool lost;''')
rep('''                const int total = flat_setup(w, G, g, sl, rem, (int)b0, fl);''','''                XP_CNT(24, 1);
                const int total = flat_setup(w, G, g, sl, rem, (int)b0, fl);''')
rep('''        // ---- the rest of long ternary lists, likewise ----------------------
        {''','''        XP_MARK(16);
        // ---- the rest of long ternary lists, likewise ----------------------
        {''')
rep('''                const int total = flat_setup(w, G, g, sl, rem, (int)t0, fl);''','''                XP_CNT(25, 1);
                const int total = flat_setup(w, G, g, sl, rem, (int)t0, fl);''')
rep('''        // ---- the rest of long watch lists, spread flat over all 64 lanes''','''        XP_MARK(17);
        if (ballot(n > S && sl == 0) != 0 && !any_cf) XP_CNT(22, 1);
        // ---- the rest of long watch lists, spread flat over all 64 lanes''')
rep('''                    if (act) flat_item<LV>(w, G, item, gg, ix, fwb, ffl, prev, next);''','''                    XP_CNT(23, 1);
                    if (act) flat_item<LV>(w, G, item, gg, ix, fwb, ffl, prev, next);''')
rep('''        // close each group's list: fully visited -> new size; interrupted -> tombstone the gap
        // between the compacted prefix and the first unvisited entry
        wave_fence();''','''        XP_MARK(18);
        // close each group's list: fully visited -> new size; interrupted -> tombstone the gap
        // between the compacted prefix and the first unvisited entry
        wave_fence();''')
rep('''        PROF_MARK(PF_CLOSE);
        if (w.confl_kind) { w.qhead = w.trail_n; return true; }''','''        PROF_MARK(PF_CLOSE);
        XP_MARK(19);
        if (w.confl_kind) { XP_CNT(28, 1); w.qhead = w.trail_n; return true; }''')
# long_eval: A ends where the tail mask is taken; B ends at phase C
rep('''    const u64 tm = ballot(need_tail);
    if (tm != 0) {''','''    XP_MARK(11);
    const u64 tm = ballot(need_tail);
    XP_CNT(30, popc64(tm));
    if (ballot(need_tail)) XP_CNT(20, 1);
    if (ballot(live && vbl != MS_VAL_TRUE && vo == MS_VAL_FALSE)) XP_CNT(26, 1);
    if (ballot(live && vbl != MS_VAL_TRUE)) XP_CNT(29, 1);
    if (tm != 0) {''', opt=True)
rep('''        for (int k0 = MS_LANE_SCAN; ballot(open) != 0; k0 += SL) {''','''        for (int k0 = MS_LANE_SCAN; ballot(open) != 0; k0 += SL) {
            XP_CNT(31, 1);''', opt=True)
rep('''    const bool push = scanning && r >= 0;
    const int t = r ^ 1;''','''    XP_MARK(12);
    const bool push = scanning && r >= 0;
    if (ballot(push)) XP_CNT(21, 1);
    const int t = r ^ 1;''')
rep('''    w.c_cl_lit += nl;
    return R;''','''    XP_MARK(13);
    w.c_cl_lit += nl;
    return R;''')
# analysis side: total cycles in on_conflict / on_fixpoint are visible as the remainder
open(p,'w').write(s)
p=os.path.join(d,'mi355sat.hip'); s=open(p).read()
a='''    uint64_t prof[16] = {0}, cyc = 0;
    for (auto& st : sts) { for (int i = 0; i < 16; i++) prof[i] += st.prof[i]; cyc += st.slice_cycles; }'''
b='''    uint64_t prof[64] = {0}, cyc = 0;
    for (auto& st : sts) { for (int i = 0; i < 64; i++) prof[i] += st.prof[i]; cyc += st.slice_cycles; }
    if (s.opts.verbose) {
        const char* xn[32] = {"q1","q2-3","q4-7","q8-15","q16-31","q32-63","q64+","RT1","RT2","RT3","evalbt","LE_A","LE_B","LE_C","commit","compact","flatb","flatt","longrest","close",
                              "n_tail","n_push","n_longrest","it_longrest","n_flatb","n_flatt","n_vofalse","steps","n_confl","n_nontrue","tails","tail_rounds"};
        fprintf(stderr, "[xp] cyc=%.3e", (double)cyc);
        for (int i = 0; i < 32; i++) fprintf(stderr, " %s=%.4g", xn[i], (i >= 7 && i < 20) ? 100.0 * (double)prof[16 + i] / (double)cyc : (double)prof[16 + i]);
        fprintf(stderr, "\\n");
    }'''
assert s.count(a)==1
s=s.replace(a,b); open(p,'w').write(s)
print("instrumented", d)
