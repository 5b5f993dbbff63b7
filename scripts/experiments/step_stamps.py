#!/usr/bin/env python3
"""Fine-grained cycle stamps WITHOUT forced memory waits: patches a COPY of timberborn_support_solver_amd/csrc (mi355sat.hip + device/)
given as argv[1]; build the copy with hipcc -shared and run it through BENCH_LIB (scripts/gpu_rung.py ... verbose=1 prints [xq]).
Diagnostic experiment builds only - never the product library.  DESIGN.md section 4, "The BCP step, measured from inside"."""
import sys, os
d = sys.argv[1]
p=os.path.join(d,'device/layout.h'); s=open(p).read()
s=s.replace("uint64_t prof[16];","uint64_t prof[64];"); open(p,'w').write(s)
p=os.path.join(d,'device/kernels.hip.h'); s=open(p).read()
def rep(a,b,cnt=1):
    global s
    if s.count(a)!=cnt: raise SystemExit(("anchor", s.count(a), a[:80]))
    s=s.replace(a,b)
rep('''#define DEV __device__ __forceinline__''','''#define DEV __device__ __forceinline__
#define XP_MARK(slot) do { u64 n_ = __builtin_readcyclecounter(); if (w.lane == 0) w.xp[slot] += n_ - w.xpt; w.xpt = n_; } while (0)
#define XP_CNT(slot, v) do { if (w.lane == 0) w.xp[slot] += (u64)(v); } while (0)
#define XP_RESET() do { w.xpt = __builtin_readcyclecounter(); } while (0)''')
rep('''    LdsI32 jd;                 // j / done''','''    u64 MS_LDS* xp; u64 xpt;
    LdsI32 jd;                 // j / done''')
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
        XP_CNT(27, 1);
        int lg =''')
rep('''        if (sl == 0) w.bfl[g] = fl;
        lds_fence();''','''        if (sl == 0) w.bfl[g] = fl;
        lds_fence();
        XP_MARK(0);''')
rep('''        PROF_MARK(PF_OFF);
        // evaluate binary''','''        PROF_MARK(PF_OFF);
        XP_MARK(1);
        // evaluate binary''')
rep('''        PROF_MARK(PF_BIN);
        // first chunk''','''        PROF_MARK(PF_BIN);
        XP_MARK(2);
        // first chunk''')
rep('''        PROF_MARK(PF_LONG);
        // ONE commit''','''        PROF_MARK(PF_LONG);
        XP_RESET();
        // ONE commit''')
rep('''        {   // in-place compaction of the first chunk of each group's watch list''','''        XP_MARK(6);
        {   // in-place compaction of the first chunk of each group's watch list''')
rep('''        PROF_MARK(PF_TERN);
        bool lost;''','''        PROF_MARK(PF_TERN);
        XP_MARK(7);
        bool lost;''')
rep('''        // ---- the rest of long ternary lists, likewise ----------------------
        {''','''        XP_MARK(8);
        // ---- the rest of long ternary lists, likewise ----------------------
        {''')
rep('''        // ---- the rest of long watch lists, spread flat over all 64 lanes''','''        XP_MARK(9);
        // ---- the rest of long watch lists, spread flat over all 64 lanes''')
rep('''                const int total = flat_setup(w, G, g, sl, rem, (int)wb, fl);
                if (sl == 0) { w.jd[g] = j;''','''                const int total = flat_setup(w, G, g, sl, rem, (int)wb, fl);
                XP_CNT(22, 1);
                XP_MARK(10);
                if (sl == 0) { w.jd[g] = j;''')
rep('''                    if (act) flat_item<LV>(w, G, item, gg, ix, fwb, ffl, prev, next);
                    const int i = S + ix;''','''                    if (act) flat_item<LV>(w, G, item, gg, ix, fwb, ffl, prev, next);
                    XP_CNT(23, 1);
                    XP_MARK(11);
                    const int i = S + ix;''')
rep('''                    w.c_watch += (uint32_t)popc64(ballot(live));
                    LongRes R = long_eval<LV>(w, sh, L, wt, live, vbl, ww, ffl, gg, h0, h1);''','''                    w.c_watch += (uint32_t)popc64(ballot(live));
                    XP_MARK(12);
                    LongRes R = long_eval<LV>(w, sh, L, wt, live, vbl, ww, ffl, gg, h0, h1);
                    XP_RESET();''')
rep('''                lds_fence();
                j = w.jd[g];
                done = w.jd[MS_MAX_GROUPS + g];''','''                lds_fence();
                XP_MARK(13);
                j = w.jd[g];
                done = w.jd[MS_MAX_GROUPS + g];''')
rep('''        PROF_MARK(PF_CLOSE);
        if (w.confl_kind) { w.qhead = w.trail_n; return true; }''','''        PROF_MARK(PF_CLOSE);
        XP_MARK(14);
        if (w.confl_kind) { XP_CNT(28, 1); w.qhead = w.trail_n; return true; }''')
# long_eval
rep('''    Gp<MsWatchHdr> whdr = WKA(MsWatchHdr, whdr);
    const bool push_a = scanning && r >= 0;''','''    XP_MARK(3);
    Gp<MsWatchHdr> whdr = WKA(MsWatchHdr, whdr);
    const bool push_a = scanning && r >= 0;''')
rep('''    const bool push = scanning && r >= 0;
    const int t = r ^ 1;''','''    XP_MARK(4);
    const bool push = scanning && r >= 0;
    const int t = r ^ 1;''')
rep('''    w.c_cl_lit += nl;
    return R;''','''    XP_MARK(5);
    w.c_cl_lit += nl;
    return R;''')
# analysis / fixpoint totals
rep('''    PROF_MARK(PF_ANALYZE);''','''    PROF_MARK(PF_ANALYZE);
    XP_CNT(29, 1);''')
open(p,'w').write(s)
p=os.path.join(d,'mi355sat.hip'); s=open(p).read()
a='''    uint64_t prof[16] = {0}, cyc = 0;
    for (auto& st : sts) { for (int i = 0; i < 16; i++) prof[i] += st.prof[i]; cyc += st.slice_cycles; }'''
b='''    uint64_t prof[64] = {0}, cyc = 0;
    for (auto& st : sts) { for (int i = 0; i < 64; i++) prof[i] += st.prof[i]; cyc += st.slice_cycles; }
    if (s.opts.verbose) {
        const char* xn[32] = {"hdr","chunks+vals","evalbt","LE_A","LE_B","LE_C","commit","compact","flatb","flatt","rest_setup","rest_item","rest_loads","rest_tail","close","","","","","",
                              "","","n_rest","it_rest","","","","steps","n_confl","n_analyze","",""};
        fprintf(stderr, "[xq] cyc=%.3e", (double)cyc);
        for (int i = 0; i < 32; i++) if (xn[i][0]) fprintf(stderr, " %s=%.4g", xn[i], i < 20 ? 100.0 * (double)prof[16 + i] / (double)cyc : (double)prof[16 + i]);
        fprintf(stderr, "\\n");
    }'''
assert s.count(a)==1
s=s.replace(a,b); open(p,'w').write(s)
print("instrumented", d)
