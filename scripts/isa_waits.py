#!/usr/bin/env python3
"""Which loads does a kernel wait for one at a time?  Reads an ISA listing with line tables (hipcc -O3 -gline-tables-only -S
--cuda-device-only -o build/prod_g.s mi355sat.hip) and lists, per kernel symbol given on the command line, the source lines of loads
that are followed by an s_waitcnt vmcnt(0) with no other load in between.  usage: scripts/isa_waits.py SYMBOL..."""
import re,collections,sys
lines=open('build/prod_g.s').read().split('\n')
def krange(name):
    s=[i for i,l in enumerate(lines) if l.startswith(name+':')][0]
    e=[i for i in range(s,len(lines)) if lines[i].startswith('.Lfunc_end')][0]
    return s,e
files={}
for l in lines:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2))
for kname in sys.argv[1:]:
    s,e=krange(kname)
    cur=None; loads=[]; res=[]
    for i in range(s,e):
        l=lines[i]
        m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',l)
        if m: cur=(files.get(int(m.group(1)),'?').split('/')[-1],int(m.group(2))); continue
        if re.search(r'\b(global_load|scratch_load|buffer_load)',l): loads.append(cur)
        m=re.search(r's_waitcnt.*vmcnt\((\d+)\)',l)
        if m and int(m.group(1))==0:
            res.append((cur,len(loads),list(loads))); loads=[]
    print(kname[:40], 'waits(0):',len(res))
    c=collections.Counter((r[2][0]) for r in res if r[1]==1)
    for k,v in sorted(c.items(), key=lambda x:(x[0] or ('',0))):
        print('   single-load wait: load at',k,'x',v)
