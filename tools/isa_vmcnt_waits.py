"""Where does hipcc wait for global memory (s_waitcnt vmcnt) inside a kernel?  Two patterns cost this repository double-digit
percentages until they were found by reading these lines: a wait in front of LDS reads / stores that follow an LDS-DMA
(global_load_lds) of the NEXT frame, and a wait in front of every predicated global load (loads behind a branch go out one
latency at a time).
usage: hipcc -O3 --offload-arch=gfx950 -Iinclude -Ipyscenedetect_amd/csrc -c <file>.hip -save-temps=obj -o /tmp/x.o
       python tools/isa_vmcnt_waits.py /tmp/<file>-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-name-regex]"""
import re,sys
s=open(sys.argv[1]).read()
pat=sys.argv[2] if len(sys.argv)>2 else '.'
for m in re.finditer(r'^(_ZN3psd\w+):\s*;.*?\n(.*?)\.amdhsa_kernel \1', s, re.S|re.M):
    name,body=m.group(1),m.group(2)
    if not re.search(pat,name): continue
    lines=body.split('\n')
    # find main loop region: between first "Loop Header" and end; list vmcnt waits with 1 line context
    w=[i for i,l in enumerate(lines) if 's_waitcnt vmcnt' in l]
    print(name[:75], 'lines',len(lines),'vmcnt waits at',w[:20], 'v_:',sum(1 for l in lines if l.startswith('\tv_')))
