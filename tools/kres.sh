#!/bin/bash
# usage: tools_kres.sh file.hip [extra flags]  -> per-kernel VGPR/SGPR/LDS/occupancy table
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaussianavatar_amd/csrc "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re
cur={}
for l in sys.stdin:
    m=re.search(r'remark: [^ ]+ +(.*?)\s*\[-Rpass',l)
    if not m: 
        m=re.search(r':\d+:\d+: remark:\s+(.*?)\s*\[-Rpass',l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        if cur: print(cur)
        cur={'fn':re.sub(r'^_ZN3gsr12_GLOBAL__N_1\d+','',t.split(':',1)[1].strip())[:28]}
    elif ':' in t:
        k,v=t.split(':',1); k=k.strip()
        if k in ('VGPRs','AGPRs','SGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','LDS Size [bytes/block]','VGPR Spill','SGPR Spill'): cur[k.split(' ')[0]]=v.strip()
if cur: print(cur)
"
