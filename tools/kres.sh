#!/bin/bash
# usage: tools/kres.sh file.hip [extra hipcc flags]  -> per-kernel VGPR/SGPR/LDS/scratch/occupancy
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaussianavatar_amd/csrc "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re
cur=None
keys={'VGPRs':'vgpr','AGPRs':'agpr','SGPRs':'sgpr','ScratchSize [bytes/lane]':'scratch','Occupancy [waves/SIMD]':'occ','LDS Size [bytes/block]':'lds'}
def flush():
    if cur: print('%-34s'%cur['fn'], ' '.join('%s=%s'%(k,cur.get(k,'?')) for k in keys.values()))
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s*\[-Rpass',l)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        flush(); n=t.split(':',1)[1].strip(); n=re.sub(r'^_ZN\d+\w*?_GLOBAL__N_1','',n); n=re.sub(r'^_Z','',n); cur={'fn':n[:34]}
    elif ':' in t and cur is not None:
        k,v=t.split(':',1)
        if k.strip() in keys: cur[keys[k.strip()]]=v.strip()
flush()
"
