#!/bin/bash
# prints VGPR/AGPR/occupancy per kernel matching $1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c /root/repo/myrrix-recommender_amd/csrc/mals_api.hip -o /tmp/x.o 2>&1 | python3 -c "
import sys,re
pat=sys.argv[1]
cur=None;d={}
for l in sys.stdin:
    if 'error' in l: print(l.strip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d[cur]={}; continue
    if cur and re.search(pat,cur):
        m=re.search(r'remark:\s+(VGPRs|AGPRs|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|SGPRs):\s*(\d+)',l)
        if m: d[cur][m.group(1)[:5]]=m.group(2)
for k,v in d.items():
    if v: print(k[9:48], v)
" "$1"
