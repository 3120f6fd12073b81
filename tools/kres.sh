#!/bin/bash
# usage: tools/kres.sh file.hip [extra hipcc flags]  -- per-kernel VGPR / AGPR / spill / scratch / occupancy of the explicit instantiations
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m:
        if "error" in ln: print(ln.rstrip())
        continue
    t=m.group(1)
    if t.startswith("Function Name:"):
        cur={"name":t.split(": ",1)[1]}; rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.split(":",1); cur[k.strip()]=v.strip()
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    if "persistent" in n or "dual" in n or "finish" in n or "wide" in n or "gramian_" in n or "refine" in n:
        print("%-70s vgpr %4s agpr %3s vspill %4s scratch %5s occ %s lds %s" % (n[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
