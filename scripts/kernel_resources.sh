#!/bin/bash
# Prints VGPR / scratch / occupancy of every solve instantiation (compile-time view).
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c /root/repo/mind_the_gaps_amd/csrc/mtg_kernels.hip -o /tmp/k_res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}
    for key in ("VGPRs","AGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]","TotalSGPRs"):
        m=re.search(key+r": (\d+)",l)
        if m and cur: rows[cur][key.split(" ")[0]]=int(m.group(1))
for k,v in rows.items():
    m=re.search(r"ILi(\d+)ELi(\d+)E",k)
    name=("solve<%s,%s>"%m.groups()) if m else k[:30]
    print("%-14s VGPR %3d AGPR %3d SGPR %3d scratch %4d occ %d"%(name,v.get("VGPRs",0),v.get("AGPRs",0),v.get("TotalSGPRs",0),v.get("ScratchSize",0),v.get("Occupancy",0)))
'
