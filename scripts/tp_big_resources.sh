#!/bin/bash
# Register / scratch / LDS use of every kernel of the rank-10 time-parallel path (compile-time view):
#   scripts/tp_big_resources.sh > profiles/rNN_tp_big_resources.txt
cd /root/repo/mind_the_gaps_amd/csrc
report() {
python3 -c '
import sys,re,subprocess
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}
    for key in ("VGPRs","AGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]","TotalSGPRs","LDS Size \[bytes/block\]"):
        m=re.search(key+r": (\d+)",l)
        if m and cur: rows[cur][key.split(" ")[0]]=int(m.group(1))
for k,v in rows.items():
    name=subprocess.run(["/usr/bin/c++filt",k],capture_output=True,text=True).stdout.strip().split("(MtgSolveArgs")[0].replace("void ","").replace("(anonymous namespace)::","")
    print("%-46s VGPR %3d AGPR %3d SGPR %3d scratch %5d B/lane  LDS %6d B  waves/SIMD %d"%(name,v.get("VGPRs",0),v.get("AGPRs",0),v.get("TotalSGPRs",0),v.get("ScratchSize",0),v.get("LDS",0),v.get("Occupancy",0)))
'
}
for f in mtg_tp_big_compose mtg_tp_big_compose4 mtg_tp_big_compose4q mtg_tp_big_filter mtg_tp_scan; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f.hip -o /tmp/tpb_res.o -Rpass-analysis=kernel-resource-usage 2>&1 | report
done
