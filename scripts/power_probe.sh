#!/bin/bash
# Board power and shader clock while the bench kernel runs back to back (evidence for DESIGN.md: the
# sweep is power bound).  Samples rocm-smi a few times per second next to a 300-step bench run.
OUT=$PWD/gpurun_out/power; mkdir -p $OUT
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | head -c 4000; echo; sleep 0.25; done ) > $OUT/smi.jsonl &
SMI=$!
python bench.py --steps 300 --warmup 5 --cpu-seconds 0 > $OUT/bench.log 2>&1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
tail -1 $OUT/bench.log | cut -c1-200
python3 - <<PY
import json
rows=[]
for l in open("$OUT/smi.jsonl"):
    l=l.strip()
    if not l.startswith("{"): continue
    try: d=json.loads(l)
    except Exception: continue
    c=d.get("card0",{})
    rows.append({k:v for k,v in c.items() if any(s in k.lower() for s in ("power","sclk","use","mclk"))})
print(len(rows),"samples")
for r in rows[::3][:25]: print(r)
PY
