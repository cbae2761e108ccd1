#!/bin/bash
# Multi-rank rehearsals on ONE card (gpurun): every form the driver or a user may start bench.py in at N > 1.
REPO=$PWD; OUT=$REPO/gpurun_out/r06; mkdir -p $OUT
run() {  # name, then the command
  name=$1; shift
  t0=$(date +%s.%N)
  "$@" > $OUT/rehearse_$name.json 2> $OUT/rehearse_$name.err
  rc=$?
  t1=$(date +%s.%N)
  python3 - "$name" "$rc" "$t0" "$t1" "$OUT/rehearse_$name.json" <<'PY'
import json, sys
name, rc, t0, t1, path = sys.argv[1:]
line = None
for l in open(path):
    if l.startswith("{"):
        line = json.loads(l)
if line is None:
    print("%-28s rc=%s  NO LINE" % (name, rc)); sys.exit(0)
ws = line.get("walker_sharded") or {}
tr = {k.split(" ")[0]: v.get("transport") for k, v in ws.items()}
wf = line.get("workflow_config3_sharded") or {}
inv = (wf.get("world_size_invariance") or {}).get("identical_to_one_rank_alone")
print("%-28s rc=%s  %.0f s  n_gpus=%s value=%.3e  extras_error=%s  transports=%s  p=%s invariance=%s issue_frac=%s"
      % (name, rc, float(t1) - float(t0), line.get("n_gpus"), line.get("value"), line.get("multi_rank_extras_error"), tr,
         wf.get("p_value"), inv, (line.get("roofline") or {}).get("fp64_issue_frac_at_clock")))
line["_rehearsal"] = {"wall_clock_of_the_whole_command_s": float(t1) - float(t0), "rc": int(rc), "command": name}
json.dump(line, open(path, "w"))
PY
}
export MASTER_ADDR=127.0.0.1
run one_rank_rccl            env MTG_BENCH_FORCE_DIST=1 MTG_SHARD_ONE_RANK=1 python3 bench.py --steps 5 --warmup 2 --no-workflow
run one_rank_rccl_injected   env MTG_BENCH_FORCE_DIST=1 MTG_SHARD_ONE_RANK=1 MTG_SHARD_FAIL_RCCL=all python3 bench.py --steps 5 --warmup 2 --no-workflow
run 2ranks_one_gpu           python3 bench.py --gpus 2 --steps 5 --warmup 2
run 2ranks_torchrun_one_gpu  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2
run 2ranks_failed_extras     env MTG_BENCH_FAIL_EXTRAS=1 python3 bench.py --gpus 2 --steps 5 --warmup 2
run 6ranks_one_gpu           python3 bench.py --gpus 6 --steps 5 --warmup 2
