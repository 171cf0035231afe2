#!/bin/bash
# The 1 -> 8 GPU curve of SURVEY.md 8e on a node that has the GPUs: bench.py at N = 1, 2, 4, 8 for BASELINE
# config 4 (65,536 x 32 KiB level-6 blobs per GPU) and config 5 (131,072 mixed 1-64 KiB blobs per GPU, 1 M on 8 GPUs), one table.
# One process per GPU under torch.distributed.run (RCCL only for the barrier and the max-over-ranks time: the streams are
# independent, no data-path collective).  N values above the visible device count are skipped.
#   tests/tools/scale_node.sh [steps] [warmup]
root=$(cd "$(dirname "$0")/../.." && pwd)
steps=${1:-10}; warm=${2:-3}
ndev=$(python3 -c "import torch; print(torch.cuda.device_count())")
export HSA_ENABLE_IPC_MODE_LEGACY=0
printf "%-10s %5s %14s %12s %10s\n" workload gpus "GiB/s (node)" "ms/step" bit_exact
for wl in "l6_32k:--workload l6_32k" "mixed:--workload mixed --streams 131072"; do
  name=${wl%%:*}; args=${wl#*:}
  base=
  for n in 1 2 4 8; do
    [ "$n" -gt "$ndev" ] && continue
    common="--gpus $n --steps $steps --warmup $warm --no-ab --no-host-path --no-variants --cpu-sample 0 --adler-gib 0 $args"
    if [ "$n" = 1 ]; then line=$(python3 $root/bench.py $common 2>/dev/null | tail -1)
    else line=$(python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) $root/bench.py $common 2>/dev/null | grep '^{' | tail -1); fi
    python3 - "$name" "$n" "$line" "$base" <<'PY'
import json, sys
name, n, line, base = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
d = json.loads(line)
eff = "" if not base else "  (%.0f %% of %d x the 1-GPU rate)" % (100 * d["value"] / (float(base) * n), n)
print("%-10s %5d %14.1f %12.3f %10s%s" % (name, n, d["value"], d["ms_per_step"], d["bit_exact"], eff))
PY
    [ "$n" = 1 ] && base=$(python3 -c "import json,sys; print(json.loads(sys.argv[1])['value'])" "$line")
  done
done
