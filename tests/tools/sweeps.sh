#!/bin/bash
# Lab tool: the workload sweep of profiles/rNN_final_sweeps.txt (kernel-only bench line per workload).  Usage (GPU box): tests/tools/sweeps.sh [out-file]
cd "$(dirname "$0")/../.."
out=${1:-gpurun_out/sweeps.txt}; mkdir -p "$(dirname "$out")"; : > "$out"
run() {
  timeout 400 python bench.py --no-ab --no-host-path --no-variants --adler-gib 0 --cpu-sample 0 --incremental-decoders 0 --steps 5 "$@" 2>/dev/null | tail -1 |
    python -c "import json,sys;d=json.loads(sys.stdin.read());print('$*', '->', d['value'], 'GiB/s kernel_ms', d.get('kernel_ms', d['ms_per_step']), 'bit_exact', d.get('bit_exact'))" | tee -a "$out"
}
run --workload fixed_4k
run --workload fixed_4k --bundles 0
run --workload fixed_bin
run --workload hetero --streams 32768
run --workload mixed
run --workload html
run --workload skewed_bytes
run --workload runs
run --workload mixed --streams 131072
run --workload l6_32k --gzip
run --workload l6_32k --streams 32768 --blob-bytes 65536
run --workload l6_32k --streams 1048576 --blob-bytes 2048 --pool 4096
run --workload l6_32k --streams 524288 --blob-bytes 4096 --pool 4096
run --workload l6_32k --streams 262144 --blob-bytes 8192 --pool 4096
[ -n "$SWEEP_RINGS" ] && for rb in 12 13 14 15; do run --ring-bits $rb; done
true
