#!/bin/bash
# Lab tool: rocprofv3 kernel stats of one bench.py command (kernel-only legs).  Usage (GPU box): tests/tools/kstats.sh <out.csv> [bench args...]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$1; shift
mkdir -p "$(dirname "$root/$out")"
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants --incremental-decoders 0 --adler-gib 0 "$@" > /tmp/ks.log 2>&1
tail -1 /tmp/ks.log | cut -c1-300
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $root/$out && cut -d, -f1-8 $root/$out | head -12
