#!/bin/bash
# Diagnostic: A/B of experimental builds (build/exp/libpzg_<tag>.so) in ONE GPU session, interleaved, REPS passes.
# Usage: tests/tools/r3_ab.sh "<tag> <tag> ..." [out-tag] ; workloads: text 32 KiB, html, 1 M x 2 KiB, literal-heavy, fixed 4 KiB
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out/${2:-ab}.txt; mkdir -p $root/gpurun_out; : > $out
run() { # tag, label, args...
  tag=$1; label=$2; shift 2
  PZG_LIB=$root/build/exp/libpzg_$tag.so timeout 300 python3 $root/bench.py --steps 8 --warmup 2 --no-ab --no-host-path --no-variants --cpu-sample 0 --adler-gib 0 "$@" 2>/dev/null | tail -1 |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$label', d['value'], d['bit_exact'])" >> $out
}
for rep in $(seq 1 ${REPS:-2}); do
  for tag in $1; do
    run $tag text --workload l6_32k
    run $tag html --workload html
    run $tag 2k --workload l6_32k --streams 1048576 --blob-bytes 2048 --pool 4096
    run $tag skew --workload skewed_bytes
    run $tag fixed --workload fixed_4k
  done
done
python3 - $out <<'PY'
import sys, collections
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    t, w, v, ok = l.split()
    acc[(t, w)].append(float(v)); assert ok == "True", l
tags = sorted({k[0] for k in acc}); wls = ["text", "html", "2k", "skew", "fixed"]
print("%-14s" % "variant" + "".join("%10s" % w for w in wls))
for t in tags:
    print("%-14s" % t + "".join("%10.1f" % (sum(acc[(t, w)]) / len(acc[(t, w)])) for w in wls))
PY
