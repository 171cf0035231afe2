#!/bin/bash
# Diagnostic: the host-buffer path of the headline batch with several range counts / helper-thread counts
nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
for cfg in "0 24" "6 24" "10 24" "16 24" "10 12" "10 48"; do
  set -- $cfg
  r=$1; t=$2
  if [ "$r" = "0" ]; then unset PZG_HOST_RANGES; else export PZG_HOST_RANGES=$r; fi
  PZG_HOST_THREADS=$t PZG_TRACE_HOST=1 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --adler-gib 0 --no-ab --no-verify 2>&1 | grep -E "host path|host_buffers" | tail -2 | sed -e 's/.*"host_buffers_variant": //' | cut -c1-220
done
