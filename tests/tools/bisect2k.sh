#!/bin/bash
# Diagnostic: 262,144 x 2 KiB streams through several builds (build/exp/libpzg_<tag>.so): how many streams come out wrong
for tag in "$@"; do
  echo "== $tag"; PZG_LIB=$PWD/build/exp/libpzg_$tag.so python tests/tools/dbg2k.py 262144 2>&1 | tail -2
done
