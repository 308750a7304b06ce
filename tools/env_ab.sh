#!/bin/bash
# A/B of environment switches on ONE box: tools/env_ab.sh OUT CURVE LOG_N STEPS "VAR=VAL VAR=VAL" "VAR=VAL" ...
# (each quoted group is one configuration; "-" = library defaults); three rounds, interleaved.
O=$1; shift; C=$1; shift; L=$1; shift; S=$1; shift
for round in 1 2 3; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    echo "== round $round [$cfg]" >> $O
    env $e timeout 600 python tools/msm_bench.py $C $L $S plain 2>/dev/null | grep -v "^$" >> $O
  done
done
