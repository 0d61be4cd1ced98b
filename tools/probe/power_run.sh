#!/bin/bash
# Runs a probe while sampling package power / shader clock with rocm-smi (one sample per ~0.5 s, wall-clock stamped).
# usage: bash tools/probe/power_run.sh <out-prefix> <command...>
OUT=$1; shift
( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | tr '\n' ' ')"; sleep 0.4; done ) > ${OUT}.smi 2>&1 &
SMI=$!
"$@" > ${OUT}.log 2>&1
kill $SMI
