mkdir -p gpurun_out/r3h; O=$PWD/gpurun_out/r3h
( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | tr '\n' ' ')"; sleep 0.3; done ) > $O/bench.smi 2>&1 &
SMI=$!
DHR_DEBUG_PLAN=1 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
kill $SMI
grep "dhr\]" $O/bench.err | tail -40
grep -a "Power" $O/bench.smi | awk '{print $0}' | sed -n '20,60p' | cut -c1-200
