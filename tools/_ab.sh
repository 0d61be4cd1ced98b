#!/bin/bash
O=gpurun_out/ab2; mkdir -p $O
for i in 1 2; do
timeout 900 python bench.py --workload dense --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/dense_$i.json
python -c "import json; d=json.load(open('$O/dense_$i.json')); print('dense', d['ms_per_step'], d['value'], d['result_checksum'], d['phase_ms_per_step'], d['candidates_per_query'])"
DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_base.so timeout 900 python bench.py --workload dense --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/dense_base_$i.json
python -c "import json; d=json.load(open('$O/dense_base_$i.json')); print('dense base', d['ms_per_step'], d['value'], d['result_checksum'], d['phase_ms_per_step'])"
done
