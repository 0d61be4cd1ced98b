#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -k "gated_image_default or golden or bound or config5 or emb_dim" 2>&1 | tail -3) > $O/tests.log; tail -2 $O/tests.log
timeout 900 python bench.py --workload beir --no-cpu-baseline > $O/exact.jsonl 2> $O/exact.err
python3 - <<P
import json
for l in open("$O/exact.jsonl"):
    d=json.loads(l); w=d['config']['workload']; p=d['phase_ms_per_step']; c=d['candidates_per_query']
    print("%-18s %6.1f  gemm %5.1f refine %5.1f rescore %5.1f | bound %6.0f exact %5.0f | fb %s | %s" % (w.split(':')[0][5:], d['ms_per_step'], p['gemm_ms'], p['refine_ms'], p['rescore_ms'], c['bound'], c['exact'], d.get('sample_fallback_queries_per_step'), d['dtype'][:12]))
P
timeout 300 python bench.py --no-cpu-baseline > $O/hyb.json 2> $O/hyb.err; python3 -c "
import json; d=json.loads(open('$O/hyb.json').read().strip().splitlines()[-1]); print('hybrid', d['ms_per_step'], d['dtype'][:20], d['result_checksum'])"
