#!/bin/bash
# Development: step times of the workloads whose recurrences request their saved operands a step ahead (round 5), and the
# split-operand forward recurrence's width rule (LC_X3_FWD_MIN_N=0: the split-operand kernel at every width).
out=gpurun_out/${1:-r5r}; mkdir -p $out
for w in c5 c2 c3 c1 c2x3 c3x3; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-cli-corpus 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', d['ms_per_step'], 'ms', d['value'], 'frames/s', d['config']['lstm_schedule'], 'fallbacks', d['config']['persist_fallbacks'], d.get('breakdown_ms_per_step'))"
done > $out/ahead_bench.txt 2>&1
for w in c2x3 c3x3; do
  LC_X3_FWD_MIN_N=0 timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-cli-corpus 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w LC_X3_FWD_MIN_N=0', d['ms_per_step'], 'ms', d['config']['lstm_schedule'], d.get('breakdown_ms_per_step'))"
done >> $out/ahead_bench.txt 2>&1
cat $out/ahead_bench.txt
