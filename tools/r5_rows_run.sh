#!/bin/bash
# Development: the single-XCD recurrences after "rows that do not exist read row 0's piece" - timing + the parity suites that cover them
out=gpurun_out/${1:-r5i}; mkdir -p $out
{ echo "== f32"; timeout 300 python tools/persist_probe.py; echo "== x3"; X3=1 timeout 300 python tools/persist_probe.py; echo "== bf16"; BF16=1 timeout 300 python tools/persist_probe.py; } 2>&1 | grep -v amdgpu > $out/persist_probe_rows.txt
cat $out/persist_probe_rows.txt
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -x -q -k "persist or recurrence or lstm or config or chain or random_shape or bf16 or x3 or split" > $out/pytest_rows.txt 2>&1; tail -5 $out/pytest_rows.txt
for w in c1 c2 c3 c2x3 c3x3; do timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'], d.get('breakdown_ms_per_step'))"; done > $out/bench_rows.txt 2>&1; cat $out/bench_rows.txt
