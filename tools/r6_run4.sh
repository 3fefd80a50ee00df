mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -15 > gpurun_out/r6/pytest_round6.txt
cat gpurun_out/r6/pytest_round6.txt
for m in 0 1 0 1; do LC_C5_SHADOW_ONLY=$m timeout 600 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('shadow_only=$m', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['breakdown_ms_per_step'], d['config']['last_loss_per_label'])
"; done > gpurun_out/r6/c5_shadow_only_ab.txt 2>&1
cat gpurun_out/r6/c5_shadow_only_ab.txt
( PROBE_SHAPES="32,320;32,384;32,448;32,512" timeout 300 python tools/persist_probe.py; echo "--- X3=1"; X3=1 PROBE_SHAPES="32,320;32,384;32,448;32,512" timeout 300 python tools/persist_probe.py ) > gpurun_out/r6/x3_width_probe.txt 2>&1
grep -E "^B=|X3" gpurun_out/r6/x3_width_probe.txt
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -25 > gpurun_out/r6/pytest_gpu.log
tail -25 gpurun_out/r6/pytest_gpu.log; head -12 gpurun_out/r6/grad_tolerance_measured.txt
