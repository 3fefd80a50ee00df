mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -30 > gpurun_out/r6/pytest_round6.txt
for m in 0 1 0 1; do LC_FUSE_DX=$m timeout 600 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('fuse_dx=$m', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['breakdown_ms_per_step'])
"; done > gpurun_out/r6/c5_fuse_dx_ab.txt 2>&1
cat gpurun_out/r6/pytest_round6.txt gpurun_out/r6/c5_fuse_dx_ab.txt
bash tools/r6_c5_pmc.sh 2>&1 | tail -80
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -5 > gpurun_out/r6/pytest_configs.txt
cat gpurun_out/r6/pytest_configs.txt gpurun_out/r6/grad_tolerance_measured.txt
