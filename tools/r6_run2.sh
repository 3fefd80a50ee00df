mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -40 > gpurun_out/r6/pytest_round6.txt
timeout 600 python tools/gemm_persist_ab.py > gpurun_out/r6/gemm_persist_ab.txt 2>&1
PROBE_BETA=1 timeout 600 python tools/gemm_persist_ab.py > gpurun_out/r6/gemm_persist_ab_beta1.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -x -q -k "bf16" 2>&1 | tail -8 > gpurun_out/r6/pytest_bf16.txt
for m in 0 1 0 1; do LC_GEMM_BF16_PERSIST=$m timeout 600 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('persist=$m', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['breakdown_ms_per_step'])
"; done > gpurun_out/r6/c5_persist_ab.txt 2>&1
cat gpurun_out/r6/pytest_round6.txt gpurun_out/r6/gemm_persist_ab.txt gpurun_out/r6/gemm_persist_ab_beta1.txt gpurun_out/r6/pytest_bf16.txt gpurun_out/r6/c5_persist_ab.txt
