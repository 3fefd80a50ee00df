mkdir -p gpurun_out/r6
( grep -E "MemTotal|MemAvailable" /proc/meminfo; cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory.current 2>&1; nproc; cat /sys/fs/cgroup/cpu.max 2>&1 ) > gpurun_out/r6/host.txt 2>&1
timeout 2400 python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -40 > gpurun_out/r6/pytest_round6.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_ops.py -x -q -k "bf16" 2>&1 | tail -8 > gpurun_out/r6/pytest_bf16.txt
( time timeout 900 python bench.py ) > gpurun_out/r6/bench_default.json 2> gpurun_out/r6/bench_default.err
tail -c 1500 gpurun_out/r6/bench_default.json
cat gpurun_out/r6/pytest_round6.txt gpurun_out/r6/pytest_bf16.txt gpurun_out/r6/host.txt
