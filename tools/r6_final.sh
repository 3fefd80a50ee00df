mkdir -p gpurun_out/r6final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6final/smoke.txt 2>&1; tail -2 gpurun_out/r6final/smoke.txt
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r6final/r6_pytest_gpu.log 2>&1; tail -4 gpurun_out/r6final/r6_pytest_gpu.log
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/r6final/r6_bench_default.json 2> gpurun_out/r6final/default.err
echo "default bench.py run: $(( $(date +%s) - t0 )) s wall" > gpurun_out/r6final/r6_bench_default_wall.txt
tail -c 900 gpurun_out/r6final/r6_bench_default.json; cat gpurun_out/r6final/r6_bench_default_wall.txt
cp gpurun_out/r6/grad_tolerance_measured.txt gpurun_out/r6final/ 2>/dev/null
