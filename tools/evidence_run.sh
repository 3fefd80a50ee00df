#!/bin/bash
# Round-end evidence on the GPU box (development only): the full GPU test suite, one bench line per BASELINE config, the
# rocprofv3 kernel statistics of the same commands, the PMC traffic passes, and the probes DESIGN.md quotes.
#   gpurun --timeout 3000 -- 'bash tools/evidence_run.sh <tag>'      -> gpurun_out/<tag>/
tag=${1:-evidence}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
for w in c4 c1 c2 c3 c5; do
    timeout 600 python bench.py --workload $w --steps 20 --warmup 5 > $out/r2_bench_$w.json 2> $out/$w.err
    cut -c1-160 $out/r2_bench_$w.json
done
for w in c2 c3 c4 c5; do
    rm -rf $out/prof_$w
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-profile > $out/prof_$w.json 2> $out/prof_$w.err
    f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $out/r2_bench_${w}_kernel_stats.csv
    rm -rf $out/prof_$w
done
for w in c4 c5; do
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $out/pmc_${w}_$c
        timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${w}_$c -o p -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> $out/pmc_${w}_$c.err
    done
    python3 tools/pmc_traffic.py $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE $w $out/r2_pmc_traffic.json > $out/r2_pmc_${w}_table.md 2>&1
    rm -rf $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE
done
timeout 300 python tools/persist_probe.py > $out/r2_persist_probe_f32.txt 2>&1
BF16=1 timeout 300 python tools/persist_probe.py > $out/r2_persist_probe_bf16.txt 2>&1
timeout 300 python tools/pair_probe.py > $out/r2_pair_probe.txt 2>&1
BWD=1 timeout 300 python tools/pair_probe.py >> $out/r2_pair_probe.txt 2>&1
timeout 500 python tools/probe.py gemm_bf16 ctc > $out/r2_probe_gemm_ctc.txt 2>&1
ls -la $out | head -50
