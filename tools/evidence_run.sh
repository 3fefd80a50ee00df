#!/bin/bash
# Round-end evidence on the GPU box (development only): the full GPU test suite, the default bench line (headline c4 +
# secondary c5 / c4x3 / c2 / c3 + the bin/nnet-train.py corpus leg) and one line per BASELINE config, the rocprofv3 kernel
# statistics of the same commands, the PMC traffic passes, and the probes DESIGN.md quotes.
#   gpurun --timeout 3300 -- 'bash tools/evidence_run.sh <tag> [round-prefix]'      -> gpurun_out/<tag>/
tag=${1:-evidence}
r=${2:-r5}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ -z "$SKIP_TESTS" ]; then
    timeout 1800 python -m pytest tests -m gpu -x -q > $out/${r}_pytest_gpu.log 2>&1; tail -3 $out/${r}_pytest_gpu.log
fi
t0=$(date +%s)
timeout 900 python bench.py > $out/${r}_bench_default.json 2> $out/default.err
echo "default bench.py run: $(( $(date +%s) - t0 )) s wall" > $out/${r}_bench_default_wall.txt
cut -c1-200 $out/${r}_bench_default.json
for w in c1 c2 c3 c5 c4x3 c3x3 c2x3; do
    timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $out/${r}_bench_$w.json 2> $out/$w.err
    cut -c1-160 $out/${r}_bench_$w.json
done
for w in c2 c3 c4 c5 c4x3 c3x3 c2x3; do
    rm -rf $out/prof_$w
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > $out/prof_$w.json 2> $out/prof_$w.err
    f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cut -c1-400 $f > $out/${r}_bench_${w}_kernel_stats.csv
    rm -rf $out/prof_$w
done
for w in c4 c5 c4x3 c3x3; do
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $out/pmc_${w}_$c
        timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${w}_$c -o p -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > /dev/null 2> $out/pmc_${w}_$c.err
    done
    python3 tools/pmc_traffic.py $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE $w $out/${r}_pmc_traffic.json > $out/${r}_pmc_${w}_table.md 2>&1
    rm -rf $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE
done
# the CTC op alone at the B = 512 shape of `roofline_ctc.large_batch`: kernel statistics and HBM-side traffic
rm -rf $out/prof_ctc
CTC_SHAPES="512,100" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ctc -o p -- python3 tools/ctc_probe.py > /dev/null 2> $out/prof_ctc.err
f=$(find $out/prof_ctc -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -E "^\"Name|ctc_" $f | cut -c1-300 > $out/${r}_ctc_b512_kernel_stats.csv
rm -rf $out/prof_ctc
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_ctc_$c
    CTC_SHAPES="512,100" timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_ctc_$c -o p -- python3 tools/ctc_probe.py > /dev/null 2> $out/pmc_ctc_$c.err
done
python3 tools/pmc_traffic.py $out/pmc_ctc_FETCH_SIZE $out/pmc_ctc_WRITE_SIZE ctc_b512 $out/${r}_pmc_traffic.json > $out/${r}_pmc_ctc_b512_table.md 2>&1
rm -rf $out/pmc_ctc_FETCH_SIZE $out/pmc_ctc_WRITE_SIZE
timeout 300 python tools/ctc_probe.py > $out/${r}_ctc_probe.txt 2>&1
# the frame statistics in phase 1 (round 3) / in phase 2 (round 4, default from 512 utterances) at the same shapes
for v in 0 1; do echo "LC_CTC_LSE2=$v"; LC_CTC_LSE2=$v CTC_SHAPES="256,100;512,100;1024,100" timeout 300 python tools/ctc_probe.py 2>&1 | grep "^ctc"; done >> $out/${r}_ctc_probe.txt
CTC_B=512 timeout 300 python tools/ctc_stamps.py > $out/${r}_ctc_stamps_b512.txt 2>&1
timeout 300 python tools/gemm_tn_probe.py > $out/${r}_gemm_tn_probe.txt 2>&1
timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_f32.txt 2>&1
BF16=1 timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_bf16.txt 2>&1
X3=1 timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_x3.txt 2>&1
timeout 300 python tools/pair_probe.py > $out/${r}_pair_probe.txt 2>&1
BWD=1 timeout 300 python tools/pair_probe.py >> $out/${r}_pair_probe.txt 2>&1
# split-operand recurrences next to the fp32 pair kernels: float64 teacher-forced errors, us per step, four-workgroup anatomy
BWD=1 timeout 600 python tools/x3_pair_probe.py > $out/${r}_x3_pair_probe.txt 2>&1
PN=768 BWD=1 timeout 600 python tools/x3_pair_probe.py > $out/${r}_x3_pair_probe_n768.txt 2>&1
timeout 500 python tools/probe.py gemm_bf16 ctc > $out/${r}_probe_gemm_ctc.txt 2>&1
# fp32 products as bf16x3: kernel rates + error against float64, long-sequence error of the three modes against the oracle,
# fp32 / bf16x3 / fp32-with-another-summation-order at full c4 size, PMC counters of the product kernels
timeout 600 python tools/gemm_x3_probe.py > $out/${r}_gemm_x3_probe.txt 2>&1
timeout 900 python tools/x3_vs_oracle.py > $out/${r}_x3_vs_oracle.txt 2>&1
timeout 600 python tools/x3_model_check.py > $out/${r}_x3_model_check.txt 2>&1
# round 5: float64 TRUTH at the benched sizes (every mode against oracle/torch_f64.py, incl. an independent fp32 implementation),
# the per-step (teacher-forced) error of the fp32 and split-operand recurrences over a whole T = 1000 trajectory, a training
# curve and 200-step runs on the current recurrences
timeout 900 python tools/x3_truth.py > $out/${r}_x3_truth_final.txt 2>&1
timeout 600 python tools/x3_local_error.py > $out/${r}_x3_local_error_final.txt 2>&1
timeout 900 python tools/x3_train_curve.py > $out/${r}_x3_train_curve.txt 2>&1
{ echo "# python bench.py --workload <w> --no-secondary --no-cli-corpus --steps 200 --warmup 5 --no-cpu-baseline"; for w in c4x3 c4 c5; do timeout 600 python bench.py --workload $w --no-secondary --no-cli-corpus --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', d['ms_per_step'], d['value'], 'fallbacks', d['config']['persist_fallbacks'], 'loss/label', d['config']['last_loss_per_label'])"; done; } > $out/${r}_long_runs.txt 2>&1
bash tools/x3_pmc_run.sh 2>&1 | grep -v "^    .*n=1 \|n=2 " > $out/${r}_x3_pmc.txt
rm -rf gpurun_out/x3pmc
ls -la $out | head -60
