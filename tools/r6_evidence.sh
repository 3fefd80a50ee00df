#!/bin/bash
# Round-6 evidence on the GPU box: the default bench line, one line per BASELINE config, rocprofv3 kernel statistics of
# the same commands, PMC traffic passes (c4, c5, CTC at B = 512), MfmaUtil of a c5 step's product kernels, the probes
# DESIGN.md quotes.   gpurun --timeout 3300 -- 'bash tools/r6_evidence.sh'   -> gpurun_out/r6ev/ (copy into profiles/)
r=r6
out=gpurun_out/r6ev
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
t0=$(date +%s)
timeout 900 python bench.py > $out/${r}_bench_default.json 2> $out/default.err
echo "default bench.py run: $(( $(date +%s) - t0 )) s wall" > $out/${r}_bench_default_wall.txt
tail -c 1200 $out/${r}_bench_default.json; cat $out/${r}_bench_default_wall.txt
for w in c1 c2 c3 c5 c4x3; do
    timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $out/${r}_bench_$w.json 2> $out/$w.err
    cut -c1-160 $out/${r}_bench_$w.json
done
for w in c2 c3 c4 c5; do
    rm -rf $out/prof_$w
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > $out/prof_$w.json 2> $out/prof_$w.err
    f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cut -c1-400 $f > $out/${r}_bench_${w}_kernel_stats.csv
    rm -rf $out/prof_$w
done
for w in c4 c5; do
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $out/pmc_${w}_$c
        timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${w}_$c -o p -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > /dev/null 2> $out/pmc_${w}_$c.err
    done
    python3 tools/pmc_traffic.py $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE $w $out/${r}_pmc_traffic.json > $out/${r}_pmc_${w}_table.md 2>&1
    rm -rf $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE
done
# MFMA utilisation of the product kernels inside a c5 step and a c4 step
for w in c5 c4; do
    rm -rf $out/pmc_mfma_$w
    timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma_$w -o p -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > /dev/null 2> $out/pmc_mfma_$w.err
    { echo "# MfmaUtil over ONE $w train step (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; tools/mfma_util.py)"; python3 tools/mfma_util.py $out/pmc_mfma_$w gemm lstm; } >> $out/${r}_mfma_util.md 2>&1
    rm -rf $out/pmc_mfma_$w
done
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
timeout 300 python tools/gemm_tn_probe.py > $out/${r}_gemm_tn_probe.txt 2>&1
timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_f32.txt 2>&1
BF16=1 timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_bf16.txt 2>&1
BF16=1 SHADOW=1 timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_bf16_shadow.txt 2>&1
timeout 300 python tools/pair_probe.py > $out/${r}_pair_probe.txt 2>&1
BWD=1 timeout 300 python tools/pair_probe.py >> $out/${r}_pair_probe.txt 2>&1
timeout 500 python tools/probe.py gemm_bf16 ctc > $out/${r}_probe_gemm_ctc.txt 2>&1
{ echo "# python bench.py --workload <w> --no-secondary --no-cli-corpus --steps 200 --warmup 5 --no-cpu-baseline"; for w in c4 c5; do timeout 600 python bench.py --workload $w --no-secondary --no-cli-corpus --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', d['ms_per_step'], d['value'], 'fallbacks', d['config']['persist_fallbacks'], 'loss/label', d['config']['last_loss_per_label'])"; done; } > $out/${r}_long_runs.txt 2>&1
cat $out/${r}_long_runs.txt $out/${r}_mfma_util.md
ls $out | head -60
