#!/bin/bash
# Round 5, after one recurrence changed late (the split-operand XCD-pair BPTT: W=c4x3; the bf16 BPTT's chunk size: W=c5): the
# pieces of tools/evidence_run.sh that depend on it - the default line, the workload's line / kernel statistics / PMC traffic, the
# probes of its kernels, the 200-step runs.   W=<workload> bash tools/r5_evidence_c4x3.sh <tag>   -> gpurun_out/<tag>/
tag=${1:-r5ev3}; r=r5; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cp profiles/r5_pmc_traffic.json $out/${r}_pmc_traffic.json
t0=$(date +%s)
timeout 900 python bench.py > $out/${r}_bench_default.json 2> $out/default.err
echo "default bench.py run: $(( $(date +%s) - t0 )) s wall" > $out/${r}_bench_default_wall.txt
cut -c1-200 $out/${r}_bench_default.json
w=${W:-c4x3}
timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $out/${r}_bench_$w.json 2> $out/$w.err
rm -rf $out/prof_$w
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > $out/prof_$w.json 2> $out/prof_$w.err
f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-400 $f > $out/${r}_bench_${w}_kernel_stats.csv
rm -rf $out/prof_$w
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_${w}_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${w}_$c -o p -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > /dev/null 2> $out/pmc_${w}_$c.err
done
python3 tools/pmc_traffic.py $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE $w $out/${r}_pmc_traffic.json > $out/${r}_pmc_${w}_table.md 2>&1
rm -rf $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE
if [ $w = c4x3 ]; then
    BWD=1 timeout 600 python tools/x3_pair_probe.py > $out/${r}_x3_pair_probe.txt 2>&1
    PN=768 BWD=1 timeout 600 python tools/x3_pair_probe.py > $out/${r}_x3_pair_probe_n768.txt 2>&1
else
    timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_f32.txt 2>&1
    BF16=1 timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_bf16.txt 2>&1
    X3=1 timeout 300 python tools/persist_probe.py > $out/${r}_persist_probe_x3.txt 2>&1
fi
{ echo "# python bench.py --workload <w> --no-secondary --no-cli-corpus --steps 200 --warmup 5 --no-cpu-baseline"; for w in c4x3 c4 c5; do timeout 600 python bench.py --workload $w --no-secondary --no-cli-corpus --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', d['ms_per_step'], d['value'], 'fallbacks', d['config']['persist_fallbacks'], 'loss/label', d['config']['last_loss_per_label'])"; done; } > $out/${r}_long_runs.txt 2>&1
cat $out/${r}_long_runs.txt
