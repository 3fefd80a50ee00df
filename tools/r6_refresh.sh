# Round 6, after the last kernel changes (one-product fp32 dX, whole-line shadow stores, shadow epilogue without dropout):
# refresh the per-config lines and kernel statistics of c4 / c5 -> gpurun_out/r6ev2/
out=gpurun_out/r6ev2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for w in c5; do
    timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $out/r6_bench_$w.json 2> $out/$w.err
    cut -c1-160 $out/r6_bench_$w.json
done
for w in c4 c5; do
    rm -rf $out/prof_$w
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o p -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > $out/prof_$w.json 2> $out/prof_$w.err
    f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cut -c1-400 $f > $out/r6_bench_${w}_kernel_stats.csv
    rm -rf $out/prof_$w
    head -7 $out/r6_bench_${w}_kernel_stats.csv | cut -c1-200
done
