mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -8 > gpurun_out/r6/pytest_round6.txt
cat gpurun_out/r6/pytest_round6.txt
for w in c4 c2 c3; do for m in 0 1 0 1; do LC_GEMM_TAIL=$m timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w tail=$m', d['ms_per_step'], d['value'], d['roofline'].get('achieved'), d['roofline'].get('frac'), d['breakdown_ms_per_step'])
"; done; done > gpurun_out/r6/gemm_tail_ab.txt 2>&1
cat gpurun_out/r6/gemm_tail_ab.txt
out=gpurun_out/r6
rm -f $out/r6_mfma_util.md
for w in c5 c4; do
    rm -rf $out/pmc_mfma_$w
    timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma_$w -o p -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-secondary --no-cli-corpus > /dev/null 2> $out/pmc_mfma_$w.err
    { echo "# MfmaUtil over ONE $w train step (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; tools/mfma_util.py)"; python3 tools/mfma_util.py $out/pmc_mfma_$w gemm lstm; echo; } >> $out/r6_mfma_util.md 2>&1
    rm -rf $out/pmc_mfma_$w
done
cat $out/r6_mfma_util.md
