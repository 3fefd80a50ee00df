#!/bin/bash
# PMC passes over tools/gemm_x3_pmc.py (run on the GPU box from the repo root); prints per-kernel counter means and durations.
export TMPDIR=/tmp
out=gpurun_out/x3pmc
mkdir -p $out
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY"; do
    n=$(echo $c | tr " " "_")
    rm -rf $out/$n
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$n -o p -- python3 tools/gemm_x3_pmc.py > /dev/null 2> $out/$n.err || tail -3 $out/$n.err
done
python3 tools/pmc_summary.py $out
python3 - <<PY
import csv,glob
for f in glob.glob("$out/GRBM*/**/*kernel_trace.csv", recursive=True):
    d={}
    for r in csv.DictReader(open(f)):
        d.setdefault(r["Kernel_Name"][:60],[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in d.items(): print(k, len(v), sum(v)/len(v)/1e3, "us")
PY
