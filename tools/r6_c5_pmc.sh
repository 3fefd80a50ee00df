#!/bin/bash
# Round 6: PMC passes over the c5 BPTT with / without the dz shadow (run on the GPU box from the repo root).
export TMPDIR=/tmp
out=gpurun_out/r6/c5pmc
mkdir -p $out
rocprofv3 -L > $out/counters_all.txt 2>&1
grep -oE "\b(TCC|TCP|TA|TD|SQ|SPI|GRBM)_[A-Za-z0-9_]+" $out/counters_all.txt | sort -u > $out/counter_names.txt
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" \
         "SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum" \
         "TCC_HIT_sum TCC_MISS_sum" "TCC_WRITE_sum TCC_WRITEBACK_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
         "WRITE_SIZE" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1)); d=$out/p$i
    rm -rf $d
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 tools/c5_bptt_pmc.py > $d.out 2> $d.err || { echo "pass $i ($c) failed:"; tail -2 $d.err; }
done
python3 tools/c5_bptt_pmc_summary.py $out > gpurun_out/r6/c5_bptt_pmc.txt 2>&1
cat gpurun_out/r6/c5_bptt_pmc.txt
