#!/bin/bash
# Development helper: lstm_ctc_amd/liblstm_ctc_hip.so.<tag> with gemm_x3.hip recompiled with extra -D flags (seconds), linked
# with the objects of the last full build.   tools/x3_dev_build.sh <tag> [-DLC_X3_NOFILL=1 ...];  LC_DEV_LIB=<tag> python tools/gemm_x3_probe.py
set -e
cd "$(dirname "$0")/../lstm_ctc_amd/csrc"
tag=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result "$@" -c gemm_x3.hip -o build/x3_dev_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblstm_ctc_hip.so.$tag build/x3_dev_$tag.o build/bn.o build/ctc.o build/gemm.o build/lstm.o build/misc.o build/error.o build/tfrecord.o
echo built ../liblstm_ctc_hip.so.$tag
