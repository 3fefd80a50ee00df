#!/bin/bash
# Development helper: lstm_ctc_amd/liblstm_ctc_hip.so.<tag> with lstm.hip recompiled with extra -D flags, linked with the
# objects of the last full build.   tools/lstm_dev_build.sh <tag> [-DLC_BF16_BWD_CS=8 ...];  LC_DEV_LIB=<tag> python tools/persist_probe.py
set -e
cd "$(dirname "$0")/../lstm_ctc_amd/csrc"
tag=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result "$@" -c lstm.hip -o build/lstm_dev_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblstm_ctc_hip.so.$tag build/lstm_dev_$tag.o build/bn.o build/ctc.o build/gemm.o build/gemm_x3.o build/misc.o build/error.o build/tfrecord.o
echo built ../liblstm_ctc_hip.so.$tag
