#!/bin/bash
# Development helper: builds lstm_ctc_amd/liblstm_ctc_hip.so.<tag> from ctc.hip compiled with -DLC_CTC_DEV (three kernel
# instantiations: ~20 s instead of 2 min) plus extra -D flags, linked with the objects of the last full build.
#   tools/ctc_dev_build.sh <tag> [-DLC_CTC_EXP=1 ...]      then:   LC_DEV_LIB=<tag> python tools/ctc_probe.py
set -e
cd "$(dirname "$0")/../lstm_ctc_amd/csrc"
tag=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DLC_CTC_DEV=1 "$@" -c ctc.hip -o build/ctc_dev_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblstm_ctc_hip.so.$tag build/ctc_dev_$tag.o build/bn.o build/gemm.o build/lstm.o build/misc.o build/error.o build/tfrecord.o
echo built ../liblstm_ctc_hip.so.$tag
