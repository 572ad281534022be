#!/bin/bash
# libimpg_phase.so = the library with -DIMPG_PHASE_CLOCKS in kernels.hip (scripts/phase_clocks.py, scripts/stage_clocks.py);
# extra flags for kernels.hip in $EXTRA, output name in $OUT
set -e
cd "$(dirname "$0")/../impg_amd/csrc"
env -u OUT -u EXTRA make -s -j8
O=/tmp/kernels_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -pthread ${EXTRA--DIMPG_PHASE_CLOCKS} -c -o $O kernels.hip
OBJS=$(ls *.o | grep -v '^kernels.o$' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../${OUT-libimpg_phase.so} $O $OBJS -lz -ldl
rm -f $O
