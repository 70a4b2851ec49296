#!/bin/bash
# Build a second copy of the library with extra compile flags on ONE source, for A/B runs inside one gpurun call:
#   bash tools/build_variant.sh pfa0 wfx_mrfft -DWFX_PFA15=0
#   WFX_LIB=wefax_amd/variants/libwefax_hip.pfa0.so python bench.py ...      (tools/exp.sh: 'WFX_LIB=...' as a setting)
set -e
NAME=$1; SRC=$2; shift 2
python -m wefax_amd.build > /dev/null
mkdir -p wefax_amd/variants wefax_amd/csrc/build/variants
O=wefax_amd/csrc/build/variants/$SRC.$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I include -I wefax_amd/csrc "$@" -c wefax_amd/csrc/$SRC.hip -o $O
OBJS=$(ls wefax_amd/csrc/build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o wefax_amd/variants/libwefax_hip.$NAME.so $OBJS $O -ldl -lpthread
ls -la wefax_amd/variants/libwefax_hip.$NAME.so
