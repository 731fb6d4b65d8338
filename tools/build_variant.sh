#!/bin/bash
# build/ab/NAME.so = the kernel library with extra -D flags on the fused translation unit (force_front.hip), for same-box A/B runs
# usage: tools/build_variant.sh NAME "-DNL_LIST=2048 -DNL_FLUSH=1024"
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
make -s -C openmm_amd/csrc kernels
mkdir -p build/ab/obj_$NAME
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-result -Wno-unused-value $FLAGS -c openmm_amd/csrc/kernels/force_front.hip -o build/ab/obj_$NAME/force_front.o
OTHERS=$(ls build/obj/kernels/*.o | grep -v force_front.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/$NAME.so build/ab/obj_$NAME/force_front.o $OTHERS -ldl
echo "built build/ab/$NAME.so"
