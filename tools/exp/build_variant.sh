#!/bin/bash
# build_variant.sh NAME "-DFLAG ..." [file=ntt]: one source of libmm_hip rebuilt with extra flags, linked with the
# current objects of the others, into tools/exp/NAME/libmm_hip.so (A/B runs: LD_LIBRARY_PATH=tools/exp/NAME)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; FLAGS=$2; F=${3:-ntt}
mkdir -p tools/exp/$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c openmm_amd/csrc/$F.hip -o tools/exp/$NAME/$F.o
OBJS=$(ls openmm_amd/csrc/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/exp/$NAME/libmm_hip.so $OBJS tools/exp/$NAME/$F.o
rm tools/exp/$NAME/$F.o
