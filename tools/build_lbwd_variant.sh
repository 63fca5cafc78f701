#!/bin/bash
# usage: tools/build_lbwd_variant.sh <outdir> [-D flags...]: libganet_hip.so with ganet_layer_bwd.hip rebuilt with the
# flags (other objects from the product build); libgsr / libgalbs copied. Select with GA_DEV=lib_dir=<outdir>.
set -e
out=$1; shift
mkdir -p $out
L=gaussianavatar_amd/_lib
cp $L/libgalbs_hip.so $L/libgsr_hip.so $out/
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igaussianavatar_amd/csrc -fhip-fp32-correctly-rounded-divide-sqrt "$@" \
  -c gaussianavatar_amd/csrc/ganet_layer_bwd.hip -o $out/ganet_layer_bwd.o
objs=$(ls $L/obj/ganet_*.o | grep -v ganet_layer_bwd)
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libganet_hip.so $objs $out/ganet_layer_bwd.o
echo built $out
