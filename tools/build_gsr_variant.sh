#!/bin/bash
# usage: tools/build_gsr_variant.sh <outdir> [extra hipcc flags...]  -> <outdir>/libgsr_hip.so rebuilt with the flags
# (all five sources: constants such as -DGSR_SUB=4 reach the layout in gsr_api.hip too); the other libraries are copied.
# Select on the GPU box with GA_DEV=lib_dir=<outdir>.
set -e
out=$1; shift
mkdir -p $out/obj
cp gaussianavatar_amd/_lib/libgalbs_hip.so gaussianavatar_amd/_lib/libganet_hip.so $out/
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igaussianavatar_amd/csrc -fhip-fp32-correctly-rounded-divide-sqrt"
objs=""
for f in gsr_api gsr_preprocess gsr_binning gsr_render gsr_sh; do
  extra=""; [ $f = gsr_preprocess ] && extra="-ffp-contract=off"
  hipcc $common $extra "$@" -c gaussianavatar_amd/csrc/$f.hip -o $out/obj/$f.o &
  objs="$objs $out/obj/$f.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgsr_hip.so $objs
echo built $out/libgsr_hip.so
