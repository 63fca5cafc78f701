#!/bin/bash
# usage: tools/build_one_variant.sh <outdir> <source stem, e.g. ganet_ssim> [-D flags...]: the library that source belongs
# to, with that one source rebuilt with the flags (other objects from the product build); the other libraries copied.
# Select with GA_DEV=lib_dir=<outdir>.
set -e
out=$1; src=$2; shift 2
mkdir -p $out
L=gaussianavatar_amd/_lib
cp $L/libgalbs_hip.so $L/libgsr_hip.so $L/libganet_hip.so $out/
extra=""; [ $src = gsr_preprocess ] && extra="-ffp-contract=off"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igaussianavatar_amd/csrc -fhip-fp32-correctly-rounded-divide-sqrt $extra "$@" \
  -c gaussianavatar_amd/csrc/$src.hip -o $out/$src.o
fam=${src%%_*}
objs=$(ls $L/obj/${fam}_*.o $L/obj/${fam}.o 2>/dev/null | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib${fam}_hip.so $objs $out/$src.o
echo built $out
