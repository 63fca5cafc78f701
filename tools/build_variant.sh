#!/bin/bash
# usage: tools/build_variant.sh <outdir> [extra hipcc flags...]
#   -> alternative libganet_hip.so AND libgsr_hip.so built with the extra flags, for A/B runs
#      (select them on the GPU box with GA_DEV=lib_dir=<outdir>); libgalbs_hip.so is copied.
#   e.g. tools/build_variant.sh build_ablate -DGSR_ABLATE_BUILD   (then GSR_ABLATE=<bits> is honoured)
set -e
out=$1; shift
mkdir -p $out/obj
cp gaussianavatar_amd/_lib/libgalbs_hip.so $out/
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igaussianavatar_amd/csrc -fhip-fp32-correctly-rounded-divide-sqrt"
objs=""
for f in ganet_bn ganet_wgrad ganet_ssim ganet_mlp ganet_mlp_bwd ganet_mlp_split ganet_wgrad_split ganet_layer_bwd ganet_layer_fwd ganet_decoder ganet_pack ganet_upsample ganet_optim ganet_conv; do
  [ -f gaussianavatar_amd/csrc/$f.hip ] || continue
  hipcc $common "$@" -c gaussianavatar_amd/csrc/$f.hip -o $out/obj/$f.o
  objs="$objs $out/obj/$f.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libganet_hip.so $objs
objs=""
for f in gsr_api gsr_preprocess gsr_binning gsr_render gsr_sh; do
  extra=""; [ $f = gsr_preprocess ] && extra="-ffp-contract=off"
  hipcc $common $extra "$@" -c gaussianavatar_amd/csrc/$f.hip -o $out/obj/$f.o
  objs="$objs $out/obj/$f.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgsr_hip.so $objs
echo built $out/libganet_hip.so $out/libgsr_hip.so
