#!/bin/bash
# usage: tools/build_variant.sh <outdir> [extra hipcc flags...]  -> alternative libganet_hip.so for A/B runs
# (select it on the GPU box with GA_LIB_DIR=<outdir>)
set -e
out=$1; shift
mkdir -p $out/obj
cp gaussianavatar_amd/_lib/libgsr_hip.so gaussianavatar_amd/_lib/libgalbs_hip.so $out/
objs=""
for f in ganet_bn ganet_wgrad ganet_ssim ganet_mlp ganet_mlp_bwd ganet_pack ganet_upsample ganet_optim; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igaussianavatar_amd/csrc -fhip-fp32-correctly-rounded-divide-sqrt "$@" -c gaussianavatar_amd/csrc/$f.hip -o $out/obj/$f.o
  objs="$objs $out/obj/$f.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libganet_hip.so $objs
echo built $out/libganet_hip.so
