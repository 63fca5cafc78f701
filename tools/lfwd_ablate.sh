#!/bin/bash
# What bounds the hidden-layer forward (GPU box): the kernel rebuilt with parts of its round removed, one and three branches.
# usage: bash tools/lfwd_ablate.sh            (variants are built where hipcc is: run the build half before gpurun)
if [ "$1" = build ]; then
  for a in 16 1 2 4 8 3 7; do bash tools/build_one_variant.sh build_v_lfwd/lfwd_a$a ganet_layer_fwd -DGANET_LFWD_ABLATE=$a > /dev/null || exit 1; done
  exit 0
fi
for a in 16 1 2 4 8 3 7; do GA_DEV=lib_dir=$PWD/build_v_lfwd/lfwd_a$a python tools/lfwd_ablate.py 2>&1 | tail -1; done
