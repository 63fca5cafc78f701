#!/bin/bash
# development: ablation builds of dz_upsample_march_kernel (GANET_DZM_ABLATE bits), timed by tools/microbench_upz.py dz
# run on the build host:  tools/dzm_ablate.sh build     on the GPU box:  tools/dzm_ablate.sh run
if [ "$1" = build ]; then
  for b in 1 3 11 4 16 32 47; do tools/build_one_variant.sh build_v_dzm$b ganet_upz -DGANET_DZM_ABLATE=$b > /dev/null; done
else
  echo "as shipped"; python tools/microbench_upz.py 1 dz 2>&1 | grep dz_up
  for b in 1 3 11 4 16 32 47; do echo "ablate $b"; GA_DEV=lib_dir=build_v_dzm$b python tools/microbench_upz.py 1 dz 2>&1 | grep dz_up; done
fi
