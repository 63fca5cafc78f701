#!/bin/bash
# Round 5's judged artefacts in one GPU call -> gpurun_out/<tag>/ (copy what is kept into profiles/r05_*):
#   everything tools/collect_final_r04.sh collects (full -m gpu suite, smoke, the driver's bench command, kernel stats + gaps,
#   PMC traffic, config 2 / stage 2 / config 5 lines + kernel tables, the reference's train.py verbatim, a 2000-step run)
#   + the pose encoder's launches of one stage-2 iteration in order (tools/unet_calls.sh)
#   + the stage-2 A/B of the pose encoder's weight-gradient side stream (tools/r05_s2ab.sh)
# usage: bash tools/collect_final_r05.sh r05_final
tag=${1:-r05_final}
bash tools/collect_final_r04.sh $tag
mkdir -p gpurun_out/$tag
bash tools/unet_calls.sh > /dev/null 2>&1; cp gpurun_out/r05unet/calls.txt gpurun_out/$tag/unet_calls.txt
bash tools/r05_s2ab.sh > gpurun_out/$tag/stage2_unet_side_stream_ab.txt 2>&1; tail -6 gpurun_out/$tag/stage2_unet_side_stream_ab.txt
