#!/bin/bash
# The whole round-3 battery in one GPU call: bench lines, then kernel stats + PMC traffic of the three model configs.
set -e
R=$GRAFT_REPO_ROOT
$R/tools/measure_r03.sh
$R/tools/measure_r03_prof.sh yolov3_b32_640
$R/tools/measure_r03_prof.sh d1_b16_640 --config efficientdet-d1
$R/tools/measure_r03_prof.sh fcos_b32_640 --config d1_fcs2_atss
