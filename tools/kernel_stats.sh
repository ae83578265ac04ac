#!/bin/bash
# rocprofv3 per-kernel device times of `bench.py --profile <args>` (program directly behind `--`): the CSV goes to
# gpurun_out/stats/<tag>.csv.     tools/kernel_stats.sh <tag> [bench args...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stats; mkdir -p $O
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tmp_$TAG
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$TAG -- python3 $R/bench.py --profile "$@" > $O/$TAG.log 2>&1 || { tail -3 $O/$TAG.log; exit 1; }
cp $(find $O/tmp_$TAG -name '*kernel_stats.csv' | head -1) $O/$TAG.csv
rm -rf $O/tmp_$TAG
cut -c1-200 $O/$TAG.csv | head -${LINES_OUT:-24}
