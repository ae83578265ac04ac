#!/bin/bash
# Round-3 measurement battery, part 2: rocprofv3 kernel stats + PMC traffic passes of the bench command.
#   measure_r03_prof.sh <tag> <bench args...>        e.g.  measure_r03_prof.sh yolov3_b32_640
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats_$TAG $O/fetch_$TAG $O/write_$TAG
# kernel stats of the DEFAULT command (hipGraph replay); the PMC passes run --eager (one dispatch record per launch)
timeout -k 5 $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $O/stats_$TAG.log 2>&1
cp $(find $O/stats_$TAG -name '*kernel_stats.csv' | head -1) $O/r03_kernel_stats_$TAG.csv
tail -1 $O/stats_$TAG.log | cut -c1-200
timeout -k 5 $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager "$@" > $O/fetch_$TAG.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager "$@" > $O/write_$TAG.log 2>&1
python3 $R/tools/pmc_traffic.py $O/fetch_$TAG $O/write_$TAG > $O/r03_pmc_traffic_$TAG.json
rm -rf $O/fetch_$TAG $O/write_$TAG $O/stats_$TAG
head -12 $O/r03_kernel_stats_$TAG.csv | cut -c1-150
