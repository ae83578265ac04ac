#!/bin/bash
# Build-container side: variant copies of the library with parts of the decoding head node's channel-block loop removed
# (csrc/sepconv.hip SP_ABL_*), into gpurun_out/abl/.  GPU side: `tools/ablate_sepconv.sh run` times the D1 step with each.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" != "run" ]; then
  mkdir -p $R/mydetection_amd/lib/abl
  for v in NOLOAD NOMFMA; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -ffp-contract=off -DSP_ABL_$v -c $R/mydetection_amd/csrc/sepconv.hip -o /tmp/sepconv_$v.o
    objs=$(ls $R/mydetection_amd/lib/obj/*.o | grep -v sepconv.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/mydetection_amd/lib/abl/libmydet_$v.so $objs /tmp/sepconv_$v.o
  done
  exit 0
fi
T=${MYDET_TOOL_TIMEOUT:-200}
for v in "" NOLOAD NOMFMA; do
  [ -n "$v" ] && export MYDET_LIB_PATH=$R/mydetection_amd/lib/abl/libmydet_$v.so
  timeout -k 5 $T python $R/bench.py --config efficientdet-d1 --steps 20 --warmup 3 --no-cpu-baseline --eager 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-full}', d['ms_per_step'], d['stages']['sepconv_decode'])"
done
