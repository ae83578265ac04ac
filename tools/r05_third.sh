#!/bin/bash
# Round-5 third GPU call: SE tail with the shipped rule (A/B), full -m gpu suite, default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
: > $O/se_ab2.txt
for rep in 1 2; do
for v in 1 0; do
  for spec in "efficientdet-d1 16" "d1_fcs2_atss 32"; do
    set -- $spec
    MYDET_SE_IN_DW=$v timeout -k 10 120 python bench.py --config $1 --batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --parity-images 2 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('se_in_dw=$v (>=1000 ch)', '$1', o['value'], o['ms_per_step'], o['launches_per_lane'], o['parity_check']['ok'])" | tee -a $O/se_ab2.txt
  done
done
done
timeout -k 10 500 python -m pytest tests -m gpu -q > $O/third_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/third_pytest.log
tail -6 $O/third_pytest.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 2>$O/third_bench.err | tail -1 > $O/third_bench.json; echo "bench rc=$?"
cut -c1-300 $O/third_bench.json
