#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d.get('value'), 'ms', d.get('ms_per_step'), 'lanes', d.get('config',{}).get('batch_lanes'), d.get('config',{}).get('batch_lanes_rule'), 'passes', d.get('config',{}).get('passes', d.get('passes')))
print('   parity_check', json.dumps(d.get('parity_check')))
print('   roofline', json.dumps({k: v for k, v in (d.get('roofline') or {}).items() if k in ('bound','frac','achieved','kernel','traffic')})[:300])
PY
}
timeout -k 5 400 python bench.py --steps 20 --warmup 5 > $O/bc_yolo.json 2>$O/bc_yolo.err || { echo "yolo rc $?"; tail -5 $O/bc_yolo.err; }
show $O/bc_yolo.json
timeout -k 5 400 python bench.py --steps 20 --warmup 5 --config efficientdet-d1 > $O/bc_d1.json 2>$O/bc_d1.err || { echo "d1 rc $?"; tail -5 $O/bc_d1.err; }
show $O/bc_d1.json
timeout -k 5 400 python bench.py --steps 20 --warmup 5 --config d1_fcs2_atss > $O/bc_fcos.json 2>$O/bc_fcos.err || { echo "fcos rc $?"; tail -5 $O/bc_fcos.err; }
show $O/bc_fcos.json
timeout -k 5 400 python bench.py --steps 10 --warmup 8 --profile > $O/bc_prof.json 2>$O/bc_prof.err || { echo "prof rc $?"; tail -5 $O/bc_prof.err; }
cat $O/bc_prof.json
timeout -k 5 400 python bench.py --steps 10 --warmup 3 --eager --no-cpu-baseline > $O/bc_eager.json 2>$O/bc_eager.err || { echo "eager rc $?"; tail -5 $O/bc_eager.err; }
show $O/bc_eager.json
timeout -k 5 400 python -m pytest tests/test_gpu_parallel.py -q -x 2>&1 | tail -3
