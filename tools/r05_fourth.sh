#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests -m gpu -q > $O/fourth_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/fourth_pytest.log
tail -6 $O/fourth_pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'], o['stages']['decode'], o['parity_check']['ok'])"
