#!/bin/bash
# how often does the full-size D1 determinism check fail, and under which switches?
run() { tag=$1; shift; ok=0; bad=0; for i in 1 2 3 4 5 6; do env "$@" timeout -k 10 120 python -m pytest tests/test_gpu_model.py -x -q -k "test_effdet_full_size_properties_640 and efficientdet" > /tmp/fl.log 2>&1 && ok=$((ok+1)) || { bad=$((bad+1)); grep -n "^E  \|test_gpu_model.py:[0-9]*: " /tmp/fl.log | head -3; }; done; echo "$tag: ok $ok bad $bad"; }
run "HEAD" X=1
run "no split-bf16 expand convs" MYDET_B3_EXPAND_MIN_ROWS=0
run "gate as its own launch" MYDET_SE_IN_DW=0
