#!/bin/bash
cd $GRAFT_REPO_ROOT
for pf in 2 15 16; do echo "== PF $pf"; MYDET_B3_PF=$pf timeout -k 10 200 python tools/r05_b3.py 2>&1 | grep "^32x" | sed 's/|b3 - f32| [^ ]* (max|y| [^)]*) //'; done
