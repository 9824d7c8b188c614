#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r5e_soaks.txt
: > $out
run() { echo "== $*" >> $out; timeout 400 "$@" 2>/dev/null | tail -1 >> $out; echo "rc=$?" >> $out; }
run python tools/soak3.py 100 9001
run python tools/soak4.py 150 9002
run python tools/soak5.py 150 9003
run python tools/soak7.py 150 9004
run python tools/soak8.py 1500
run python tools/soak9.py 60 9006
run python tools/soak10.py 100 9007
VKX_RGB_RUN=8 run python tools/soak10.py 60 9008
run python tools/soak.py 120 9009
run python tools/soak2.py 100 9010
cat $out
