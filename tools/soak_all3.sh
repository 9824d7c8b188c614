#!/bin/bash
# Round 6: the long soak on the final kernels of the round (other seeds; see tools/soak_all.sh).
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r6_soaks.txt
: > $out
run() { echo "== $*" >> $out; timeout 500 "$@" 2>/dev/null | tail -1 | cut -c1-900 >> $out; echo "rc=$?" >> $out; }
run python tools/soak.py 150 88001
run python tools/soak4.py 150 88002
run python tools/soak5.py 120 88003
run python tools/soak7.py 150 88004
run python tools/soak8.py 1200
run python tools/soak10.py 120 88005
run python tools/soak3.py 80 88006
run python tools/soak9.py 60 88007
cat $out
