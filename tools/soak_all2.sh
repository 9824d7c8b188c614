#!/bin/bash
# The second long soak of the round: other seeds, the round's last kernels (see tools/soak_all.sh).
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r5i_soaks.txt
: > $out
run() { echo "== $*" >> $out; timeout 500 "$@" 2>/dev/null | tail -1 | cut -c1-900 >> $out; echo "rc=$?" >> $out; }
run python tools/soak.py 240 77001
run python tools/soak4.py 240 77002
run python tools/soak5.py 200 77003
run python tools/soak7.py 240 77004
run python tools/soak8.py 3000
run python tools/soak10.py 120 77005
run python tools/soak3.py 80 77006
run python tools/soak9.py 60 77007
cat $out
