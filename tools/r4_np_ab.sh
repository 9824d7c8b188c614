#!/bin/bash
# tools/r4_np_ab.sh <lib or env assignments> ... : np_probe timing (16 planes, tile records) for each variant
cd ${GRAFT_REPO_ROOT:-/root/repo}
for V in "$@"; do
  if [[ "$V" == *.so ]]; then E="VKX_LIB=$PWD/vkit_amd/$V"; else E="$V"; fi
  echo "== $V"
  env $E NP_KIND=${NP_KIND:-tiles} VKX_NP_PIPELINE=0 python tools/np_probe.py ${NP_B:-32} timing 2>&1 | grep -E "k_np|exact"
done
