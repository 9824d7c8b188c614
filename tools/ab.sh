#!/bin/bash
# A/B of one PMC group between two env settings.  Usage: tools/ab.sh "<counters>" "<envA>" "<envB>"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --batch 16 --steps 2 --warmup 1 --cpu-sample 0 --verify 0 --noise-workers 0 --extra-legs 0"
for v in A B; do
  if [ $v = A ]; then E="$2"; else E="$3"; fi
  rm -rf /tmp/ab_$v
  env $E timeout 200 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /tmp/ab_$v -o ab -- $BENCH > /tmp/ab_$v.log 2>&1
  echo "== $v ($E)"
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/ab_$v/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_chain_fused' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
        acc['ns'].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in acc.items(): print(k, sum(v)/len(v))
PY
done
