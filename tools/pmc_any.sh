#!/bin/bash
# PMC passes (instruction mix, activity, HBM bytes) of any command: tools/pmc_any.sh <tag> <kernel-name substring> <command...>
# Each --pmc group is its own pass with --kernel-trace only.  Prints per-dispatch means for the kernels that match.
TAG=$1; KEY=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
CMD="$*"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd $ROOT && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1)
  echo "pass $i rc=$?"
done
python - "$OUT" "$KEY" <<'PY'
import csv, glob, os, re, sys, collections
out, key = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(path)):
        if key in r['Kernel_Name']:
            m_ = re.search(r'\bk_[A-Za-z0-9_]+', r['Kernel_Name'])
            short = m_.group(0) if m_ else r['Kernel_Name'].split('(')[0][-60:]
            acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
            acc[short]['ns'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for name, d in acc.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    w = m.get('SQ_WAVES', 1)
    print(f'== {name} (dispatches {len(d["ns"])}, waves {w:.0f}, {m["ns"] / 1e3:.1f} us under the profiler)')
    for k in sorted(m):
        print(f'  {k:24s} {m[k]:14.6g}  per wave {m[k] / w:10.2f}')
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        print(f'  HBM bytes per dispatch: fetch {m["FETCH_SIZE"] * 1024 * 2 / 1e6:.1f} MB (x2 gfx950) + write {m["WRITE_SIZE"] * 1024 / 1e6:.1f} MB')
PY
