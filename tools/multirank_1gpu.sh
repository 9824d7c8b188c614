#!/bin/bash
# The N > 1 code paths on the one-GPU box: (a) one rank under torch.distributed.run with the nccl backend (the collectives of the evidence
# block and the SUM / MAX reductions through RCCL), (b) two ranks sharing the GPU (gloo), for --config c3 and c4.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r6_multirank_1gpu.log; : > $OUT
run() { echo "== $*" >> $OUT; timeout 600 "$@" > /tmp/mr.out 2> /tmp/mr.err; echo "rc=$?" >> $OUT; grep '^{' /tmp/mr.out | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    keep={k:d.get(k) for k in ('metric','value','unit','n_gpus','steps','ms_per_step','scaling')}
    keep['distributed']=d['config'].get('distributed'); keep['sharding']=d['config'].get('sharding'); keep['verified']=d['config'].get('verified', d['config'].get('verified_against_oracle'))
    if d.get('reference_api'): keep['reference_api']={k:d['reference_api'][k] for k in ('pages_per_s','pages','workers')}
    print(json.dumps(keep))
" >> $OUT; tail -3 /tmp/mr.err | cut -c1-300 >> $OUT; }
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --batch 64 --extra-legs 0 --cpu-sample 0
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --config c4 --steps 5 --warmup 2 --api-seconds 2
run python bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 --extra-legs 0 --cpu-sample 0
run python bench.py --gpus 2 --config c4 --steps 5 --warmup 2 --api-seconds 2
cat $OUT | cut -c1-1500
