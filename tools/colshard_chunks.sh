#!/bin/bash
# bench.py under a 1-rank RCCL group with the collectives forced on: what the sharded layers' machinery costs on ONE GPU
# (every collective is a device-local copy through RCCL).  Prints ms per step per configuration.
export ALLSET_FORCE_COLLECTIVES=1
PORT=29533
run() {
  PORT=$((PORT + 1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 \
    --no-cpu-baseline --partitions primary --steps 10 --warmup 3 "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; l=sys.stdin.readline(); print(round(json.loads(l)['ms_per_step'],2) if l.strip() else 'FAILED')"
}
echo "--shard rows                                   $(run --shard rows)"
echo "--shard columns --pipeline-chunks 1            $(run --shard columns --pipeline-chunks 1)"
echo "--shard columns --pipeline-chunks 2            $(run --shard columns --pipeline-chunks 2)"
echo "--shard columns --pipeline-chunks 4            $(run --shard columns --pipeline-chunks 4)"
echo "--shard rows --model pma                       $(run --shard rows --model pma)"
echo "--shard columns --model pma --pipeline-chunks 1  $(run --shard columns --model pma --pipeline-chunks 1)"
echo "--shard columns --model pma --pipeline-chunks 4  $(run --shard columns --model pma --pipeline-chunks 4)"
C5="--dtype bf16 --feature-dim 256 --model pma --degree-dist zipf --n-per-gpu 250000"
echo "--shard rows    $C5    $(run --shard rows $C5)"
echo "--shard columns $C5 --pipeline-chunks 1    $(run --shard columns $C5 --pipeline-chunks 1)"
echo "--shard columns $C5 --pipeline-chunks 4    $(run --shard columns $C5 --pipeline-chunks 4)"
