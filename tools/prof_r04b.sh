set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/mall_block_probe.py > $OUT/r04_mall_block_probe.txt 2>&1; cat $OUT/r04_mall_block_probe.txt
timeout 600 python bench.py --dropout 0.0 --no-cpu-baseline > $OUT/r04b_bench_nodrop.json 2>/dev/null
python tools/bench_summary.py $OUT/r04b_bench_nodrop.json
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_bench_cli.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5
echo finished
