cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
ALLSET_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/overlap_trace.py --rows 200000 --chunks 4 > $OUT/r04_overlap_trace.txt 2>$OUT/r04_overlap_trace.err
tail -5 $OUT/r04_overlap_trace.err; cat $OUT/r04_overlap_trace.txt
timeout 900 python -m pytest tests/test_gpu_two_ranks.py -x -q -k "bench" > $OUT/r04d_pytest.txt 2>&1; tail -5 $OUT/r04d_pytest.txt
echo finished
