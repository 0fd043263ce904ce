cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
ALLSET_HYPOTHESIS_RANDOM=1 ALLSET_HYPOTHESIS_EXAMPLES=${EX:-150} timeout 3000 python -m pytest tests/test_gpu_random_shapes.py tests/test_gpu_dist_random.py -q -x > $OUT/r04_sweep_${EX:-150}.txt 2>&1; tail -15 $OUT/r04_sweep_${EX:-150}.txt
