cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_batchnorm.py tests/test_gpu_two_ranks.py -x -q > $OUT/r04e_pytest.txt 2>&1; tail -6 $OUT/r04e_pytest.txt
timeout 600 python bench.py --norm bn --no-cpu-baseline > $OUT/r04_bn_bench_line.json 2>$OUT/r04_bn.err; tail -2 $OUT/r04_bn.err
python tools/bench_summary.py $OUT/r04_bn_bench_line.json
ALLSET_BN_TORCH=1 timeout 600 python bench.py --norm bn --no-cpu-baseline > $OUT/r04_bn_torch_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/r04_bn_torch_bench_line.json
timeout 600 python bench.py --no-cpu-baseline > $OUT/r04e_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/r04e_bench_line.json
echo finished
