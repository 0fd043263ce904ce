cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/r04f_pytest.txt 2>&1; tail -4 $OUT/r04f_pytest.txt
timeout 600 python bench.py --norm bn --no-cpu-baseline > $OUT/r04_bn_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/r04_bn_bench_line.json
echo finished
