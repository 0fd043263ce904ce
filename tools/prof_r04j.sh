cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py --no-cpu-baseline > $OUT/r04j_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/r04j_bench_line.json
timeout 600 python bench.py --no-cpu-baseline --dropout 0.2 > $OUT/r04j_bench_p02.json 2>/dev/null
python tools/bench_summary.py $OUT/r04j_bench_p02.json
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/r04j_pytest.txt 2>&1; tail -4 $OUT/r04j_pytest.txt
echo finished
