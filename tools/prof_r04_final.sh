cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r04_final_smoke.txt 2>&1; tail -4 $OUT/r04_final_smoke.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/r04_final_pytest.txt 2>&1; grep -n "passed\|failed" $OUT/r04_final_pytest.txt | tail -2
timeout 900 python bench.py > $OUT/r04_final_bench_line.json 2>$OUT/r04_final_bench.err
python tools/bench_summary.py $OUT/r04_final_bench_line.json
echo finished
