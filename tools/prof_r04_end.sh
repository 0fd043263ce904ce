# end-of-round verification: smoke(), the full GPU suite, the default bench line + the same command under rocprofv3 --kernel-trace --stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r04_end_smoke.txt 2>&1; tail -5 $OUT/r04_end_smoke.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/r04_end_pytest.txt 2>&1; grep -n "passed\|failed" $OUT/r04_end_pytest.txt | tail -2
timeout 900 python bench.py > $OUT/r04_end_bench_line.json 2>$OUT/r04_end_bench.err
python tools/bench_summary.py $OUT/r04_end_bench_line.json
rm -rf $OUT/prof_r04_end
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r04_end -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r04_end_traced_bench_line.json 2>/dev/null
S=$(find $OUT/prof_r04_end -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/r04_end_bench_kernel_stats.csv; head -7 $OUT/r04_end_bench_kernel_stats.csv | cut -c1-150
find $OUT/prof_r04_end -name '*kernel_trace.csv' -delete
timeout 600 python bench.py --model pma --no-cpu-baseline --partitions primary > $OUT/r04_end_pma_bench_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000 --no-cpu-baseline --partitions primary > $OUT/r04_end_c5_shape_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/r04_end_pma_bench_line.json $OUT/r04_end_c5_shape_bench_line.json | grep json
echo finished
