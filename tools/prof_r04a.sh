# round 4, first GPU call: full GPU suite, default bench line, refreshed HBM-traffic PMC passes (stamped)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r04a_pytest.txt 2>&1; tail -5 $OUT/r04a_pytest.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_r04_$c $OUT/pmc_r04c5_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_r04_$c -- python tools/pmc_probe.py > $OUT/pmc_r04_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_r04c5_$c -- python tools/pmc_probe_c5.py > $OUT/pmc_r04c5_$c.log 2>&1
done
python tools/traffic_json.py c3 $OUT/pmc_r04_FETCH_SIZE $OUT/pmc_r04_WRITE_SIZE
python tools/traffic_json.py c5 $OUT/pmc_r04c5_FETCH_SIZE $OUT/pmc_r04c5_WRITE_SIZE
mkdir -p $OUT/profiles_new && cp profiles/hbm_traffic*.json $OUT/profiles_new/
find $OUT/pmc_r04* -name '*kernel_trace.csv' -delete
timeout 900 python bench.py > $OUT/r04a_bench_line.json 2>$OUT/r04a_bench.err; tail -3 $OUT/r04a_bench.err
python tools/bench_summary.py $OUT/r04a_bench_line.json
echo finished
