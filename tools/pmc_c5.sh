cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_c5_$c -- python tools/pmc_probe_c5.py > $OUT/pmc_c5_$c.log 2>&1
done
tail -1 $OUT/pmc_c5_FETCH_SIZE.log
python tools/pmc_sum.py pma_fwd $OUT/pmc_c5_FETCH_SIZE $OUT/pmc_c5_WRITE_SIZE
python tools/pmc_sum.py pma_bwd_src $OUT/pmc_c5_FETCH_SIZE $OUT/pmc_c5_WRITE_SIZE
find $OUT/pmc_c5_* -name '*kernel_trace.csv' -delete
