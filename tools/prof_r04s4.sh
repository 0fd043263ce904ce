# round 4, end of the second session (fp16x3 in both directions).  Full GPU suite, smoke, default bench line + the same command under
# rocprofv3 --kernel-trace --stats, refreshed HBM-traffic passes (stamped), PMC passes over the dense kernels, variant lines.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r04s4_smoke.txt 2>&1; tail -3 $OUT/r04s4_smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > $OUT/r04s4_pytest.txt 2>&1; grep -n "passed\|failed\|FAILED" $OUT/r04s4_pytest.txt | tail -8
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_r04s4_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_r04s4_$c -- python tools/pmc_probe.py > $OUT/pmc_r04s4_$c.log 2>&1
done
python tools/traffic_json.py c3 $OUT/pmc_r04s4_FETCH_SIZE $OUT/pmc_r04s4_WRITE_SIZE
mkdir -p $OUT/profiles_new && cp profiles/hbm_traffic.json profiles/hbm_traffic_pma.json $OUT/profiles_new/
timeout 900 python bench.py > $OUT/r04s4_bench_line.json 2>$OUT/r04s4_bench.err; tail -2 $OUT/r04s4_bench.err
python tools/bench_summary.py $OUT/r04s4_bench_line.json
rm -rf $OUT/prof_r04s4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r04s4 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r04s4_traced_bench_line.json 2>/dev/null
S=$(find $OUT/prof_r04s4 -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/r04s4_bench_kernel_stats.csv; head -7 $OUT/r04s4_bench_kernel_stats.csv | cut -c1-150
find $OUT/prof_r04s4 -name '*kernel_trace.csv' -delete
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf $OUT/pmc_r04s4f_$n
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_r04s4f_$n -- python tools/fused_probe.py > /dev/null 2>$OUT/pmc_r04s4f_$n.err
done
python tools/pmc_sum.py fused_linear_bwd_f16x3 $OUT/pmc_r04s4f_* > $OUT/r04s4_pmc_dense.txt
python tools/pmc_sum.py fused_linear_fwd_roles $OUT/pmc_r04s4f_* >> $OUT/r04s4_pmc_dense.txt
cat $OUT/r04s4_pmc_dense.txt | head -30
find $OUT/pmc_r04s4* -name '*kernel_trace.csv' -delete
timeout 600 python bench.py --model pma --no-cpu-baseline --partitions primary > $OUT/r04s4_pma_bench_line.json 2>/dev/null
timeout 600 python bench.py --norm bn --no-cpu-baseline > $OUT/r04s4_bn_bench_line.json 2>/dev/null
timeout 600 python bench.py --degree-dist poisson --no-cpu-baseline > $OUT/r04s4_poisson_bench_line.json 2>/dev/null
timeout 600 python bench.py --dropout 0 --no-cpu-baseline > $OUT/r04s4_bench_line_dropout0.json 2>/dev/null
python tools/bench_summary.py $OUT/r04s4_pma_bench_line.json $OUT/r04s4_poisson_bench_line.json $OUT/r04s4_bench_line_dropout0.json | grep json
timeout 600 python tools/small_graph_step.py > $OUT/r04s4_small_graph_step.txt 2>&1; tail -12 $OUT/r04s4_small_graph_step.txt
echo finished
