set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py > $OUT/r03_final_bench_line.json 2>$OUT/r03_bench_line.err
python tools/bench_summary.py $OUT/r03_final_bench_line.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r03f -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03_final_traced_bench_line.json 2>$OUT/r03_traced.err
find $OUT/prof_r03f -name '*kernel_stats.csv' | head
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_r03f_$n -- python tools/fused_probe.py > /dev/null 2>$OUT/pmc_r03f_$n.err
done
python tools/pmc_sum.py fused_linear_bwd_roles $OUT/pmc_r03f_* > $OUT/r03f_pmc_bwd_roles.txt
python tools/pmc_sum.py fused_linear_fwd_roles $OUT/pmc_r03f_* > $OUT/r03f_pmc_fwd.txt
cat $OUT/r03f_pmc_bwd_roles.txt $OUT/r03f_pmc_fwd.txt
# keep only the small csvs
find $OUT/pmc_r03f_* -name '*kernel_trace.csv' -delete
timeout 600 python bench.py --model pma --no-cpu-baseline --partitions primary > $OUT/r03_final_pma_bench_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000 --no-cpu-baseline --partitions primary > $OUT/r03_final_c5_shape_bench_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000 --no-cpu-baseline --partitions primary --hip-graph > $OUT/r03_final_c5_shape_graph_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/r03_final_pma_bench_line.json $OUT/r03_final_c5_shape_bench_line.json $OUT/r03_final_c5_shape_graph_bench_line.json
echo finished
