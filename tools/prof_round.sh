# One full GPU batch for a round: usage  gpurun -- "TAG=r05 bash tools/prof_round.sh".  (The round-4 batches prof_r04*.sh were this script with the tag spelled out.)  Full GPU suite, smoke, default bench line + the same command under
# rocprofv3 --kernel-trace --stats, refreshed HBM-traffic passes (stamped), PMC passes over the dense kernels, variant lines.
TAG=${TAG:-r06}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; tail -3 $OUT/${TAG}_smoke.txt
# the whole suite under the kernel / entry-point audit (tools/kernel_audit.py): rocprofv3 --kernel-trace --stats + ALLSET_ABI_TRACE
rm -rf $OUT/audit
ALLSET_ABI_TRACE=$OUT/abi_trace.json timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/audit -- python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider > $OUT/${TAG}_pytest.txt 2>&1; grep -n "passed\|failed\|FAILED" $OUT/${TAG}_pytest.txt | tail -8
find $OUT/audit -name '*kernel_trace.csv' -delete; find $OUT/audit -name '*agent_info.csv' -delete; find $OUT/audit -name '*domain_stats.csv' -delete
python tools/kernel_audit.py report $OUT/abi_trace.json $OUT/audit > $OUT/${TAG}_kernel_audit.md 2>&1; grep -n "never launched\.$" $OUT/${TAG}_kernel_audit.md
# the C ABI compiled ON the box (ROCm 7.0 runtime, hipcc 7.2): a touched source sends tests/test_gpu_c_example.py down its compile path
hipcc --version > $OUT/${TAG}_box_hipcc.txt 2>&1; touch examples/abi_example.cpp; timeout 900 python -m pytest tests/test_gpu_c_example.py -q -m gpu -p no:cacheprovider >> $OUT/${TAG}_box_hipcc.txt 2>&1; tail -2 $OUT/${TAG}_box_hipcc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_${TAG}_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${TAG}_$c -- python tools/pmc_probe.py > $OUT/pmc_${TAG}_$c.log 2>&1
done
python tools/traffic_json.py c3 $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE
mkdir -p $OUT/profiles_new && cp profiles/hbm_traffic.json profiles/hbm_traffic_pma.json $OUT/profiles_new/
timeout 900 python bench.py > $OUT/${TAG}_bench_line.json 2>$OUT/${TAG}_bench.err; tail -2 $OUT/${TAG}_bench.err
python tools/bench_summary.py $OUT/${TAG}_bench_line.json
rm -rf $OUT/prof_${TAG}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_traced_bench_line.json 2>/dev/null
S=$(find $OUT/prof_${TAG} -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${TAG}_bench_kernel_stats.csv; head -7 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-150
find $OUT/prof_${TAG} -name '*kernel_trace.csv' -delete
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf $OUT/pmc_${TAG}f_$n
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${TAG}f_$n -- python tools/fused_probe.py > /dev/null 2>$OUT/pmc_${TAG}f_$n.err
done
python tools/pmc_sum.py fused_linear_bwd_f16x3 $OUT/pmc_${TAG}f_* > $OUT/${TAG}_pmc_dense.txt
python tools/pmc_sum.py fused_linear_fwd_roles $OUT/pmc_${TAG}f_* >> $OUT/${TAG}_pmc_dense.txt
cat $OUT/${TAG}_pmc_dense.txt | head -30
find $OUT/pmc_${TAG}* -name '*kernel_trace.csv' -delete
timeout 600 python bench.py --model pma --no-cpu-baseline --partitions primary > $OUT/${TAG}_pma_bench_line.json 2>/dev/null
timeout 600 python bench.py --norm bn --no-cpu-baseline > $OUT/${TAG}_bn_bench_line.json 2>/dev/null
timeout 600 python bench.py --degree-dist poisson --no-cpu-baseline > $OUT/${TAG}_poisson_bench_line.json 2>/dev/null
timeout 600 python bench.py --dropout 0 --no-cpu-baseline > $OUT/${TAG}_bench_line_dropout0.json 2>/dev/null
python tools/bench_summary.py $OUT/${TAG}_pma_bench_line.json $OUT/${TAG}_poisson_bench_line.json $OUT/${TAG}_bench_line_dropout0.json | grep json
timeout 600 python bench.py --d 256 --no-cpu-baseline > $OUT/${TAG}_d256_bench_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000 --no-cpu-baseline --partitions primary > $OUT/${TAG}_c5_shape_bench_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000 --no-cpu-baseline --partitions primary --hip-graph > $OUT/${TAG}_c5_shape_graph_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/${TAG}_d256_bench_line.json $OUT/${TAG}_c5_shape_bench_line.json $OUT/${TAG}_c5_shape_graph_bench_line.json | grep -v "^    "
timeout 600 python bench.py --d 512 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_d512_bench_line.json 2>/dev/null
# configs[4] per-GPU shape: the rocprofv3 summary of the same command, the gather kernels' HBM-traffic passes (stamped), the bf16 Linear's ablation arms
rm -rf $OUT/prof_${TAG}_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_c5 -- python bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000 --steps 20 --warmup 5 --no-cpu-baseline --partitions primary > $OUT/${TAG}_c5_traced_bench_line.json 2>/dev/null
S=$(find $OUT/prof_${TAG}_c5 -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${TAG}_c5_kernel_stats.csv; head -14 $OUT/${TAG}_c5_kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof_${TAG}_c5
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_c5_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_c5_$c -- python tools/pmc_probe_c5.py > $OUT/pmc_c5_$c.log 2>&1
done
python tools/traffic_json.py c5 $OUT/pmc_c5_FETCH_SIZE $OUT/pmc_c5_WRITE_SIZE; cp profiles/hbm_traffic_c5.json $OUT/profiles_new/
find $OUT/pmc_c5_* -name '*kernel_trace.csv' -delete; find $OUT/pmc_c5_* -name '*counter_collection.csv' -delete
(python tools/linear_bf16_ablation.py; python tools/linear_bf16_ablation.py --warm) > $OUT/${TAG}_bf16_linear_ablation.txt 2>&1; grep -c "us" $OUT/${TAG}_bf16_linear_ablation.txt
timeout 600 python tools/preprocess_bench.py > $OUT/${TAG}_preprocess_bench.txt 2>&1; tail -3 $OUT/${TAG}_preprocess_bench.txt
timeout 600 python tools/small_graph_step.py > $OUT/${TAG}_small_graph_step.txt 2>&1; tail -12 $OUT/${TAG}_small_graph_step.txt
# dataset scale: kernel inventory of the replayed training steps (launches and kernel time per step by name)
for c in cora_ds_add citeseer_pma_h4; do
  rm -rf $OUT/prof_sg_$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sg_$c -- python tools/small_graph_kernels.py $c > /dev/null 2>&1
  S=$(find $OUT/prof_sg_$c -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${TAG}_small_graph_${c}_kernel_stats.csv; rm -rf $OUT/prof_sg_$c
done
rm -rf $OUT/prof_sg_t; MODEL_ARGS="MLP_hidden=512,heads=8" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sg_t -- python tools/small_graph_kernels.py citeseer_pma_h4 > /dev/null 2>&1
S=$(find $OUT/prof_sg_t -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${TAG}_small_graph_citeseer_512x8_kernel_stats.csv; rm -rf $OUT/prof_sg_t
timeout 600 python tools/tuned_config_step.py > $OUT/${TAG}_tuned_config_step.txt 2>&1; grep -v amdgpu $OUT/${TAG}_tuned_config_step.txt | tail -6
timeout 300 python tools/graph_replay_probe.py >> $OUT/${TAG}_small_graph_step.txt 2>&1
timeout 600 python bench.py --hip-graph --no-cpu-baseline > $OUT/${TAG}_graph_bench_line.json 2>/dev/null
timeout 600 python bench.py --model pma --hip-graph --no-cpu-baseline --partitions primary > $OUT/${TAG}_pma_graph_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/${TAG}_graph_bench_line.json $OUT/${TAG}_pma_graph_bench_line.json | grep -v "^    "
echo finished
