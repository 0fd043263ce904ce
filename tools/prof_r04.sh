# round 4 profile batch: bench line, the same command under rocprofv3 --kernel-trace --stats, variant lines, overlap trace
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
P=r04
timeout 900 python bench.py > $OUT/${P}_bench_line.json 2>$OUT/${P}_bench.err
python tools/bench_summary.py $OUT/${P}_bench_line.json
rm -rf $OUT/prof_${P}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${P} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${P}_traced_bench_line.json 2>$OUT/${P}_traced.err
S=$(find $OUT/prof_${P} -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${P}_bench_kernel_stats.csv; head -8 $OUT/${P}_bench_kernel_stats.csv | cut -c1-160
find $OUT/prof_${P} -name '*kernel_trace.csv' -delete
timeout 600 python bench.py --degree-dist poisson --no-cpu-baseline > $OUT/${P}_poisson_bench_line.json 2>/dev/null
timeout 600 python bench.py --degree-dist zipf --no-cpu-baseline > $OUT/${P}_zipf_bench_line.json 2>/dev/null
timeout 600 python bench.py --self-loops --no-cpu-baseline > $OUT/${P}_selfloops_bench_line.json 2>/dev/null
timeout 600 python bench.py --d 256 --no-cpu-baseline > $OUT/${P}_d256_bench_line.json 2>/dev/null
timeout 600 python bench.py --model pma --no-cpu-baseline --partitions primary > $OUT/${P}_pma_bench_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000 --no-cpu-baseline --partitions primary > $OUT/${P}_c5_shape_bench_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000 --no-cpu-baseline --partitions primary --hip-graph > $OUT/${P}_c5_shape_graph_bench_line.json 2>/dev/null
python tools/bench_summary.py $OUT/${P}_poisson_bench_line.json $OUT/${P}_zipf_bench_line.json $OUT/${P}_selfloops_bench_line.json $OUT/${P}_d256_bench_line.json $OUT/${P}_pma_bench_line.json $OUT/${P}_c5_shape_bench_line.json $OUT/${P}_c5_shape_graph_bench_line.json
ALLSET_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/overlap_trace.py --rows 200000 --chunks 4 > $OUT/${P}_overlap_trace.txt 2>/dev/null
tail -3 $OUT/${P}_overlap_trace.txt
timeout 600 python tools/small_graph_step.py > $OUT/${P}_small_graph_step.txt 2>/dev/null; cat $OUT/${P}_small_graph_step.txt
echo finished
