TAG=r06x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/finx
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; tail -2 $OUT/${TAG}_smoke.txt
rm -rf $OUT/audit
ALLSET_ABI_TRACE=$OUT/abi_trace.json timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/audit -- python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider > $OUT/${TAG}_pytest.txt 2>&1; grep -n "passed\|failed\|FAILED" $OUT/${TAG}_pytest.txt | tail -8
find $OUT/audit -name '*kernel_trace.csv' -delete; find $OUT/audit -name '*agent_info.csv' -delete; find $OUT/audit -name '*domain_stats.csv' -delete
python tools/kernel_audit.py report $OUT/abi_trace.json $OUT/audit > $OUT/${TAG}_kernel_audit.md 2>&1; grep -n "never launched\.$" $OUT/${TAG}_kernel_audit.md
rm -rf $OUT/audit $OUT/abi_trace.json
timeout 900 python bench.py > $OUT/${TAG}_bench_line.json 2>$OUT/${TAG}_bench.err; tail -2 $OUT/${TAG}_bench.err
python tools/bench_summary.py $OUT/${TAG}_bench_line.json
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_traced_bench_line.json 2>/dev/null
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${TAG}_bench_kernel_stats.csv; head -4 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof
timeout 300 python tools/tuned_config_step.py > $OUT/${TAG}_tuned_config_step.txt 2>&1; grep "non-zeros" $OUT/${TAG}_tuned_config_step.txt
timeout 600 python bench.py --model pma --no-cpu-baseline --partitions primary > $OUT/${TAG}_pma_bench_line.json 2>/dev/null; python tools/bench_summary.py $OUT/${TAG}_pma_bench_line.json | head -3
