cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > $OUT/r04k_a$i.json 2>/dev/null; python tools/bench_summary.py $OUT/r04k_a$i.json | head -1
done
ALLSET_EXTRA_CXXFLAGS=-DALLSET_ABL_DROP16 python -m allset_amd.build --force > /dev/null 2>&1
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > $OUT/r04k_b$i.json 2>/dev/null; python tools/bench_summary.py $OUT/r04k_b$i.json | head -1
done
python tools/bench_summary.py $OUT/r04k_a2.json $OUT/r04k_b2.json
