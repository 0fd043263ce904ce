# Re-take the stamped HBM-traffic profiles (profiles/hbm_traffic*.json) after a change of csrc/segreduce.hip, pma.hip or common.h:
# usage  gpurun -- "bash tools/pmc_traffic_batch.sh"; copies land in gpurun_out/profiles_new/ (copy them into profiles/).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/profiles_new
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_t_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_t_$c -- python tools/pmc_probe.py > $OUT/pmc_t_$c.log 2>&1
done
python tools/traffic_json.py c3 $OUT/pmc_t_FETCH_SIZE $OUT/pmc_t_WRITE_SIZE
cp profiles/hbm_traffic.json profiles/hbm_traffic_pma.json $OUT/profiles_new/
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_c5_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_c5_$c -- python tools/pmc_probe_c5.py > $OUT/pmc_c5_$c.log 2>&1
done
python tools/traffic_json.py c5 $OUT/pmc_c5_FETCH_SIZE $OUT/pmc_c5_WRITE_SIZE; cp profiles/hbm_traffic_c5.json $OUT/profiles_new/
rm -rf $OUT/pmc_t_FETCH_SIZE $OUT/pmc_t_WRITE_SIZE $OUT/pmc_c5_FETCH_SIZE $OUT/pmc_c5_WRITE_SIZE
python - <<'P'
import json
for f in ("hbm_traffic.json", "hbm_traffic_pma.json", "hbm_traffic_c5.json"):
    d = json.load(open("profiles/" + f)); print(f, d.get("kernel_source_sha"), {k: v for k, v in d.items() if k.endswith("per_launch")})
P
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']; print('traffic', r['traffic'], r['traffic_source'][:80])"
