cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_small4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_small4 -- python tools/small_graph_kernels.py cora_ds_add > /dev/null 2>&1
S=$(find $OUT/prof_small4 -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    c=int(r['Calls'])
    if c>=200:
        per=c/200; t=float(r['AverageNs'])*per/1e3; tot+=t
        print(f"{per:5.1f} x {float(r['AverageNs'])/1e3:7.2f} us = {t:7.1f} us/step  {r['Name'][:110]}")
print("total kernel us/step", tot)
PY
find $OUT/prof_small4 -name '*kernel_trace.csv' -delete
