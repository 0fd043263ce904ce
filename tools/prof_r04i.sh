cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for W in 4 8; do for M in deepsets pma; do
  ALLSET_DIST_BACKEND=gloo timeout 900 python bench.py --gpus $W --steps 2 --warmup 1 --n-per-gpu 16000 --model $M --chunk-entry 2 --region-timeout 200 > $OUT/r04_w${W}_${M}.json 2> $OUT/r04_w${W}_${M}.err
  echo "world $W $M rc=$?"; python - $OUT/r04_w${W}_${M}.json <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print("NO LINE"); raise SystemExit
j=json.loads(l[0]); p=j.get('partitions',{})
print({k:(round(v['ms_per_step'],2) if 'ms_per_step' in v else v.get('error','?')[:120]) for k,v in p.items() if isinstance(v,dict)}, j['config']['partition'])
PY
  grep -i "error\|Traceback" $OUT/r04_w${W}_${M}.err | head -5
done; done
