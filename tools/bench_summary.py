import json,sys
for f in sys.argv[1:]:
    l=[x for x in open(f) if x.startswith("{")]
    if not l: print(open(f).read()[-1500:]); continue
    j=json.loads(l[0]); print(f, round(j["ms_per_step"],3), round(j["aggregation"]["ms_per_step"],3), round(j["dense_tail"]["ms_per_step"],3))
    for k,v in ((j.get("roofline") or {}).get("per_kernel") or {}).items(): print("   ", k, round(v["avg_ms"],4), v["calls_per_step"], round(v["frac"],3) if v["frac"] else None)
