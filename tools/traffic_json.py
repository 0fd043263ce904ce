#!/usr/bin/env python
"""Refresh profiles/hbm_traffic*.json from fresh rocprofv3 --pmc passes and stamp them with the gather kernels' source hash
(bench.kernel_source_sha): bench.py reports ``roofline.traffic`` only while the stamp matches the sources the loaded library
was built from.

usage: traffic_json.py c3 <FETCH_SIZE dir> <WRITE_SIZE dir>       (passes over tools/pmc_probe.py   -> hbm_traffic.json, _pma.json)
       traffic_json.py c5 <FETCH_SIZE dir> <WRITE_SIZE dir>       (passes over tools/pmc_probe_c5.py -> hbm_traffic_c5.json)
Counter values are KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 16 B/lane reads at half; the calibration
launches of pmc_probe.py -- a 1 GiB copy and a no-reuse gather of known size -- are re-checked here and stored)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_dispatch(d, counter):
    """{kernel short name: [value per dispatch, in dispatch order]} for one pass directory."""
    rows = collections.defaultdict(dict)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = row["Kernel_Name"]
                rows[k][int(row["Dispatch_Id"])] = rows[k].get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
    return {k: [v[i] for i in sorted(v)] for k, v in rows.items()}


def pick(table, needle):
    hits = [(k, v) for k, v in table.items() if needle in k]
    assert hits, (needle, list(table)[:20])
    out = []
    for _, v in hits:
        out += v
    return out


def main():
    import bench
    shape, fdir, wdir = sys.argv[1:4]
    F, W = per_dispatch(fdir, "FETCH_SIZE"), per_dispatch(wdir, "WRITE_SIZE")
    sha = bench.kernel_source_sha()
    stamp = {"kernel_source_sha": sha, "taken": os.environ.get("ALLSET_ROUND", "round 6")}
    avg = lambda xs: sum(xs) / len(xs)
    if shape == "c3":
        sf, sw = pick(F, "segreduce_kernel"), pick(W, "segreduce_kernel")
        assert len(sf) == 9 and len(sw) == 9, (len(sf), len(sw))           # 3 no-reuse + 3 V->E + 3 E->V
        known = 16_000_000 * 512 + 16_000_000 * 4 + 1_000_001 * 4
        path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        prof = json.load(open(path))
        read = 2048.0 * avg(sf[3:])
        write = 1024.0 * avg(sw[3:])
        prof.update(stamp)
        prof["calibration"]["round4"] = {"noreuse_raw_FETCH_KiB": sf[:3], "known/(raw*1024)": known / (avg(sf[:3]) * 1024.0),
                                         "noreuse_WRITE_KiB": sw[:3]}
        prof["segreduce_fwd_read_bytes_per_launch"] = read
        prof["segreduce_fwd_write_bytes_per_launch"] = write
        prof["segreduce_fwd_bytes_per_launch"] = read + write
        prof["raw_KiB_c3_launches"] = {"FETCH_SIZE": sf[3:], "WRITE_SIZE": sw[3:]}
        prof["note"] = (f"FETCH_SIZE is taken at the L2's memory-side interface: Infinity-Cache (256 MiB) hits are included, so this is an "
                        f"upper bound on DRAM bytes. measured/algorithmic = {(read + write) / prof['algorithmic_bytes_per_launch']:.3f} "
                        "(average of the 3 V->E and 3 E->V launches of tools/pmc_probe.py)")
        json.dump(prof, open(path, "w"), indent=1)
        print(path, prof["segreduce_fwd_bytes_per_launch"], prof["calibration"]["round4"]["known/(raw*1024)"])
        path = os.path.join(ROOT, "profiles", "hbm_traffic_pma.json")
        prof = json.load(open(path))
        prof.update(stamp)
        for key, needle in (("pma_fwd", "pma_fwd"), ("pma_bwd_stats", "pma_bwd_stats"), ("pma_bwd_src", "pma_bwd_src")):
            f, w = pick(F, needle), pick(W, needle)
            prof["raw_KiB"][needle + "_kernel" if needle != "pma_bwd_stats" else "pma_bwd_stats_flat_kernel"] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
            prof[key + "_bytes_per_launch"] = 2048.0 * avg(f) + 1024.0 * avg(w)
        a = prof["algorithmic_bytes_per_launch"]
        prof["note_round4"] = ("measured / algorithmic: " + ", ".join(f"{k} {prof[k + '_bytes_per_launch'] / a[k]:.3f}" for k in a))
        json.dump(prof, open(path, "w"), indent=1)
        print(path, prof["note_round4"])
    else:
        path = os.path.join(ROOT, "profiles", "hbm_traffic_c5.json")
        prof = json.load(open(path))
        prof.update(stamp)
        for key in ("pma_fwd", "pma_bwd_src"):
            f, w = pick(F, key), pick(W, key)
            prof["raw_KiB_per_launch"][key + "_kernel"] = {"FETCH_SIZE": avg(f), "WRITE_SIZE": avg(w)}
            prof[key + "_bytes_per_launch"] = 2048.0 * avg(f) + 1024.0 * avg(w)
        json.dump(prof, open(path, "w"), indent=1)
        print(path, {k: prof[k + "_bytes_per_launch"] for k in ("pma_fwd", "pma_bwd_src")})


if __name__ == "__main__":
    main()
