#!/usr/bin/env python3
"""Turns gpurun_out/prof (written by profiles/collect.sh on the GPU box) into the committed summaries:
   profiles/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc.json and profiles/pmc_extend.json
   (per-launch HBM-side traffic of k_extend, read by bench.py's roofline.traffic).
   Counter handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB and come from the L2's
   fabric-side request counters; on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of wide (16 B/lane) loads.
   The other access widths of these kernels were calibrated on known byte counts (profiles/micro/fetch_calib.hip ->
   profiles/r02_fetch_calib.txt): every L2 miss is one fabric request = one 128-byte line fill, tallied at 64 B, whatever
   the width of the access that caused it -> the read side is doubled for every kernel."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "gpurun_out", "prof")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(here, f"{tag}_bench.json"))
def newest(pattern):
    """gpurun merges a call's files INTO gpurun_out/, so earlier collections' files (other process ids) stay next to the new ones:
    only the most recent file of a kind belongs to this collection."""
    return max(glob.glob(pattern), key=os.path.getmtime)

stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(here, f"{tag}_kernel_stats.csv"))
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for f in [newest(os.path.join(d, "*", "*counter_collection.csv")) for d in sorted(glob.glob(os.path.join(src, "pmc_*"))) if os.path.isdir(d)]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void tg::", "").replace("tg::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE":
            calls[k] += 1
out = {}
for k, t in tot.items():
    n = max(1, calls[k])
    fetch = t.get("FETCH_SIZE", 0.0)*1024*2   # gfx950 correction for 16-byte-per-lane loads
    write = t.get("WRITE_SIZE", 0.0)*1024
    hit, miss = t.get("TCC_HIT_sum", 0.0), t.get("TCC_MISS_sum", 0.0)
    out[k] = {"launches": calls[k], "fetch_bytes_per_launch": fetch/n, "write_bytes_per_launch": write/n,
              "hbm_bytes_per_launch": (fetch + write)/n, "l2_hit_rate": hit/max(1.0, hit + miss)}
json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
# what bench.py's roofline.traffic reads (per kernel, per launch); `_source` is quoted next to the number
latest = dict(out)
latest["_source"] = f"profiles/{tag}_pmc.json"
import hashlib
csrc = os.path.join(os.path.dirname(here), "tungsten_amd", "csrc")
latest["_kernels_sha"] = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in ("kernels.h", "device.h"))).hexdigest()[:16]  # bench.py: is this pass still about the kernels it prices?
json.dump(latest, open(os.path.join(here, "pmc_latest.json"), "w"), indent=1, sort_keys=True)
ext = next((v for k, v in out.items() if k.startswith("k_extend_p<false") or k.startswith("k_extend_q<false")), None)
if ext:
    json.dump({"kernel": "k_extend_p<false> (k_extend_q<false, 4> in profiles taken before it existed)", "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
               "hbm_bytes_per_launch": round(ext["hbm_bytes_per_launch"]), "fetch_bytes_per_launch": round(ext["fetch_bytes_per_launch"]),
               "write_bytes_per_launch": round(ext["write_bytes_per_launch"]), "l2_hit_rate": round(ext["l2_hit_rate"], 4),
               "note": "FETCH_SIZE doubled per the gfx950 16-B/lane correction; includes Infinity-Cache hits"},
              open(os.path.join(here, "pmc_extend.json"), "w"), indent=1)
print(json.dumps(ext or {}, indent=1))
