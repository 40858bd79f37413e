#!/usr/bin/env python3
"""Per-kernel summary of a profiles/pmc_sq.sh pass: averages per launch and the derived ratios."""
import csv, glob, json, os, sys, collections
src = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for f in glob.glob(os.path.join(src, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void tg::", "").replace("tg::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            calls[k] += 1
out = {}
for k, t in sorted(tot.items()):
    if not k.startswith("k_"):
        continue
    n = max(1, calls[k])
    wc = max(1.0, t["SQ_WAVE_CYCLES"])
    out[k] = {"launches": calls[k], "wave_cycles_per_launch": t["SQ_WAVE_CYCLES"]/n, "valu_insts_per_launch": t["SQ_INSTS_VALU"]/n,
              "vmem_rd_insts_per_launch": t["SQ_INSTS_VMEM_RD"]/n,
              "wait_any_share": t["SQ_WAIT_ANY"]/wc, "wait_inst_share": t["SQ_WAIT_INST_ANY"]/wc, "active_inst_share": t["SQ_ACTIVE_INST_ANY"]/wc,
              "valu_issue_share": t["SQ_ACTIVE_INST_VALU"]/wc,
              "lane_utilisation": t["SQ_THREAD_CYCLES_VALU"]/max(1.0, 64.0*t["SQ_ACTIVE_INST_VALU"])}
json.dump(out, open(os.path.join(src, "sq_summary.json"), "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(f"{k:40s} n={v['launches']:4d} lanes {v['lane_utilisation']:.3f} wait {v['wait_any_share']:.3f} valu-issue {v['valu_issue_share']:.3f} "
          f"active {v['active_inst_share']:.3f} VALU insts/launch {v['valu_insts_per_launch']:.3e} VMEM rd {v['vmem_rd_insts_per_launch']:.3e}")
