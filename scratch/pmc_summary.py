"""Summarise the counter passes of scratch/pmc_bf16.sh: per kernel, mean counter value per launch, clock and MFMA busy fraction."""
import collections, csv, glob, json, os, re, sys
O = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(O, "s*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "End_Timestamp" in r:
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, c in d.items():
    if "mlp" not in k or "pack" in k:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m["launches"] = max(len(v) for v in c.values())
    if dur[k]:
        m["duration_us"] = sum(dur[k]) / len(dur[k]) / 1e3
    if "GRBM_GUI_ACTIVE" in m and "duration_us" in m:
        m["clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8 / (m["duration_us"] * 1e3)
        m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_WAVE_CYCLES" in m:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if n in m:
                m[n + "/WAVE_CYCLES"] = m[n] / m["SQ_WAVE_CYCLES"]
    out[k] = m
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
