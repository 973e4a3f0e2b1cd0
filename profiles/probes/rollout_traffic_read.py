"""Print the per-launch FETCH_SIZE / WRITE_SIZE of car_rollout_kernel from the two rocprofv3 counter passes of
rollout_traffic_probe.py (usage: python rollout_traffic_read.py <dir with pmc_fetch/ and pmc_write/>)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for sub, counter, corr in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
    f = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "car_rollout_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    out[counter] = [float(r["Counter_Value"]) * 1024.0 * corr / 1e6 for r in rows]
K, T = 65536, 16
res = {"variants": ["states + actions rows", "actions rows only", "no rows"],
       "fetch_MB (x2 gfx950 correction)": out["FETCH_SIZE"], "write_MB": out["WRITE_SIZE"],
       "algorithmic_read_MB": K * (48 + 16 * T) / 1e6,
       "algorithmic_write_MB": [K * (48 * (T + 1) + 16 * T + 56) / 1e6, K * (16 * T + 56) / 1e6, K * 56 / 1e6]}
print(json.dumps(res, indent=1))
