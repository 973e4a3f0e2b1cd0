"""Turn the rocprofv3 outputs of profiles/run_profiles.sh (rocpd sqlite or csv) into the committed summaries:
  <tag>_bench_kernel_stats.csv   per-kernel calls / total / average duration of the bench command
  <tag>_pmc_traffic.json         per-kernel HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes
  <tag>_pmc_sq.json              per-kernel SQ counters per launch (MFMA busy, LDS bank conflicts, wave / wait cycles)
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, so reads are
doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B-per-lane stores.
usage: python profiles/summarize.py gpurun_out/prof_r01d r01"""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys


def kernel_rows(path):
    db = glob.glob(os.path.join(path, "*.db"))
    if db:
        c = sqlite3.connect(db[0])
        return [(n, e - s) for n, s, e in c.execute("select name, start, end from kernels")]
    f = glob.glob(os.path.join(path, "*kernel_trace.csv"))[0]
    return [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f))]


def counter_rows(path, counter):
    db = glob.glob(os.path.join(path, "*.db"))
    if db:
        c = sqlite3.connect(db[0])
        return list(c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)))
    f = glob.glob(os.path.join(path, "*counter_collection.csv"))[0]
    return [(r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]


def short(name):
    for k in ("conv3_halo16x3_kernel", "conv3_halo16_kernel", "gemm16_kernel", "conv2d_small_kernel", "conv_gemm_kernel", "gn2d_kernel", "gn1d_kernel",
              "final_proj_flow_kernel", "car_rollout_kernel", "lidar_scan_kernel", "nn_argmin_kernel", "local_map_kernel",
              "im2col2d_kernel", "maxpool2d_kernel", "encoder_stem_kernel", "mppi_rollout_kernel", "mppi_partial_kernel",
              "mppi_ant_rollout_kernel", "mppi_ant_partial_kernel", "mppi_ant_finish_kernel", "mppi_ant_min_kernel", "mppi_finish_kernel", "mppi_min_kernel", "gn1d_short_kernel", "cond_vector_ant_kernel", "ant_rollout_kernel",
              "ant_collision_kernel", "ant_gather_hist_kernel", "ant_copy_actions_kernel", "accept_commit_kernel", "accept_scan_kernel"):
        if k in name:
            return k
    return name.split("(")[0][:60]


def main(root, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    agg = collections.defaultdict(list)
    for n, d in kernel_rows(os.path.join(root, "trace")):
        agg[n].append(d)
    total = sum(sum(v) for v in agg.values())
    with open(os.path.join(here, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), f"{sum(v) / len(v):.1f}", f"{100.0 * sum(v) / total:.2f}", min(v), max(v)])
    traffic = {}
    for sub, counter, corr in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
        per = collections.defaultdict(list)
        for n, v in counter_rows(os.path.join(root, sub), counter):
            per[short(n)].append(v * 1024.0 * corr)
        for n, v in per.items():
            traffic.setdefault(n, {"launches": len(v)})[counter.lower() + "_bytes_per_launch"] = sum(v) / len(v)
    for n, t in traffic.items():
        t["hbm_bytes_per_launch"] = t.get("fetch_size_bytes_per_launch", 0.0) + t.get("write_size_bytes_per_launch", 0.0)
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) of "
                     "`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-early-exit-line --no-profile`",
           "corrections": "KiB -> bytes; FETCH_SIZE x 2 on gfx950 (128-B requests tallied as 64 B); WRITE_SIZE as is",
           "kernels": dict(sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}
    with open(os.path.join(here, f"{tag}_pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    for n in list(out["kernels"])[:6]:
        print(n, {k: round(v / 1e6, 2) if "bytes" in k else v for k, v in out["kernels"][n].items()})
    # SQ pass: per kernel and launch, every counter the pass collected
    rows = []
    for sub in ("pmc_sq", "pmc_sq2"):
        sqdir = os.path.join(root, sub)
        f = glob.glob(os.path.join(sqdir, "*counter_collection.csv"))
        if f:
            rows += [(r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f[0]))]
        elif glob.glob(os.path.join(sqdir, "*.db")):
            c = sqlite3.connect(glob.glob(os.path.join(sqdir, "*.db"))[0])
            rows += list(c.execute("select kernel_name, counter_name, value from counters_collection"))
    if rows:
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for n, cn, v in rows:
            per[short(n)][cn].append(v)
        sq = {}
        for n, d in per.items():
            e = {cn: sum(v) / len(v) for cn, v in d.items()}
            e["launches"] = max(len(v) for v in d.values())
            if e.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["mfma_busy_over_sq_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"]
            if e.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in e:
                e["lds_conflict_share"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
            sq[n] = e
        with open(os.path.join(here, f"{tag}_pmc_sq.json"), "w") as fo:
            json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_* (own pass); averages per launch, summed over the chip as rocprofv3 reports them",
                       "kernels": dict(sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1]["launches"]))}, fo, indent=1)
        for n in list(sq)[:4]:
            print("SQ", n, {k: (round(v, 3) if isinstance(v, float) and v < 10 else v) for k, v in sq[n].items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
