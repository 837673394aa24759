#!/usr/bin/env python3
"""usage: python scripts/summarize_profiles.py <tag>
Copies the rocprofv3 --stats summary of gpurun_out/prof_<tag> into profiles/<tag>_kernel_stats.csv and condenses the
two PMC passes into profiles/<tag>_pmc_hbm_traffic.csv / .json (KB per launch raw, and bytes per launch with the
gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE x2 for the coalesced streaming kernels, WRITE_SIZE as is)."""
import collections
import csv
import glob
import json
import shutil
import sys

tag = sys.argv[1]
STREAMING = ("k_part_scatter", "k_part_hist", "k_build_fused", "k_build_lean", "k_dedup", "k_chunk_place", "k_chunk_count", "k_gather_new", "k_compact_hits",
             "k_shard", "k_flag_tile_sums", "k_occupied_flags")
ks = glob.glob("gpurun_out/prof_%s/*/*_kernel_stats.csv" % tag)
if ks:
    shutil.copy(ks[0], "profiles/%s_kernel_stats.csv" % tag)
kp = glob.glob("gpurun_out/prof_%s_phases/*/*_kernel_stats.csv" % tag)      # bench.py WITH its informational legs (erase, count, W1, LP ...)
if kp:
    shutil.copy(kp[0], "profiles/%s_phases_kernel_stats.csv" % tag)
res = {}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    fs = glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (name, tag))
    if not fs:
        continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        e = res.setdefault(k, {})
        e[ctr + "_launches"] = n
        e[ctr + "_KB_per_launch_raw"] = v / n
for k, e in res.items():
    f = e.get("FETCH_SIZE_KB_per_launch_raw", 0.0) * 1024
    w = e.get("WRITE_SIZE_KB_per_launch_raw", 0.0) * 1024
    streaming = k.startswith(STREAMING)
    e["read_bytes_per_launch"] = f * (2 if streaming else 1)
    e["read_correction"] = "x2 (coalesced streaming read, gfx950)" if streaming else "raw (random access, uncalibrated)"
    e["write_bytes_per_launch"] = w
    e["hbm_bytes_per_launch"] = e["read_bytes_per_launch"] + w
# HBM bytes of one insert batch = sum over the kernels of the insert path (everything but the query kernels), per bench step
# (the PMC passes run bench.py --steps 1 --warmup 0: launches = launches per batch)
INSERT_PATH = ("k_part_", "k_build_fused", "k_build_lean", "k_fused_", "k_make_tiles", "k_scan", "k_init_cursors", "k_sample_dups", "k_seg_offsets", "k_dedup", "k_chunk_")
ins = sum(e["hbm_bytes_per_launch"] * max(e.get("FETCH_SIZE_launches", 0), e.get("WRITE_SIZE_launches", 0)) for k, e in res.items() if k.startswith(INSERT_PATH))
res["_insert_path"] = {"hbm_bytes_per_batch": ins, "kernels": sorted(k for k in res if k.startswith(INSERT_PATH))}
json.dump(res, open("profiles/%s_pmc_hbm_traffic.json" % tag, "w"), indent=1, sort_keys=True)
with open("profiles/%s_pmc_hbm_traffic.csv" % tag, "w") as f:
    f.write("kernel,launches,FETCH_KB_raw_per_launch,WRITE_KB_raw_per_launch,read_bytes_corrected,write_bytes,hbm_bytes_per_launch\n")
    for k, e in sorted(((k, e) for k, e in res.items() if not k.startswith("_")), key=lambda x: -x[1]["hbm_bytes_per_launch"]):
        f.write("%s,%d,%.1f,%.1f,%.0f,%.0f,%.0f\n" % (k, e.get("FETCH_SIZE_launches", 0), e.get("FETCH_SIZE_KB_per_launch_raw", 0),
                                                   e.get("WRITE_SIZE_KB_per_launch_raw", 0), e["read_bytes_per_launch"],
                                                   e["write_bytes_per_launch"], e["hbm_bytes_per_launch"]))
open("profiles/LATEST", "w").write(tag + "\n")      # bench.py reads roofline.traffic from the summary this names
print(open("profiles/%s_pmc_hbm_traffic.csv" % tag).read())
