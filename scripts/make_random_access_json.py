#!/usr/bin/env python3
"""usage: python scripts/make_random_access_json.py <tag>
profiles/<tag>_random_access.json from this round's measurements: the random-access roofline run (gpurun_out/ra_<tag>.txt, output of
scripts/random_access_roofline.hip, copied to profiles/<tag>_random_access_roofline.txt) and the memory read requests k_find issues per
query (TCC_EA0_RDREQ of the bench command, gpurun_out/l2_<tag>.txt -> profiles/<tag>_l2_counters.txt).  bench.py prices its find path with it."""
import json
import re
import shutil
import sys

tag = sys.argv[1]
ra = open("gpurun_out/ra_%s.txt" % tag).read()
shutil.copy("gpurun_out/ra_%s.txt" % tag, "profiles/%s_random_access_roofline.txt" % tag)
shutil.copy("gpurun_out/l2_%s.txt" % tag, "profiles/%s_l2_counters.txt" % tag)


def rate(label, wg):
    for line in ra.splitlines():
        if line.startswith(label) and ("%d WG/CU" % wg) in line:
            return float(re.search(r"([0-9.e+]+) touches/s", line).group(1))
    return None


rd = None
for line in open("gpurun_out/l2_%s.txt" % tag):
    if line.startswith("TCC_EA0_RDREQ_sum") and "k_find" in line:
        rd = float(line.split("total")[1])
out = {
    "what": "random 64-byte sector reads of a 2 GiB buffer on one MI355X (scripts/random_access_roofline.hip, output in %s_random_access_roofline.txt, "
            "measured this round) and the memory read requests k_find issues per query (TCC_EA0_RDREQ of the bench command, %s_l2_counters.txt)" % (tag, tag),
    "peak_sector_touches_per_s": {"independent_1e8": rate("sector, independent  ", 4), "two_dependent_1e8": rate("sector, two dependent touches", 4),
                                  "independent_1e7": rate("sector, independent, 1e7 queries", 8), "two_dependent_1e7": rate("sector, two touches, 1e7 queries", 8)},
    "slot16_touches_per_s": rate("slot (16 B), independent", 4), "line128_touches_per_s": rate("128-byte line, independent", 4),
    "k_find_rdreq_per_query": rd / 1e7 if rd else None,
}
out["peak_used"] = out["peak_sector_touches_per_s"]["two_dependent_1e7"]
out["peak_used_why"] = ("two dependent 64-byte touches per query at the find workload's size (10^7 queries, launch ramp and tail included): the access pattern "
                        "closest to a probe that needs a second sector half of the time.  Excluded: the 1e8-query figures (no ramp / tail at the workload's "
                        "size), 16-byte touches (5.4e10/s: a probe reads whole sectors, not single slots) and 128-byte lines (fewer, larger touches)")
json.dump(out, open("profiles/%s_random_access.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
