#!/bin/bash
# usage (GPU box, repo root): bash scripts/profile_all.sh <tag>   -- everything profiles/<tag>_* is made from: kernel stats (headline and
# with the informational legs), FETCH/WRITE PMC passes, L2 counters, the random-access roofline.  Then, here: python scripts/summarize_profiles.py
# <tag> && python scripts/make_random_access_json.py <tag>
TAG=${1:-r3a}
OUT=gpurun_out
mkdir -p $OUT
bash scripts/profile_round.sh $TAG && bash scripts/profile_l2.sh $TAG > $OUT/l2_$TAG.txt 2>&1 && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ra_roofline scripts/random_access_roofline.hip && timeout -k 10 120 /tmp/ra_roofline > $OUT/ra_$TAG.txt 2>&1
echo "profile_all $TAG rc=$?"
