#!/bin/bash
# scratch: a sweep of short bench runs (single GPU call)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/sw
run() { n=$1; shift; env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-k63 --no-bloom-mphf --no-host-landed --no-share-of-8 $EXTRA > gpurun_out/sw/$n.json 2> gpurun_out/sw/$n.err
python - <<PY
import json
d=json.load(open("gpurun_out/sw/$n.json")); t=d["config"]["kernel_ms_per_step"]; s=d["roofline"]["single_lane"]["kernel_ms_per_step"]
print("$n: ms_per_step %.1f A %.1f B %.1f | single dedupe %.1f count %.1f scatter %.1f sort %.1f big %.1f wg %.1f split %.1f gather %.1f B %.1f" % (d["ms_per_step"], t["total_stage_a"], t["total_stage_b"], s["dedupe_bin"] + s["dedupe_sort"], s["expand_count"], s["expand_scatter"], s["bucket_sort"], s["bucket_sort_big"], s["bucket_sort_wg"], s["split_levels"], s["compact"], s["total_stage_b"]))
PY
}
V=$R/gatb-core_amd/csrc/variants
EXTRA="" run t48 A=1
EXTRA="" run t96 GKC_LIB=$V/libgkc_hip_t96.so
EXTRA="" run t160 GKC_LIB=$V/libgkc_hip_t160.so
EXTRA="" run t24 GKC_LIB=$V/libgkc_hip_t24.so
