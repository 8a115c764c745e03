#!/bin/bash
# scratch: a sweep of short bench runs (single GPU call)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/sw
run() { n=$1; shift; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-k63 --no-bloom-mphf --no-host-landed --no-share-of-8 --no-two-pass $EXTRA > gpurun_out/sw/$n.json 2> gpurun_out/sw/$n.err
python - <<PY
import json
d=json.load(open("gpurun_out/sw/$n.json")); t=d["config"]["kernel_ms_per_step"]; s=d["roofline"]["single_lane"]["kernel_ms_per_step"]
print("$n: ms_per_step %.1f A %.1f B %.1f | single: %s" % (d["ms_per_step"], t["total_stage_a"], t["total_stage_b"], {k: round(v, 1) for k, v in s.items()}))
PY
}
EXTRA="--partitions 256" run p256 A=1
EXTRA="--partitions 1024" run p1024 A=1
