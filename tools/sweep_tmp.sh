cd $GRAFT_REPO_ROOT
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-k63 --no-bloom-mphf --no-host-landed --no-share-of-8 > gpurun_out/sw_$n.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/sw_$n.json"))
t = d["config"]["kernel_ms_per_step"]; s = d["roofline"]["single_lane"]["kernel_ms_per_step"]
print("$n: ms_per_step %.1f B %.1f | single dedupe %.1f count %.1f scatter %.1f sort %.1f split %.1f gather %.1f B %.1f" % (d["ms_per_step"], t["total_stage_b"], s["dedupe"], s["expand_count"], s["expand_scatter"], s["bucket_sort"], s["split_levels"], s["compact"], s["total_stage_b"]))
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_exact or edge or alternative" 2>&1 | tail -2
run base A=1
run w208 GKC_SCATTER_WGS=208
run w240 GKC_SCATTER_WGS=240
run sub12 GKC_MAX_SUB_BITS=12
run sub12w208 GKC_MAX_SUB_BITS=12 GKC_SCATTER_WGS=208
run nodd GKC_DEDUPE=0
