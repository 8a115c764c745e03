cd $GRAFT_REPO_ROOT
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-bloom-mphf --no-host-landed --no-share-of-8 > gpurun_out/sw_$n.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/sw_$n.json"))
s = d["roofline"]["single_lane"]["kernel_ms_per_step"]; t = d["config"]["kernel_ms_per_step"]
print("$n: ms_per_step %.1f  B two-lane %.1f  k63 %.1f | single B %.1f | timed sort %.1f scatter %.1f gather %.1f split %.1f" % (d["ms_per_step"], t["total_stage_b"], d["config"]["k63"]["ms_per_step"], s["total_stage_b"], t["bucket_sort"], t["expand_scatter"], t["compact"], t["split_levels"]))
PY
}
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
run base A=1
run l3 GKC_STAGEB_LANES=3
run l3w144 GKC_STAGEB_LANES=3 GKC_SCATTER_WGS=144
run w144 GKC_SCATTER_WGS=144
run w208 GKC_SCATTER_WGS=208
run w100000 GKC_SCATTER_WGS=100000
