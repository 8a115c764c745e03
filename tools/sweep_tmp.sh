cd $GRAFT_REPO_ROOT
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-k63 --no-bloom-mphf --no-host-landed --no-share-of-8 > gpurun_out/sw_$n.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/sw_$n.json"))
s = d["roofline"]["single_lane"]["kernel_ms_per_step"]; t = d["config"]["kernel_ms_per_step"]
print("$n: ms_per_step %.1f  B two-lane %.1f  single: sort %.1f scatter %.1f B %.1f | timed sort %.1f scatter %.1f" % (d["ms_per_step"], t["total_stage_b"], s["bucket_sort"], s["expand_scatter"], s["total_stage_b"], t["bucket_sort"], t["expand_scatter"]))
PY
}
run base A=1
run permswap GKC_LIB=$GRAFT_REPO_ROOT/gatb-core_amd/csrc/variants/libgkc_hip_permswap.so
run base2 A=1
run lanes3 GKC_STAGEB_LANES=3
