cd $GRAFT_REPO_ROOT
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-k63 --no-bloom-mphf --no-host-landed --no-share-of-8 > gpurun_out/sw_$n.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/sw_$n.json"))
t = d["config"]["kernel_ms_per_step"]; s = d["roofline"]["single_lane"]["kernel_ms_per_step"]
print("$n: ms_per_step %.1f A %.1f B %.1f | single emit %.1f scatter %.1f sort %.1f gather %.1f B %.1f" % (d["ms_per_step"], t["total_stage_a"], t["total_stage_b"], s["scan_emit"], s["expand_scatter"], s["bucket_sort"], s["compact"], s["total_stage_b"]))
PY
}
for rep in 1 2; do
run base$rep A=1
run nt$rep GKC_LIB=$GRAFT_REPO_ROOT/gatb-core_amd/csrc/variants/libgkc_hip_nt.so
done
