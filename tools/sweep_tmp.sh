cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for e in 2 1 0; do
  GKC_SPLIT_EXTRA=$e python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-k63 --no-bloom-mphf --no-host-landed > gpurun_out/sw_$e.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/sw_$e.json"))
s = d["roofline"]["single_lane"]["kernel_ms_per_step"]
print("extra=$e ms_per_step %.1f" % d["ms_per_step"], "single: split %.1f compact %.1f big %.1f wg %.1f sort %.1f B %.1f" % (s["split_levels"], s["compact"], s["bucket_sort_big"], s["bucket_sort_wg"], s["bucket_sort"], s["total_stage_b"]))
PY
done
