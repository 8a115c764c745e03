python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliced or size_independent or alternative or fuzz or bit_exact or oversize" 2>&1 | tail -6
run() { tag=$1; shift; env "$@" python bench.py --partitions ${P:-256} --steps 2 --warmup 1 --no-cpu-baseline --no-host-landed --no-k63 --no-bloom-mphf --no-share-of-8 --no-two-pass > gpurun_out/c_$tag.json 2> gpurun_out/c_$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/c_$tag.json")); k=d["config"]["kernel_ms_per_step"]
    print("$tag", round(d["ms_per_step"],1), d["verified"], {a:round(b) for a,b in k.items()})
    print("   single", {a:round(b) for a,b in d["roofline"]["single_lane"]["kernel_ms_per_step"].items()})
except Exception as e: print("$tag failed", e)
PY
}
P=256 run p256 A=1
P=256 run p256_off GKC_SLICES=0
P=64 run p64 A=1
P=1024 run p1024 A=1
P=1024 run p1024_s GKC_SLICE_MIN=8000000
P=4096 run p4096 A=1
