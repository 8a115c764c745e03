run() { python bench.py --k $1 --partitions $2 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/err.txt | python -c "
import json,sys
L=sys.stdin.read().strip().splitlines()
if L: d=json.loads(L[-1]); print(d['ms_per_step'], d['config']['kernel_ms_per_step'])
else: print('FAILED')"; tail -n 2 gpurun_out/err.txt | grep -i error; }
export GKC_SCAN_COARSE_MAX=8192
for P in 6144 8192; do echo "k63 P=$P"; run 63 $P; done
for P in 6144 8192; do echo "k31 P=$P"; run 31 $P; done
