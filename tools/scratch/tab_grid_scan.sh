for g in 768 1024 1280; do
  echo "n=524288 grid $g: $(CLSIMHIP_GRID=$g python3 bench.py --workload tab --bunch 524288 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["kernel_ms_per_pass"])')"
done
for g in 768 1024; do
  echo "n=262144 grid $g: $(CLSIMHIP_GRID=$g python3 bench.py --workload tab --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["kernel_ms_per_pass"])')"
done
