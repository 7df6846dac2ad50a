# position-split sizes of the grouped weight gradients per backward phase: training throughput for a few settings
for v in "" 8,4,4,2 8,2,2,2 8,1,1,2 8,2,2,1 16,2,2,2 4,2,2,2 8,1,1,1; do
  echo "DDPM_TPS_PHASES=$v"
  DDPM_TPS_PHASES=$v timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-sample --no-cpu --no-roofline --no-extra 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
done
