set -e
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for pr in 0 1 -1; do
  echo "side priority=$pr: $(DDPM_SIDE_PRIORITY=$pr timeout -k 10 200 python -u bench.py --no-sample --no-cpu --no-roofline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done
