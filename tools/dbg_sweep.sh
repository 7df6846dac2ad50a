set -e
for tps in 4 8 16; do for tl in 2 4; do
  echo "tps=$tps last=$tl: $(DDPM_TPS=$tps DDPM_TPS_LAST=$tl timeout -k 10 200 python -u bench.py --no-sample --no-cpu --no-roofline --steps 60 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done; done
