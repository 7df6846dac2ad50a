set -e
for ph in 4 5 6; do for tl in 1 2 4; do
  echo "phases=$ph tps_last=$tl: $(DDPM_TPS_LAST=$tl timeout -k 10 200 python -u bench.py --phases $ph --no-sample --no-cpu --no-roofline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done; done
