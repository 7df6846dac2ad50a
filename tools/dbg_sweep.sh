set -e
for ph in 2 3 4; do
  echo "phases=$ph: $(timeout -k 10 200 python -u bench.py --phases $ph --no-sample --no-cpu --no-roofline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done
