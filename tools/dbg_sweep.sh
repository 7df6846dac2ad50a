set -e
for mx in 0 64 128 100000; do
  echo "gnb max cout=$mx: $(DDPM_GNB_MAX=$mx timeout -k 10 200 python -u bench.py --no-sample --no-cpu --no-roofline --steps 60 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done
