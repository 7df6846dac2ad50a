# step time under several sets of bench flags, interleaved twice: tools/sweep_flags.sh "--phases 4" "--phases 5" ...
for rep in 1 2; do for cfg in "$@"; do
  echo "$cfg: $(timeout -k 10 200 python -u bench.py --no-sample --no-cpu --no-roofline --no-extra --steps 100 $cfg 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done; done
