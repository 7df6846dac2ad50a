# step time under several environment settings, interleaved twice: tools/sweep_env.sh "A=1 B=2" "A=0" ...
for rep in 1 2; do for cfg in "$@"; do
  echo "$cfg: $(env $cfg timeout -k 10 200 python -u bench.py --no-sample --no-cpu --no-roofline --no-extra --steps 100 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done; done
