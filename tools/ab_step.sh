# A/B of the training step under two settings of one environment variable: tools/ab_step.sh VAR A B [bench flags]
set -e
var=$1; a=$2; b=$3; shift 3
for v in $a $b $a $b; do
  echo "$var=$v: $(env $var=$v timeout -k 10 200 python -u bench.py --no-sample --no-cpu --no-roofline --no-extra --steps 100 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done
