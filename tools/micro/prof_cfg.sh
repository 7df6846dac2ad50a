# rocprofv3 kernel statistics of one bench configuration: bash tools/micro/prof_cfg.sh <out file> <bench.py arguments...>
set -e
R=$GRAFT_REPO_ROOT
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/s -o r -- python3 $R/bench.py --no-cpu --no-roofline --no-extra "$@" > /tmp/prof/o.txt 2> /tmp/prof/err.txt
python3 - "$R/$OUT" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/prof/s/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(sys.argv[1], 'w') as o:
    o.write(f"total kernel ms {tot / 1e6:.1f}\n")
    for r in rows[:30]:
        o.write(f"{r['Name'][:72]:72s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%\n")
PY
