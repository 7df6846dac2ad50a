set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
DDPM_TAIL_ON_MAIN=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -o r -- python3 $R/bench.py --steps 20 --warmup 5 --no-sample --no-cpu --no-roofline --no-extra > /tmp/prof/o.json 2> /tmp/prof/stats.err
python3 $R/tools/micro/timeline.py $(find /tmp/prof/stats -name '*kernel_trace.csv' | head -1) > $R/gpurun_out/timeline_tail.txt
