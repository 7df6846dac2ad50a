set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && mkdir -p /tmp/tl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-sample --no-cpu --no-roofline --no-extra > /tmp/tl/bench.out 2> /tmp/tl/bench.err || { tail -5 /tmp/tl/bench.err; exit 1; }
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/micro/timeline.py $f $GRAFT_REPO_ROOT/gpurun_out/main_sequence.txt > $GRAFT_REPO_ROOT/gpurun_out/timeline.txt
tail -1 /tmp/tl/bench.out | cut -c1-300
