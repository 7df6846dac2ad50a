set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p128 && mkdir -p /tmp/p128
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p128 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --size 128 --channels 3 --batch 8 --attention-stages 2 --steps 1 --warmup 1 --no-cpu --no-roofline --sample-batch 128 > /tmp/p128/out.json 2> /tmp/p128/err.txt || { tail -5 /tmp/p128/err.txt; exit 1; }
f=$(find /tmp/p128 -name '*kernel_stats.csv' | head -1)
head -22 $f > $GRAFT_REPO_ROOT/gpurun_out/p128_stats.csv
tail -1 /tmp/p128/out.json | cut -c1-120
