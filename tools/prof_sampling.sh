# Runs on the GPU box: rocprofv3 kernel statistics of the headline sampler (2 x 512 images, a few steps) -> gpurun_out/
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s && mkdir -p /tmp/prof_s
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o r -- python3 $R/tools/micro/sample32.py > $OUT/prof_sampling.out 2> $OUT/prof_sampling.err
python3 $R/tools/profile_summary.py stats /tmp/prof_s $OUT/sampling_stats.csv
