# Runs on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh'): the bench line, the rocprofv3 kernel statistics of the same
# command, and the two PMC passes for HBM traffic; condensed results land in gpurun_out/final/ (copy them to profiles/).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 -u $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -o r -- python3 $R/bench.py --steps 20 --warmup 5 --no-sample --no-cpu --no-roofline --no-extra > $OUT/bench_under_rocprof.json 2> /tmp/prof/stats.err
python3 $R/tools/profile_summary.py stats /tmp/prof/stats $OUT/kernel_stats.csv
python3 $R/tools/micro/timeline.py $(find /tmp/prof/stats -name '*kernel_trace.csv' | head -1) > $OUT/timeline.txt
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/fetch -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-sample --no-cpu --no-roofline --no-graph --no-extra > /tmp/prof/fetch.out 2> /tmp/prof/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/write -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-sample --no-cpu --no-roofline --no-graph --no-extra > /tmp/prof/write.out 2> /tmp/prof/write.err
python3 $R/tools/profile_summary.py traffic /tmp/prof/fetch /tmp/prof/write $OUT/hbm_traffic.json
# matrix-core evidence: one counter pass of its own (no trace domains beside --pmc), the program directly after --
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof/mfma -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-sample --no-cpu --no-roofline --no-graph --no-extra > /tmp/prof/mfma.out 2> /tmp/prof/mfma.err
python3 $R/tools/profile_summary.py mfma /tmp/prof/mfma /tmp/prof/stats $OUT/mfma_util.json
# the same two passes on the 64x64x3 training step (batch 128): the 3x3 convolutions with Cin >= 128 at 32x32 and 16x16 positions
# are the matrix-core-bound shapes of this network
B64="--size 64 --channels 3 --batch 128 --no-sample --no-cpu --no-roofline --no-extra"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats64 -o r -- python3 $R/bench.py --steps 10 --warmup 3 $B64 > /tmp/prof/stats64.out 2> /tmp/prof/stats64.err
python3 $R/tools/profile_summary.py stats /tmp/prof/stats64 $OUT/kernel_stats_64x64x3.csv
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof/mfma64 -o r -- python3 $R/bench.py --steps 3 --warmup 2 --no-graph $B64 > /tmp/prof/mfma64.out 2> /tmp/prof/mfma64.err
python3 $R/tools/profile_summary.py mfma /tmp/prof/mfma64 /tmp/prof/stats64 $OUT/mfma_util_64x64x3.json
# the attention kernels: a training step at 128x128x3 with self-attention at the 16x16 stage (stage 2), kernel statistics and
# the matrix-core counters in passes of their own
BAT="--size 128 --channels 3 --batch 16 --attention-stages 2 --no-sample --no-cpu --no-roofline --no-extra"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/statsat -o r -- python3 $R/bench.py --steps 10 --warmup 3 $BAT > /tmp/prof/statsat.out 2> /tmp/prof/statsat.err
python3 $R/tools/profile_summary.py stats /tmp/prof/statsat $OUT/kernel_stats_128x128x3_attention.csv
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof/mfmaat -o r -- python3 $R/bench.py --steps 3 --warmup 2 --no-graph $BAT > /tmp/prof/mfmaat.out 2> /tmp/prof/mfmaat.err
python3 $R/tools/profile_summary.py mfma /tmp/prof/mfmaat /tmp/prof/statsat $OUT/mfma_util_128x128x3_attention.json
# the attention forward at a saturating batch: sampling steps of 256 images (2048 head problems of 256 tokens), plain launches
export NOGRAPH=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/statsas -o r -- python3 $R/tools/micro/sample128.py 256 > /tmp/prof/statsas.out 2> /tmp/prof/statsas.err
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof/mfmaas -o r -- python3 $R/tools/micro/sample128.py 256 > /tmp/prof/mfmaas.out 2> /tmp/prof/mfmaas.err
unset NOGRAPH
python3 $R/tools/profile_summary.py mfma /tmp/prof/mfmaas /tmp/prof/statsas $OUT/mfma_util_sample128_attention.json
ls -la $OUT
