# Runs on the GPU box: the matrix-core counters of the convolution sweep, one counter pass (rocprofv3 --pmc, nothing else) per
# case, the program directly after "--"; condensed into gpurun_out/mfma_conv_sweep.json (copy to profiles/).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cs && mkdir -p /tmp/prof_cs
i=0
DEFAULT="128,128,16,4096,fwd,pipe 128,128,16,4096,dgrad,pipe 256,256,8,16384,fwd,pipe 256,256,8,16384,dgrad,pipe 256,512,4,65536,fwd,pipe 512,256,4,65536,fwd,pipe 512,256,4,65536,dgrad,pipe 128,128,16,4096,fwd,old 256,256,8,16384,fwd,old 512,256,4,65536,fwd,old 128,128,16,4096,wgrad,old 256,256,8,16384,wgrad,old 512,256,4,65536,wgrad,old 128,128,16,4096,wgrad,pipe 256,256,8,16384,wgrad,pipe 256,512,4,65536,wgrad,pipe 512,256,4,65536,wgrad,pipe"
CASES=${CASES:-$DEFAULT}
for case in $CASES; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_cs/c$i -o r -- python3 $R/tools/micro/conv_sweep.py --one $case --reps 5 > /tmp/prof_cs/c$i.json 2> $OUT/pcs_c$i.err || echo "case $i failed: $?"
  echo "case $i $case done"
done
python3 $R/tools/micro/conv_sweep_summary.py /tmp/prof_cs $OUT/mfma_conv_sweep.json
