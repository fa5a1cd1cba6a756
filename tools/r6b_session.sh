set -e
export TMPDIR=/tmp
R=$PWD; OUT=gpurun_out/r6b; mkdir -p $OUT
echo "== clock stamps"; SC_HIP_LIB=$R/spatial-clip_amd/lib/libspatialclip_hip_clk.so timeout -k 10 300 python tools/attn_bwd3_clock.py 2>&1 | tee $OUT/attn_bwd3_clock.txt
echo "== wall only, default library"; timeout -k 10 300 python tools/attn_bwd3_clock.py 2>&1 | tee $OUT/attn_bwd3_wall.txt
for G in 128 256; do
  export GRIDS=$G
  bash tools/gpu_run.sh r6b "pmcx:GRBM_GUI_ACTIVE,TCC_EA0_RDREQ_sum,TCC_TAG_STALL_sum,TCP_PENDING_STALL_CYCLES_sum@tools/attn_bwd3_clock.py" || echo "pmcx failed for $G"
  mv $OUT/pmcx_*.json $OUT/attn_pmc_grid$G.json 2>/dev/null || true
  mv $OUT/pmcx_*.txt $OUT/attn_pmc_grid$G.txt 2>/dev/null || true
  mv $OUT/pmcx_*.log $OUT/attn_pmc_grid$G.log 2>/dev/null || true
  rm -rf $OUT/pmcx_*
done
unset GRIDS
echo "== vendor kernel names"
(cd /tmp && WHICH=gemm N=4 REPS=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/yard -o y -- python3 $R/tools/vendor_yardstick.py > $R/$OUT/yard.log 2>&1)
find $OUT/yard -name '*kernel_stats.csv' -exec cp {} $OUT/yardstick_kernel_stats.csv \;
find $OUT -name '*kernel_trace.csv' -size +5M -delete
head -40 $OUT/yardstick_kernel_stats.csv | cut -c1-260
