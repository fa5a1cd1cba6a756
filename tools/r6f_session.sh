set -e
export TMPDIR=/tmp
R=$PWD; OUT=gpurun_out/r6f; mkdir -p $OUT
timeout -k 10 300 python tools/fp8_ktile_probe.py 2>&1 | tee $OUT/fp8_probe_wall.txt
for D in random zeros; do
  export DATA=$D PMC_N=6
  bash tools/gpu_run.sh r6f "pmcx:GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_LDS,SQ_INSTS_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE@tools/fp8_ktile_probe.py" || echo "pmcx failed"
  mv $OUT/pmcx_*.json $OUT/fp8_pmc_$D.json 2>/dev/null || true
  mv $OUT/pmcx_*.txt $OUT/fp8_pmc_$D.txt 2>/dev/null || true
  rm -rf $OUT/pmcx_*
done
unset DATA PMC_N
bash tools/gpu_run.sh r6f benchq:--model+ViT-B-32+--batch+32 benchq:--model+ViT-B-32 "py:tools/bench_shards_training.py" 
