set -e
export TMPDIR=/tmp
R=$PWD; OUT=gpurun_out/r6g; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest_gpu_full.log | tail -15
timeout -k 10 300 python tools/fp8_ktile_probe.py 2>&1 | tee $OUT/fp8_probe_wall.txt
export DATA=random PMC_N=6
bash tools/gpu_run.sh r6g "pmcx:GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_LDS,SQ_INSTS_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE@tools/fp8_ktile_probe.py" || echo "pmcx failed"
mv $OUT/pmcx_*.json $OUT/fp8_pmc_random_after.json 2>/dev/null || true
mv $OUT/pmcx_*.txt $OUT/fp8_pmc_random_after.txt 2>/dev/null || true
rm -rf $OUT/pmcx_*
unset DATA PMC_N
bash tools/gpu_run.sh r6g py:tools/bench_fp8.py
