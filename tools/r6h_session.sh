set -e
export TMPDIR=/tmp
R=$PWD; OUT=gpurun_out/r6h; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest_gpu_full.log | tail -8
timeout -k 10 300 python tools/fp8_ktile_probe.py 2>&1 | tee $OUT/fp8_probe_wall.txt
bash tools/gpu_run.sh r6h bench bench:--loss+spatial+--no-cpu-baseline
