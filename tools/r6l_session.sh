set -e
export TMPDIR=/tmp
R=$PWD; OUT=gpurun_out/r6l; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest_gpu_full.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_run.sh r6l bench "bench:--model+ViT-B-16+--loss+spatial" "bench:--model+ViT-B-32+--loss+spatial" "bench:--batch+32+--model+ViT-B-32+--loss+spatial"
