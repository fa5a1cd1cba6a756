set -e
export TMPDIR=/tmp
R=$PWD; OUT=gpurun_out/r6k; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_text_fullsize.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tee $OUT/pytest_subset.log | tail -6
bash tools/gpu_run.sh r6k benchq:--model+ViT-B-32+--batch+32 benchq:--model+ViT-B-32 benchq:--model+ViT-B-16 benchq:--model+ViT-L-14-genetr+--loss+spatial benchq:--model+ViT-L-14-genetr+--loss+spatial+--dtype+fp8 benchq
export SC_TOWER_OVERLAP=0
bash tools/gpu_run.sh r6k benchq:--model+ViT-B-32+--batch+32 benchq:--model+ViT-B-32 benchq:--model+ViT-B-16 benchq:--model+ViT-L-14-genetr+--loss+spatial benchq:--model+ViT-L-14-genetr+--loss+spatial+--dtype+fp8
