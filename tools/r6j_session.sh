set -e
export TMPDIR=/tmp
R=$PWD; OUT=gpurun_out/r6j; mkdir -p $OUT
bash tools/gpu_run.sh r6j "bench:--model+ViT-L-14-genetr+--loss+spatial+--no-cpu-baseline+--no-loss-delta" "bench:--dtype+fp8+--model+ViT-L-14-genetr+--loss+spatial+--no-cpu-baseline+--no-loss-delta"
bash tools/gpu_run.sh r6j pmc
bash tools/gpu_run.sh r6j "prof:single+--model+ViT-L-14-genetr+--loss+spatial+--dtype+fp8+--graph+off"
cp $OUT/kernel_stats_single.csv $OUT/kernel_stats_vitl_fp8_single.csv
bash tools/gpu_run.sh r6j "prof:single+--model+ViT-L-14-genetr+--loss+spatial+--graph+off"
cp $OUT/kernel_stats_single.csv $OUT/kernel_stats_vitl_bf16_single.csv
