set -e
export TMPDIR=/tmp
R=$PWD; OUT=gpurun_out/r6i; mkdir -p $OUT
bash tools/gpu_run.sh r6i "bench:--model+ViT-L-14-genetr+--loss+spatial+--no-cpu-baseline+--no-loss-delta" "bench:--model+ViT-L-14-genetr+--loss+spatial+--dtype+fp8+--no-cpu-baseline+--no-loss-delta"
bash tools/gpu_run.sh r6i "bench:--model+ViT-B-16+--loss+spatial" "bench:--model+ViT-B-32+--loss+spatial" "bench:--model+ViT-B-32+--batch+32+--loss+spatial"
bash tools/gpu_run.sh r6i prof:single prof:side
(cd /tmp && SC_OVERLAP=0 SC_ADAMW_BEHIND=0 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/census -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --graph off > $R/$OUT/census.log 2>&1)
python tools/step_launch_census.py "$OUT/census/**/t_kernel_trace.csv" 4 | tee $OUT/step_launch_census.txt
find $OUT -name '*kernel_trace.csv' -delete
