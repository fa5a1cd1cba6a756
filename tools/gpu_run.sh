#!/bin/bash
# One parametrised GPU-session script (replaces the per-session tools/gpu_r*.sh of rounds 3-4).
#   gpurun --timeout N -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>/; steps are joined with && semantics (set -e): a failed or killed GPU step ends
# the session.  Steps:
#   tests:<files>[@<-k expression>]              pytest -m gpu on the given files (spaces written as '+')
#   bench[:extra args]                           python bench.py --steps 20 --warmup 5 [extra]
#   benchq[:extra args]                          quick bench line: no CPU baseline, no oracle, no kernel events
#   prof[:single|side]                           rocprofv3 --kernel-trace --stats of the bench command (SC_OVERLAP=0 / default)
#   pmc                                          the FETCH_SIZE / WRITE_SIZE / MFMA PMC passes + summaries
#   py:<script and args>                         python <script ...> (tools/*.py micro-benchmarks)
#   pmcpy:<script and args>                      the three PMC passes over python3 <script ...>
#   pmcx:<CTR,CTR,...>@<script and args>         one rocprofv3 --pmc pass with the named counters over python3 <script ...>
#   env:VAR=VALUE / unset:VAR                    environment of the steps that follow (A/B switches)
set -e -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
R=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
for STEP in "$@"; do
  KIND=${STEP%%:*}
  ARG=""; [[ "$STEP" == *:* ]] && ARG=${STEP#*:}
  ARG=${ARG//+/ }
  echo "=== [$TAG] $KIND $ARG ($(date +%T))"
  case $KIND in
    tests)  # tests:<files>[@<-k expression>]
      FILES=${ARG%%@*}; KEXPR=""; [[ "$ARG" == *@* ]] && KEXPR=${ARG#*@}
      LOG="$OUT/tests_$(echo "$ARG" | tr -c 'A-Za-z0-9' _ | cut -c1-60).log"
      if [ -n "$KEXPR" ]; then timeout -k 10 1100 python -m pytest $FILES -k "$KEXPR" -m gpu -x -q -s 2>&1 | tee "$LOG" | tail -40
      else timeout -k 10 1100 python -m pytest $FILES -m gpu -x -q -s 2>&1 | tee "$LOG" | tail -40; fi ;;
    bench)  # (file name = first 40 characters of the arguments + a checksum of all of them: two long argument lists that share a prefix must not overwrite each other)
      BN="bench_$(echo "$ARG" | tr -c 'A-Za-z0-9' _ | cut -c1-40)_$(echo "$ARG" | cksum | cut -d' ' -f1)"
      timeout -k 10 900 python bench.py --steps 20 --warmup 5 $ARG > "$OUT/$BN.json" 2> "$OUT/$BN.err" ; tail -c 400 "$OUT/$BN.json" ;;
    benchq) timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events $ARG 2> "$OUT/benchq.err" | tee -a "$OUT/benchq.jsonl" | cut -c1-400 ;;
    prof)
      MODE=${ARG%% *}; MODE=${MODE:-single}; EXTRA=""; [[ "$ARG" == *" "* ]] && EXTRA=${ARG#* }      # prof:single+--model+X ...: extra bench.py arguments
      D=$PWD/$OUT/prof_$MODE; rm -rf $D
      # single: every kernel alone on the chip -- weight gradients on the chain stream, the optimiser as one launch in front of the forward AND the two towers in sequence
      if [ "$MODE" = single ]; then export SC_OVERLAP=0 SC_ADAMW_BEHIND=0 SC_TOWER_OVERLAP=0; else unset SC_OVERLAP SC_ADAMW_BEHIND SC_TOWER_OVERLAP; fi
      (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta $EXTRA > $R/$OUT/prof_$MODE.log 2>&1)
      unset SC_OVERLAP SC_ADAMW_BEHIND SC_TOWER_OVERLAP
      find $D -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_$MODE.csv" \;
      find $D -name '*kernel_trace.csv' -size +20M -delete
      head -30 "$OUT/kernel_stats_$MODE.csv" | cut -c1-200 ;;
    pmc)
      export SC_OVERLAP=0 SC_ADAMW_BEHIND=0
      B="python3 $R/bench.py --no-cpu-baseline --no-kernel-events --no-loss-delta"
      (cd /tmp && timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_f -o f -- $B --steps 3 --warmup 1 > $R/$OUT/pmc_f.log 2>&1)
      (cd /tmp && timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_w -o w -- $B --steps 3 --warmup 1 > $R/$OUT/pmc_w.log 2>&1)
      (cd /tmp && timeout -k 10 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$OUT/pmc_m -o m -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_m.log 2>&1)
      unset SC_OVERLAP SC_ADAMW_BEHIND
      python tools/pmc_summary.py $(find $OUT/pmc_f -name "f_counter_collection.csv") $(find $OUT/pmc_w -name "w_counter_collection.csv") $OUT/pmc_traffic_summary.json > $OUT/pmc_traffic.txt 2>&1; head -14 $OUT/pmc_traffic.txt
      python tools/pmc_generic.py $OUT/pmc_mfma_summary.json "$OUT/pmc_m/**/m_counter_collection.csv" > $OUT/pmc_mfma.txt 2>&1; head -14 $OUT/pmc_mfma.txt
      find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete ;;
    pmcpy)  # pmcpy:<script and args>: FETCH_SIZE / WRITE_SIZE / MFMA passes over a tools/*.py micro-benchmark (N=3 REPS=1 keep it short)
      T="$(echo "$ARG" | tr -c 'A-Za-z0-9' _ | cut -c1-40)"
      (cd /tmp && N=3 REPS=1 timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmcpy_f_$T -o f -- python3 $R/$ARG > $R/$OUT/pmcpy_f_$T.log 2>&1)
      (cd /tmp && N=3 REPS=1 timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmcpy_w_$T -o w -- python3 $R/$ARG > $R/$OUT/pmcpy_w_$T.log 2>&1)
      (cd /tmp && N=3 REPS=1 timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$OUT/pmcpy_m_$T -o m -- python3 $R/$ARG > $R/$OUT/pmcpy_m_$T.log 2>&1)
      python tools/pmc_summary.py $(find $OUT/pmcpy_f_$T -name "f_counter_collection.csv") $(find $OUT/pmcpy_w_$T -name "w_counter_collection.csv") $OUT/pmcpy_traffic_$T.json > $OUT/pmcpy_traffic_$T.txt 2>&1; head -14 $OUT/pmcpy_traffic_$T.txt
      python tools/pmc_generic.py $OUT/pmcpy_mfma_$T.json "$OUT/pmcpy_m_$T/**/m_counter_collection.csv" > $OUT/pmcpy_mfma_$T.txt 2>&1; head -14 $OUT/pmcpy_mfma_$T.txt
      find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete ;;
    pmcx)   # pmcx:<COUNTER,COUNTER,...>@<script and args>: ONE counter pass (kernel-trace only beside it) over a tools/*.py micro-benchmark
      CTRS=${ARG%%@*}; CMD=${ARG#*@}
      T="$(echo "$CTRS $CMD" | tr -c 'A-Za-z0-9' _ | cut -c1-60)"
      (cd /tmp && N=${PMC_N:-3} REPS=1 timeout -k 10 600 rocprofv3 --pmc ${CTRS//,/ } --kernel-trace --output-format csv -d $R/$OUT/pmcx_$T -o x -- python3 $R/$CMD > $R/$OUT/pmcx_$T.log 2>&1)
      python tools/pmc_generic.py $OUT/pmcx_$T.json "$OUT/pmcx_$T/**/x_counter_collection.csv" > $OUT/pmcx_$T.txt 2>&1; head -14 $OUT/pmcx_$T.txt
      find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete ;;
    env)    export "$ARG"; echo "exported $ARG" ;;
    unset)  unset "$ARG"; echo "unset $ARG" ;;
    py)     timeout -k 10 900 python $ARG 2>&1 | tee "$OUT/py_$(echo "$ARG" | tr -c 'A-Za-z0-9' _ | cut -c1-60).log" | tail -60 ;;
    *) echo "unknown step $KIND"; exit 2 ;;
  esac
done
echo "=== [$TAG] done ($(date +%T))"
