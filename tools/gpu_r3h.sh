#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
PROBE_STEP=1 PROBE_BWD=1 PROBE_OPT=1 timeout -k 10 100 python tools/leak_probe.py 2>&1 | grep -E "STEP=|after gc|alive:" > $O/leak.txt; cat $O/leak.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
