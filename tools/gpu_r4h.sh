#!/bin/bash
# round 4: whole GPU suite after the verdict / advisor items (fp8 state, head at the configs[4] size, e4m3 bounds, grouping, ring attention)
O=$PWD/gpurun_out/r4h; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; rc=$?; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | tail -20
grep -E "configs4|\[head|fp8 model" $O/tests.txt | head
exit $rc
