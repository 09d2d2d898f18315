#!/bin/bash
# Samples the socket power / clocks (rocm-smi, read-only) while scripts/kbench.py loops the render launch: is the launch
# at the power cap while its shader clock sits below 2.4 GHz?  Output: gpurun_out/power_sample.txt
O=gpurun_out; mkdir -p $O
rocm-smi --showmaxpower --showpower --showclocks > $O/power_idle.txt 2>&1
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/power_sample.txt 2>&1 &
SAMPLER=$!
python3 scripts/kbench.py --reps 1500 --tag power > $O/power_kbench.txt 2>&1
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
PRV_BLOCKS_PER_CU=1 python3 scripts/kbench.py --reps 20 --tag bpc1 >> $O/power_kbench.txt 2>&1
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/power_sample_bpc1.txt 2>&1 &
SAMPLER=$!
PRV_BLOCKS_PER_CU=1 python3 scripts/kbench.py --reps 1200 --tag power_bpc1 >> $O/power_kbench.txt 2>&1
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
tail -3 $O/power_kbench.txt
