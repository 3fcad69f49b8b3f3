#!/bin/bash
# sample socket power and clocks while the benchmark step runs (and before / after it: idle reference)
mkdir -p gpurun_out/r06k
OUT=gpurun_out/r06k/power_during_step.txt
( for i in $(seq 1 400); do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' ')"; sleep 0.25; done ) > $OUT.raw &
SP=$!
sleep 3
python bench.py --steps 600 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/r06k/power_bench.json 2> gpurun_out/r06k/power_bench.err
sleep 3
kill $SP 2>/dev/null
wait $SP 2>/dev/null
head -c 1500 $OUT.raw; echo; tail -c 1200 $OUT.raw
rocm-smi --showmaxpower 2>/dev/null | tail -5
