# kernel trace of a short plan-mode bench run + tools/timeline.py on a step inside the timed loop
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python bench.py --steps 8 --warmup 6 --no-extra --no-cpu-baseline > gpurun_out/timeline_bench.log 2>&1
python tools/timeline.py $(find /tmp/prof_tl -name "tl_results.db" | head -1) 25 10 > gpurun_out/timeline.txt 2>&1
