#!/bin/bash
# Usage (on the GPU box, repo root): tools/profile_round.sh <tag>   -> text summaries under gpurun_out/profiles_<tag>/
set -u
TAG=${1:-rXX}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --single-stream"   # kernels timed alone (same mode as bench.py's roofline step)
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $CMD > $OUT/bench_kernel_trace.log 2>&1
python tools/prof_summary.py /tmp/prof_kt/kt_results.db 40 > $OUT/kernel_stats.txt
# the launch list of the TIMED step: the default mode (recorded launch plan replayed on four streams).  The eager single-stream run
# above additionally launches torch fill / copy / cast kernels that a plan replay does not contain; rocprofv3
# serialises dispatches, so the durations of both passes are those of kernels running alone.
PCMD="python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d /tmp/prof_plan -o kp -- $PCMD > $OUT/bench_kernel_trace_plan.log 2>&1
python tools/prof_summary.py /tmp/prof_plan/kp_results.db 60 > $OUT/kernel_stats_plan.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d /tmp/prof_a -o a -- $CMD > /dev/null 2>&1
python tools/pmc_summary.py /tmp/prof_a/a_results.db > $OUT/pmc_mfma.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_b -o b -- $CMD > /dev/null 2>&1
python tools/pmc_summary.py /tmp/prof_b/b_results.db > $OUT/pmc_fetch.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_c -o c -- $CMD > /dev/null 2>&1
python tools/pmc_summary.py /tmp/prof_c/c_results.db > $OUT/pmc_write.txt
tail -1 $OUT/bench_kernel_trace.log | cut -c1-400
