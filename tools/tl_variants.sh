# Gantt charts of the step under rocprofv3 for several launch modes (NB the tracer delays the later-enqueued queue: tools/timeline.py header).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; rm -rf /tmp/tl_$tag; env "$@" rocprofv3 --kernel-trace -d /tmp/tl_$tag -o tl -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extra $EXTRA > gpurun_out/tlv_$tag.log 2>&1; python tools/timeline.py /tmp/tl_$tag/tl_results.db 1 8 > gpurun_out/tlv_$tag.txt 2>&1; }
EXTRA="" run default A=1
EXTRA="--no-opt-overlap" run noopt A=1
EXTRA="--resident-inputs" run resident A=1
EXTRA="" run q8 GPU_MAX_HW_QUEUES=8
EXTRA="--eager" run eager A=1
for t in default noopt resident q8 eager; do echo "== $t"; grep "step:" gpurun_out/tlv_$t.txt; done
