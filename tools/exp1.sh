mkdir -p gpurun_out/r02b
for mt in 48 100 200 300; do
  XL_GEMM_PP_MIN_TILES=$mt python bench.py --steps 20 --warmup 5 --gemm-table --no-cpu-baseline > gpurun_out/r02b/bench_mt$mt.json 2> gpurun_out/r02b/bench_mt$mt.err
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --single-stream > gpurun_out/r02b/bench_single.json 2> gpurun_out/r02b/bench_single.err
for f in gpurun_out/r02b/*.json; do echo $f; python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["gemm_ms_per_step"], d["host_enqueue_ms_per_step"])
PY
done
