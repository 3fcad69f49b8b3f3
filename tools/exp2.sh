mkdir -p gpurun_out/r02d
for m in 0 1; do
  XL_GEMM_BN192=$m python bench.py --steps 20 --warmup 5 --gemm-table --no-cpu-baseline > gpurun_out/r02d/bench_bn$m.json 2> gpurun_out/r02d/bench_bn$m.err
  XL_GEMM_BN192=$m python tools/gemm_bench.py > gpurun_out/r02d/gemm_bench_bn$m.txt 2>&1
done
for f in gpurun_out/r02d/bench_bn*.json; do echo $f; python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["gemm_ms_per_step"], d["host_enqueue_ms_per_step"])
PY
done
paste gpurun_out/r02d/gemm_bench_bn0.txt gpurun_out/r02d/gemm_bench_bn1.txt | cut -c1-200
