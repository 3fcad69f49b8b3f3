mkdir -p gpurun_out/r04a
run() { name=$1; shift; env "$@" python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline > gpurun_out/r04a/env_$name.json 2>gpurun_out/r04a/env_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r04a/env_$name.json"))
print("$name", d["ms_per_step"], d["ms_per_step_p50"], "gemm_ms", d["roofline"]["gemm_ms_per_step"], "lang", d["roofline"]["blocks"]["language_layers"])
PY
}
run base XL_DUMMY=1
run pp_lang XL_GEMM_DUO=0 XL_GEMM_PP_MIN_TILES=32
run mfma128_lang XL_GEMM_DUO=0
run base2 XL_DUMMY=1
