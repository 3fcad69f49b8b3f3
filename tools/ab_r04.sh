# A/B of environment switches on one box: bash tools/ab_r04.sh OUT "NAME ENV=.. ENV=.." "NAME2 ..." (each run: bench.py, 60 timed steps)
out=$1; shift
mkdir -p $out
for spec in "$@"; do
  set -- $spec; name=$1; shift
  env "$@" python bench.py --steps 60 --warmup 15 --no-extra --no-cpu-baseline > $out/ab_$name.json 2> $out/ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$out/ab_$name.json"))
    print("$name", "ms", d["ms_per_step"], "p10/p50/p90", d["ms_per_step_p10"], d["ms_per_step_p50"], d["ms_per_step_p90"], "gemm_ms", d["roofline"]["gemm_ms_per_step"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("$name FAILED", e); print(open("$out/ab_$name.err").read()[-1500:])
PY
done
