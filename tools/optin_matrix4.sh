cd /root/repo; O=gpurun_out/optin; mkdir -p $O
for spec in "gens2 XL_SCRATCH_GENS=2" "gens3 XL_SCRATCH_GENS=3" "duo0 XL_GEMM_DUO=0"; do
  set -- $spec; name=$1; shift
  env "$@" timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q > $O/$name.log 2>&1
  echo "$name: $(grep -E 'passed|failed|error' $O/$name.log | tail -1)"
done
