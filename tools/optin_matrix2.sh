cd /root/repo; O=gpurun_out/optin; mkdir -p $O
python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "paired or pair" > $O/new_pair_tests.log 2>&1; echo "paired tests (default env): $(grep -E 'passed|failed|error' $O/new_pair_tests.log | tail -1)"
for spec in "pair XL_PAIR_BLOCKS=1" "splitepi XL_GEMM_SPLIT_EPI=1"; do
  set -- $spec; name=$1; shift
  env "$@" python -m pytest tests/test_engine_gpu.py -m gpu -x -q > $O/$name.log 2>&1
  echo "$name: $(grep -E 'passed|failed|error' $O/$name.log | tail -1)"
done
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "split_k_with_epilogue" > $O/splitk.log 2>&1; echo "split tests: $(grep -E 'passed|failed|error' $O/splitk.log | tail -1)"
