# the engine-level GPU tests under every opt-in kernel / schedule switch (each must give the same answers as the defaults)
cd /root/repo; O=gpurun_out/optin; mkdir -p $O
for spec in "q2 XL_GEMM_Q=2" "pair XL_PAIR_BLOCKS=1" "pairside XL_PAIR_BLOCKS=1 XL_PAIR_SIDE=1" "persist XL_GEMM_PERSIST=1" "bn192 XL_GEMM_BN192=2" "splitepi XL_GEMM_SPLIT_EPI=1" "duo2 XL_GEMM_DUO=2" "slabs XL_GEMM_WGRAD_SLABS=1"; do
  set -- $spec; name=$1; shift
  env "$@" python -m pytest tests/test_engine_gpu.py -m gpu -x -q > $O/$name.log 2>&1
  echo "$name: $(grep -E 'passed|failed|error' $O/$name.log | tail -1)"
done
