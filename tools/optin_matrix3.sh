# the engine-level GPU tests under the non-default values of the remaining switches
cd /root/repo; O=gpurun_out/optin; mkdir -p $O
for spec in "gens2 XL_SCRATCH_GENS=2" "nodefer XL_DEFER_REDUCE=0" "wpair1 XL_WGRAD_PAIR=1" "noslabs XL_GEMM_SLABS=0" "nocompact XL_COMPACT_HEAD=0" "pp2 XL_GEMM_PP=2" "pp0 XL_GEMM_PP=0" "nopack XL_PACK_LANG=0" "lnplain XL_LN_BWD_DMA=0" "nofused XL_FUSED_PREDICT=0" "grouporder0 XL_GEMM_GROUP_ORDER=0" "duo0 XL_GEMM_DUO=0"; do
  set -- $spec; name=$1; shift
  env "$@" timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q > $O/$name.log 2>&1
  echo "$name: $(grep -E 'passed|failed|error' $O/$name.log | tail -1)"
done
