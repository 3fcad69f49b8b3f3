# The engine-level GPU tests (tests/test_engine_gpu.py) under the non-default value of every kernel / schedule switch: each mode must
# give the answers of the defaults.  ~5 minutes per mode on one MI355X; pass the names to run a subset.
#   bash tools/optin_matrix.sh [name ...]      -> gpurun_out/optin/<name>.log, one summary line per mode
# Expected exceptions (assertions that the DEFAULT feature is on, not numerics): nocompact (compact_head), nolastffn (the row-subset
# feed-forward block), nofused
# (fused_predict_available), wpair1 / gens2 (share of gradient ranges stored instead of accumulated).
WANT="$*"; cd "$(dirname "$0")/.."; O=gpurun_out/optin; mkdir -p $O
# the kernel-variant switches (q2, pair, pairside, persist, bn192, splitepi, relay*) exist only in the EXPERIMENTAL build (round 6): this
# script runs every mode against libxlxmert_hip_exp.so -- build it first: XL_EXPERIMENTAL=1 python -m xlxmert_amd.build
export XL_EXPERIMENTAL=1
ALL=("q2 XL_GEMM_Q=2" "pair XL_PAIR_BLOCKS=1" "pairside XL_PAIR_BLOCKS=1 XL_PAIR_SIDE=1" "persist XL_GEMM_PERSIST=1" "bn192 XL_GEMM_BN192=2"
     "splitepi XL_GEMM_SPLIT_EPI=1" "relay1 XL_GEMM_RELAY=1" "relay2 XL_GEMM_RELAY=2" "duo2 XL_GEMM_DUO=2" "duo0 XL_GEMM_DUO=0" "slabs XL_GEMM_WGRAD_SLABS=1" "gens2 XL_SCRATCH_GENS=2"
     "gens3 XL_SCRATCH_GENS=3" "nodefer XL_DEFER_REDUCE=0" "wpair1 XL_WGRAD_PAIR=1" "noslabs XL_GEMM_SLABS=0" "nocompact XL_COMPACT_HEAD=0"
     "pp2 XL_GEMM_PP=2" "pp0 XL_GEMM_PP=0" "nopack XL_PACK_LANG=0" "lnplain XL_LN_BWD_DMA=0" "nofused XL_FUSED_PREDICT=0"
     "grouporder0 XL_GEMM_GROUP_ORDER=0" "nolastffn XL_COMPACT_LAST_FFN=0" "nokeepbits XL_SDPA_KEEP_BITS=0")
for spec in "${ALL[@]}"; do
  set -- $spec; name=$1; shift
  if [ -n "$WANT" ] && ! echo " $WANT " | grep -q " $name "; then continue; fi
  env "$@" timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q > $O/$name.log 2>&1
  echo "$name ($*): $(grep -E 'passed|failed|error' $O/$name.log | tail -1)"
done
