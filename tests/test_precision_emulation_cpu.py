"""VERDICT r05 item 6c, answered on the CPU: how much of the bf16 path's gradient error would an fp32 RESIDUAL STREAM remove?

SURVEY section 7 specified an fp32 residual stream for the bf16 mode; the engine keeps its residual stream (pre-LayerNorm sums and
LayerNorm outputs) in bf16 like every other activation.  Building the fp32 form in the engine means another element type for every
pre-LayerNorm sum, for the LayerNorm kernels' inputs and for the residual operand of 50 epilogues per step -- so the question is
first put to the oracle: `lxmert_oracle.EMU` replays the engine's STORAGE precision on the CPU (every stored activation and activation
gradient through round-to-bf16, bf16 contraction operands, fp32 accumulation / statistics / weight gradients), once as built
("bf16") and once with the residual stream left in fp32 ("bf16_fp32res"), on the benchmarked architecture (9/5/5, d = 768, H = 12,
dff = 3072, 10k codebook; B = 4, ragged lengths = tests/golden/full_955's geometry), and every gradient tensor is compared with the
exact-fp32 oracle's."""
import pytest
import torch

import lxmert_oracle as O


def _grads(mode, oc, sd, inp):
    O.EMU["mode"] = mode
    try:
        leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight") for k, v in sd.items()}
        leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
        out = O.xlxmert_vis_mask_forward(leaf, oc, inp["input_ids"], inp["visual_pos"], inp["attention_mask"], inp["cluster_ids"],
                                         inp["vis_mask"], inp["obj_labels"])
        out["total_loss"].backward()
        return float(out["total_loss"]), {k: v.grad.double() for k, v in leaf.items() if v.grad is not None and k != "obj_predict_head.out_cluster.weight"}
    finally:
        O.EMU["mode"] = None


def _errors(g, ref):
    rel = {}
    for k, r in ref.items():
        n = r.norm().item()
        if k.endswith(".key.bias"):          # exact-zero gradient (softmax is invariant to a shift of all keys): no relative error
            continue
        if n > 1e-9:
            rel[k] = (g[k] - r).norm().item() / n
    vals = sorted(rel.values())
    worst = max(rel, key=rel.get)
    return rel, worst, rel[worst], vals[len(vals) // 2]


def test_fp32_residual_stream_would_remove_little_of_the_bf16_gradient_error():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    oc = O.OracleConfig()
    sd = O.make_state_dict(oc, 7)
    inp = O.make_inputs(oc, 11, 4, 20, 8)
    loss0, ref = _grads(None, oc, sd, inp)
    loss_b, gb = _grads("bf16", oc, sd, inp)
    loss_r, gr = _grads("bf16_fp32res", oc, sd, inp)
    rel_b, worst_b, wb, med_b = _errors(gb, ref)
    rel_r, worst_r, wr, med_r = _errors(gr, ref)
    # the same tensor in both modes, and the aggregate over all tensors
    import math
    rms_b = math.sqrt(sum(v * v for v in rel_b.values()) / len(rel_b))
    rms_r = math.sqrt(sum(v * v for v in rel_r.values()) / len(rel_r))
    print(f"loss fp32 {loss0:.5f} | bf16 storage {loss_b:.5f} | bf16 + fp32 residual stream {loss_r:.5f}")
    print(f"bf16 storage            : worst tensor {worst_b} {wb:.4f}, median {med_b:.4f}, rms {rms_b:.4f} over {len(rel_b)} tensors")
    print(f"bf16 + fp32 residual    : worst tensor {worst_r} {wr:.4f}, median {med_r:.4f}, rms {rms_r:.4f}; the bf16 mode's worst tensor here: {rel_r[worst_b]:.4f}")
    assert len(rel_b) >= 150
    # sanity of the emulation: bf16 storage costs percent-level gradient error, like the engine against the reference
    # (tests/test_engine_gpu.py: worst tensor 11 % at bs 256, median 4.1 %), and the fp32 mode is the exact oracle
    assert 1e-3 < med_b < 0.12 and wb < 0.6
    # the finding (stated, not assumed): an fp32 residual stream removes only part of the error -- most of it comes from the bf16
    # contraction operands and the other stored activations, which the precision mode leaves as they are
    assert rms_r < rms_b * 1.02              # it does not hurt ...
    assert rms_r > rms_b * 0.35              # ... and it does not remove "most" of the error (measured ratio printed above)
