"""Engine sequencing (forward chain + hand-derived backward chain) checked on CPU against the oracle,
with the kernels replaced by tests/fake_ops.FakeOps (test infrastructure, see its header)."""
import pytest
import torch

import lxmert_oracle as O
from _util import golden_cfg, golden_inputs, load_golden, maxdiff
from fake_ops import FakeOps
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.engine import Engine
from xlxmert_amd.params import ParamStore


def make_engine(g, need_lang, dtype=torch.float32):
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    sd = O.make_state_dict(oc, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    store = ParamStore(cfg, "cpu", dtype, task="vis_mask" if not need_lang else "all")
    store.load_named(sd)
    eng = Engine(cfg, store, FakeOps(dtype), B, L, V, need_lang=need_lang)
    eng.sync_compute_weights()
    eng.set_inputs(inp["input_ids"], inp["attention_mask"], inp["token_type_ids"], inp["visual_pos"],
                   cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], obj_labels=inp["obj_labels"])
    return eng, oc, sd, inp


@pytest.mark.parametrize("name", ["tiny_222", "tiny_955"])
def test_forward_matches_golden(name):
    g = load_golden(name)
    eng, oc, sd, inp = make_engine(g, need_lang=True)
    lang, vis, pooled = eng.encoder_forward()
    feat, logits = eng.head_forward()
    B, L = inp["input_ids"].shape
    real = inp["attention_mask"].reshape(-1)
    assert maxdiff(lang.view(B * L, -1)[real], torch.from_numpy(g["lang"]).view(B * L, -1)[real]) < 5e-5
    assert maxdiff(vis.view(g["vis"].shape), g["vis"]) < 5e-5
    assert maxdiff(pooled, g["pooled"]) < 5e-5
    assert maxdiff(feat.view(g["feat"].shape), g["feat"]) < 5e-5
    assert maxdiff(logits.view(g["obj"].shape), g["obj"]) < 1e-4
    losses = eng.losses_forward_backward(want_grad=False)
    assert abs(losses[0].item() - g["obj_loss"].item()) < 2e-5
    assert abs(losses[1].item() - g["feat_loss"].item()) < 2e-5


def test_vis_mask_step_gradients():
    g = load_golden("tiny_222")
    eng, oc, sd, inp = make_engine(g, need_lang=False)
    losses = eng.vis_mask_forward_backward()
    assert abs(losses[0].item() - g["obj_loss"].item()) < 2e-5
    assert abs(losses[1].item() - g["feat_loss"].item()) < 2e-5
    st = eng.store
    names = [str(n) for n in g["grad_names"]]
    for k in names:
        got = st.gview(k)
        assert maxdiff(got, g["grad:" + k]) < 3e-5, k
    # tensors the reference leaves without a gradient sit outside the optimizer range
    used = {m.name for u in st.units if u.used for m in u.members}
    assert used == set(names), used ^ set(names)


def test_need_lang_engine_gives_same_vis_grads():
    """Full (lang+vis+pooled) engine with a zero language gradient == the dead-branch-eliminated engine."""
    g = load_golden("tiny_222")
    eng, oc, sd, inp = make_engine(g, need_lang=True)
    eng.encoder_forward()
    eng.head_forward()
    eng.store.grad.zero_()
    eng.losses_forward_backward(True)
    eng.GA.zero_()
    eng.head_backward(eng.GA[eng.ML:])
    eng.encoder_backward(have_lang_grad=True)
    for k in [str(n) for n in g["grad_names"]]:
        assert maxdiff(eng.store.gview(k), g["grad:" + k]) < 3e-5, k
