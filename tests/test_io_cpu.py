"""the reference's on-disk formats (host-side)."""
import pickle

import numpy as np
import pytest
import torch

from xlxmert_amd import io as xio


def test_checkpoint_with_ddp_prefix_roundtrip(tmp_path):
    sd = {"module.bert.pooler.dense.weight": torch.randn(4, 4), "module.mask_feat": torch.zeros(8), "epoch": torch.tensor(3)}
    p = tmp_path / "Epoch20_LXRT.pth"
    torch.save(sd, p)
    got = xio.load_state_dict(str(p))
    assert set(got) == {"bert.pooler.dense.weight", "mask_feat", "epoch"}
    assert torch.equal(got["bert.pooler.dense.weight"], sd["module.bert.pooler.dense.weight"])

    class M:
        def state_dict(self):
            return {"a.weight": torch.ones(2)}
    path = xio.save_checkpoint(M(), str(tmp_path), "BEST")
    assert path.endswith("BEST_LXRT.pth") and set(torch.load(path)) == {"module.a.weight"}


def test_centroid_file_naming_and_shape_check(tmp_path):
    name = xio.centroid_filename("resnext101", "mscoco_train", 10000, 300, 2048, 8)
    assert name == "resnext101_mscoco_train_centroids10000_iter300_d2048_grid8.npy"
    assert xio.centroid_filename("x", "y", 50, 2, 32, 4, imsize=224).endswith("_grid4_imsize224.npy")
    c = np.random.default_rng(0).random((50, 32)).astype(np.float64)
    np.save(tmp_path / "c.npy", c)
    t = xio.load_centroids(str(tmp_path / "c.npy"), 50, 32)
    assert t.dtype == torch.float32 and t.shape == (50, 32)
    with pytest.raises(ValueError):
        xio.load_centroids(str(tmp_path / "c.npy"), 10000, 2048)


def test_cluster_id_pickle(tmp_path):
    d = {"COCO_val2014_000000000042": list(range(64))}
    with open(tmp_path / "ids.pkl", "wb") as f:
        pickle.dump(d, f)
    got = xio.load_cluster_ids(str(tmp_path / "ids.pkl"))
    assert got["COCO_val2014_000000000042"].dtype == np.int64 and got["COCO_val2014_000000000042"].shape == (64,)


def test_grid_feature_h5_reader_against_a_stand_in_for_h5py(monkeypatch):
    """h5py is not part of this image, so `load_grid_features_h5` cannot meet a real file here.  What CAN run is its own logic --
    the `{img_id}/features` key layout of ref feature_extraction/coco_extract_grid_feature.py:256-260, the [g, g, F] -> [g*g, F]
    flattening of ref lxmert_data.py:288-294, the img_ids filter, the float32 conversion -- against a stand-in that offers the
    three h5py calls the reader makes (File as a context manager, keys(), path lookup).  No HDF5 parsing is tested or claimed."""
    import sys
    import types
    import numpy as np
    from xlxmert_amd import io as xio
    rng = np.random.default_rng(0)
    store = {f"{k}/features": rng.standard_normal((8, 8, 16)).astype(np.float64) for k in ("COCO_1", "COCO_2", "vg_7")}

    class _File:
        def __init__(self, path, mode="r"):
            assert path == "feats.h5" and mode == "r"
        def __enter__(self):
            return self
        def __exit__(self, *a):
            return False
        def keys(self):
            return sorted({k.split("/")[0] for k in store})
        def __getitem__(self, key):
            return store[key]

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=_File))
    out = xio.load_grid_features_h5("feats.h5")
    assert sorted(out) == ["COCO_1", "COCO_2", "vg_7"]
    for k, a in out.items():
        assert a.shape == (64, 16) and a.dtype == np.float32
        assert np.array_equal(a, store[f"{k}/features"].reshape(64, 16).astype(np.float32))       # row = y * 8 + x, as the loader flattens
    only = xio.load_grid_features_h5("feats.h5", img_ids=["vg_7"])
    assert list(only) == ["vg_7"]
    # without h5py the reader says what it needs instead of failing somewhere inside
    monkeypatch.setitem(sys.modules, "h5py", None)
    with pytest.raises(ImportError, match="h5py"):
        xio.load_grid_features_h5("feats.h5")


def test_reference_written_checkpoint_loads_key_for_key_and_reproduces_the_reference_outputs(tmp_path):
    """tests/golden/ckpt_tiny_LXRT.pth is the reference model's own `state_dict()` saved behind DDP's `module.` prefix
    (oracle/gen_golden.py::gen_ckpt, ref lxmert_pretrain.py:675-677); ckpt_tiny_io.npz holds that model's outputs.  The loader
    + parameter store must take every key (both aliases of tied tensors included), leave none over, and the oracle evaluated on
    the loaded tensors must reproduce the reference's outputs; save_checkpoint must write a file the reference's own loader
    semantics (utils.py:42-49) reads back completely."""
    import os
    import lxmert_oracle as O
    from _util import GOLDEN, golden_cfg, golden_inputs, load_golden, maxdiff
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.params import ParamStore
    g = load_golden("ckpt_tiny_io")
    path = os.path.join(GOLDEN, "ckpt_tiny_LXRT.pth")
    raw = torch.load(path)
    assert sorted(raw.keys()) == [str(k) for k in g["keys"]] and all(k.startswith("module.") for k in raw)
    sd = xio.load_state_dict(path)
    assert sd.keys() == xio.load_state_dict_reference_semantics(path).keys()
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "l_layers", "x_layers",
                                                      "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")})
    store = ParamStore(cfg, "cpu", torch.float32, task="all")
    missing = store.load_named(sd, strict=True)
    assert missing == []
    ours = store.named_state()
    assert set(ours) == set(sd), set(ours) ^ set(sd)                # key sets equal both ways, aliases included
    for k, v in sd.items():
        assert torch.equal(ours[k].cpu(), v), k
    assert torch.equal(sd["cls.predictions.decoder.weight"], sd["bert.embeddings.word_embeddings.weight"])      # tied in 4.1.1
    assert ours["cls.predictions.decoder.weight"].data_ptr() == ours["bert.embeddings.word_embeddings.weight"].data_ptr()
    assert ours["vis_emb.weight"].data_ptr() == ours["obj_predict_head.out_cluster.weight"].data_ptr()
    inp = golden_inputs(g)
    out = O.xlxmert_vis_mask_forward({k: v.clone() for k, v in ours.items()}, oc, inp["input_ids"], inp["visual_pos"],
                                     inp["attention_mask"], inp["cluster_ids"], inp["vis_mask"], inp["obj_labels"])
    assert abs(out["obj_loss"].item() - g["obj_loss"].item()) < 2e-5 and abs(out["feat_loss"].item() - g["feat_loss"].item()) < 2e-5

    class M:
        def state_dict(self):
            return ours
    back = xio.load_state_dict_reference_semantics(xio.save_checkpoint(M(), str(tmp_path), "Epoch01"))
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
