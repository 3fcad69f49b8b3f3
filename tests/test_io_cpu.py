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
