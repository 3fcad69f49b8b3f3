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
