"""On-disk formats of the reference: checkpoints, centroid files, cluster-id pickles and
grid-feature h5 files -- enough to run the path on the published artefacts.  Host-side plumbing only; nothing here touches
the GPU."""
import os
import pickle

import numpy as np
import torch


def load_state_dict(state_dict_path, loc="cpu"):
    """`*_LXRT.pth` as written by the reference (ref x-lxmert/src/utils.py:42-49, lxmert_pretrain.py:675-677): a plain
    state dict saved from a DDP-wrapped model, keys prefixed `module.`.  The prefix is stripped.  (The reference's loader
    silently DROPS keys without the prefix; this one keeps them, so single-GPU checkpoints load too.)"""
    sd = torch.load(state_dict_path, map_location=loc)
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_state_dict_reference_semantics(state_dict_path, loc="cpu"):
    """exactly ref x-lxmert/src/utils.py:42-49: keys WITHOUT the `module.` prefix are dropped (what a reference-side reader
    would see of a file; used to check that save_checkpoint's output is complete for it)."""
    sd = torch.load(state_dict_path, map_location=loc)
    return {k[len("module."):]: v for k, v in sd.items() if k.startswith("module.")}


def save_checkpoint(model, output_dir, name, ddp_prefix=True):
    """`{output}/{name}_LXRT.pth` (ref lxmert_pretrain.py:675-677); `ddp_prefix` writes the `module.` keys the reference's
    own loader insists on."""
    sd = model.state_dict()
    if ddp_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    path = os.path.join(output_dir, f"{name}_LXRT.pth")
    torch.save({k: v.detach().cpu() for k, v in sd.items()}, path)
    return path


def centroid_filename(encoder, cluster_src, n_centroids=10000, n_iter=300, feat_dim=2048, grid_size=8, imsize=None):
    """ref feature_extraction/run_kmeans.py:106-112 and lxmert_pretrain.py:72-77."""
    base = f"{encoder}_{cluster_src}_centroids{n_centroids}_iter{n_iter}_d{feat_dim}_grid{grid_size}"
    return base + (f"_imsize{imsize}" if imsize is not None else "") + ".npy"


def load_centroids(path, n_centroids=None, feat_dim=None):
    """k-means codebook `[K, F] float32` (ref lxmert_pretrain.py:75-77 -> model.set_visual_embedding)."""
    c = np.load(path)
    if c.ndim != 2 or (n_centroids is not None and c.shape[0] != n_centroids) or (feat_dim is not None and c.shape[1] != feat_dim):
        raise ValueError(f"{path}: expected [{n_centroids or 'K'}, {feat_dim or 'F'}] centroids, found {c.shape}")
    return torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))


def load_cluster_ids(path):
    """pickle `img_id -> int64[n_grids]` (ref feature_extraction/run_kmeans.py:153-165, lxmert_data.py:163-177)."""
    with open(path, "rb") as f:
        d = pickle.load(f)
    return {k: np.asarray(v, dtype=np.int64) for k, v in d.items()}


def load_grid_features_h5(path, img_ids=None):
    """h5 file `{img_id}/features [g, g, F] float32` (ref feature_extraction/coco_extract_grid_feature.py:256-260,
    lxmert_data.py:288-294) -> dict img_id -> float32 [g*g, F].  Needs h5py (not part of this image)."""
    try:
        import h5py
    except ImportError as e:
        raise ImportError("reading grid-feature h5 files needs h5py") from e
    out = {}
    with h5py.File(path, "r") as f:
        for k in (img_ids if img_ids is not None else f.keys()):
            a = np.asarray(f[f"{k}/features"], dtype=np.float32)
            out[k] = a.reshape(-1, a.shape[-1])
    return out
