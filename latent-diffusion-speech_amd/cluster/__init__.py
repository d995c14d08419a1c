"""K-means semantic codebook helpers with the reference's names and behaviour (reference cluster/__init__.py:5-27):
`semantic_codebook.pt` is a torch-saved dict {'n_features_in_', '_n_threads', 'cluster_centers_'} that is poured into a
scikit-learn KMeans object.  The token -> unit-embedding lookup of the TTS path (22_infer_tts.py:43-52,106) is a row gather
of `cluster_centers_`; `codebook_to_device` / lds.native.gather_rows run it in liblds."""
import numpy as np
import torch


def get_cluster_model(ckpt_path):
    from sklearn.cluster import KMeans
    checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=False)     # numpy arrays inside: not a weights-only file
    km = KMeans(checkpoint["n_features_in_"])
    for key in ("n_features_in_", "_n_threads", "cluster_centers_"):
        km.__dict__[key] = checkpoint[key]
    return km


def get_cluster_result(model, x):
    """x: np.array [t, dim] -> cluster ids [t]"""
    return model.predict(x)


def get_cluster_center_result(model, x):
    """x: np.array [t, dim] -> the centre of each frame's cluster [t, dim]"""
    return model.cluster_centers_[model.predict(x)]


def get_center(model, token):
    return model.cluster_centers_[token]


def codebook_to_device(model_or_centers, device):
    """cluster centres as a contiguous fp32 device tensor [n_codes, dim] for lds.native.gather_rows"""
    centers = getattr(model_or_centers, "cluster_centers_", model_or_centers)
    return torch.from_numpy(np.ascontiguousarray(centers, dtype=np.float32)).to(device)
