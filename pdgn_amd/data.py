"""ShapeNetCore ingestion and normalisation modes of the reference (datasets_4point.py:266-380,
models/PDGNet_v2.py:415-430), batched on whatever device the clouds live on.

The reference normalises cloud by cloud in a Python loop at load time; here the five `scale_mode`s are
tensor expressions over a (S, N, 3) stack.  `ShapeNetCore` reads the same HDF5 layout
(`f[synsetid][split] -> (S, N, 3)`) -- from a path when `h5py` is installed, or from any mapping with that
shape (which is also how the tests drive it: this image has no h5py).
"""
import os
import random

import numpy as np
import torch

SCALE_MODES = (None, "global_unit", "shape_unit", "shape_bbox", "shape_half", "shape_34")

# ShapeNetCore.v2 synset ids of the 55 categories the reference knows (datasets_4point.py:238-262)
_SYNSETS = """02691156 airplane|02747177 can|02773838 bag|02801938 basket|02808440 bathtub|02818832 bed|02828884 bench|
02843684 birdhouse|02871439 bookshelf|02876657 bottle|02880940 bowl|02924116 bus|02933112 cabinet|02942699 camera|
02946921 tin_can|02954340 cap|02958343 car|02992529 cellphone|03001627 chair|03046257 clock|03085013 keyboard|
03207941 dishwasher|03211117 monitor|03261776 earphone|03325088 faucet|03337140 file|03467517 guitar|03513137 helmet|
03593526 jar|03624134 knife|03636649 lamp|03642806 laptop|03691459 speaker|03710193 mailbox|03759954 microphone|
03761084 microwave|03790512 motorcycle|03797390 mug|03928116 piano|03938244 pillow|03948459 pistol|03991062 pot|
04004475 printer|04074963 remote_control|04090263 rifle|04099429 rocket|04225987 skateboard|04256520 sofa|
04330267 stove|04379243 table|04401088 telephone|04460130 tower|04468005 train|04530566 vessel|04554684 washer"""
synsetid_to_cate = dict(item.split() for item in _SYNSETS.replace("\n", "").split("|"))
cate_to_synsetid = {v: k for k, v in synsetid_to_cate.items()}


def dataset_statistics(all_points):
    """get_statistics (:289-316): per-axis mean over every point, one std over every coordinate."""
    B, N, _ = all_points.shape
    return {"mean": all_points.reshape(B * N, -1).mean(dim=0), "std": all_points.reshape(-1).std(dim=0)}


def normalize_clouds(pcs, mode, global_std=None):
    """(S, N, 3) -> (normalised clouds, shift (S,1,3), scale (S,1,1)) for a `scale_mode` of :326-348
    (`shape_unit` / `shape_bbox` are also the two modes of PDGNet_v2.normalize_point_clouds :415-430)."""
    if mode not in SCALE_MODES:
        raise ValueError("unknown scale_mode %r" % (mode,))
    S = pcs.shape[0]
    if mode is None:
        shift = torch.zeros(S, 1, 3, dtype=pcs.dtype, device=pcs.device)
        scale = torch.ones(S, 1, 1, dtype=pcs.dtype, device=pcs.device)
    elif mode == "shape_bbox":
        pc_max, pc_min = pcs.max(dim=1, keepdim=True)[0], pcs.min(dim=1, keepdim=True)[0]
        shift = (pc_min + pc_max) / 2
        scale = (pc_max - pc_min).max(dim=2, keepdim=True)[0] / 2
    else:
        shift = pcs.mean(dim=1, keepdim=True)
        if mode == "global_unit":
            if global_std is None:
                raise ValueError("global_unit needs the data set's std (dataset_statistics)")
            scale = torch.as_tensor(global_std, dtype=pcs.dtype, device=pcs.device).reshape(1, 1, 1).expand(S, 1, 1)
        else:
            scale = pcs.reshape(S, -1).std(dim=1).view(S, 1, 1)
            if mode == "shape_half":
                scale = scale / 0.5
            elif mode == "shape_34":
                scale = scale / 0.75
    return (pcs - shift) / scale, shift, scale


def normalize_point_clouds(pcs, mode):
    """PDGNet_v2.normalize_point_clouds (:415-430) for a whole (S,N,3) stack (mode None: unchanged)."""
    if mode is None:
        return pcs
    if mode not in ("shape_unit", "shape_bbox"):
        raise ValueError("the test phase normalises with shape_unit or shape_bbox, got %r" % (mode,))
    return normalize_clouds(pcs, mode)[0]


def multires_sample(pcs, sizes=(256, 512, 1024), generator=None):
    """The three sub-resolutions of __getitem__ (:372-379): indices drawn WITH replacement, one draw per cloud."""
    S, N, _ = pcs.shape
    out = []
    for r in sizes:
        sel = torch.randint(0, N, (S, r), generator=generator, device=pcs.device if generator is None else generator.device)
        out.append(torch.gather(pcs, 1, sel.to(pcs.device).unsqueeze(2).expand(S, r, 3)))
    return out


def _open(source):
    if not isinstance(source, (str, bytes, os.PathLike)):
        return source, None
    try:
        import h5py
    except ImportError as e:                                    # pragma: no cover - h5py absent in this image
        raise ImportError("reading %r needs h5py; pass a {synsetid: {split: array}} mapping instead" % (source,)) from e
    f = h5py.File(source, "r")
    return f, f


class ShapeNetCore(torch.utils.data.Dataset):
    """datasets_4point.ShapeNetCore (:266-380): same constructor arguments, same per-item tuple
    (256 / 512 / 1024 resampled points, the full cloud, the category name), same deterministic shuffle."""

    GRAVITATIONAL_AXIS = 1

    def __init__(self, cates_list, split, scale_mode, path, transform=None):
        super().__init__()
        cates = [cates_list] if isinstance(cates_list, str) else list(cates_list)
        assert split in ("train", "val", "test")
        assert scale_mode in SCALE_MODES
        if "all" in cates:
            cates = list(cate_to_synsetid.keys())
        self.cate_synsetids = sorted(cate_to_synsetid[s] for s in cates)
        self.split, self.scale_mode, self.transform, self.path = split, scale_mode, transform, path
        f, closer = _open(path)
        try:
            every = [torch.as_tensor(np.asarray(f[sid][sp])) for sid in self.cate_synsetids for sp in ("train", "val", "test")]
            self.stats = dataset_statistics(torch.cat(every, dim=0))
            self.pointclouds = []
            for sid in self.cate_synsetids:
                pcs = torch.as_tensor(np.asarray(f[sid][split]))
                norm, shift, scale = normalize_clouds(pcs, scale_mode, self.stats["std"])
                for j in range(pcs.shape[0]):
                    self.pointclouds.append({"pointcloud": norm[j], "cate": synsetid_to_cate[sid], "id": j,
                                             "shift": shift[j], "scale": scale[j]})
        finally:
            if closer is not None:
                closer.close()
        self.pointclouds.sort(key=lambda d: d["id"])
        random.Random(2020).shuffle(self.pointclouds)           # the reference's deterministic shuffle (:362-363)

    def __len__(self):
        return len(self.pointclouds)

    def __getitem__(self, idx):
        data = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in self.pointclouds[idx].items()}
        if self.transform is not None:
            data = self.transform(data)
        pc = data["pointcloud"]
        subs = [pc[np.random.choice(pc.shape[0], r), :].float() for r in (256, 512, 1024)]
        return subs[0], subs[1], subs[2], pc, data["cate"]

    def stack(self, device=None):
        """All clouds of the split as one (S,N,3) tensor in data-set order (the test phase's `ref_pcs`, :293-298)."""
        pcs = torch.stack([d["pointcloud"] for d in self.pointclouds], 0)
        return pcs.to(device) if device is not None else pcs
