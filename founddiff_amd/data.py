"""CT slice I/O, the normalisation contract and the mixed-dose datasets on either side of the sampling path.

Reference contract:
  * slices are `.npy` arrays in HU + 1024, stored (1, H, W) (`ToTensor(expand_dims=False)` asserts ndim 3,
    /root/reference/data/transforms.py:618-634); `Normalize`: x = clip((m - 1024 + 1000) / 3000, 0, 1)
    (data/transforms.py:572-587), float32;
  * a test item is the pair [NDCT (target), LDCT (condition)]  (data/pdf_dataset.py:424-466): the low-dose file
    `<loc>-...-<dose>-<index>.npy` names its anatomy (`loc` = head | lung | ab) and the index of its full-dose
    partner in that anatomy's NDCT list; the last '-' token of both file names must agree;
  * dose labels (data/pdf_dataset.py:480-510, data/dose_dataset.py:101-129): full_1mm -> 1, quarter_1mm -> 4
    (10 for lung in the test set), simulated dose fraction 0.5 / 0.33 / 0.25 / 0.20 / 0.17 / 0.12 / 0.10 / 0.05 ->
    2 / 3 / 4 / 5 / 6 / 8 / 10 / 20;
  * results are saved as np.save(<name[:-4]>, out.reshape(H, W)) in [0,1]  (src/DADiff.py:1913-1914);
  * previews use the HU window clip(x*3000-1000, -160, 240), (x+160)/400  (src/DADiff.py:1794-1795) and
    torchvision's save_image grid.
The reference's datasets glob hard-coded private paths inside their constructors; here the file lists are
arguments, everything else (pairing, labels, names, normalisation) follows the reference.
"""
import os

import numpy as np
import torch

DOSE_LABELS = {0.5: 2, 0.33: 3, 0.25: 4, 0.20: 5, 0.17: 6, 0.12: 8, 0.10: 10, 0.05: 20}


def normalize_hu(m, min_value=-1000.0, max_value=2000.0):
    """transforms.Normalize (data/transforms.py:572-587): stored value - 1024 -> [0, 1]."""
    m = np.asarray(m, dtype=np.float32) - 1024.0
    return np.clip((m - min_value) / (max_value - min_value), 0.0, 1.0).astype(np.float32)


def to_tensor(m, expand_dims=False):
    """transforms.ToTensor (data/transforms.py:618-634)."""
    assert m.ndim in (3, 4), "Supports only 3D (DxHxW) or 4D (CxDxHxW) images"
    if expand_dims and m.ndim == 3:
        m = np.expand_dims(m, axis=0)
    return torch.from_numpy(m.astype(np.float32))


def load_slice(path):
    """One stored slice -> (1, H, W) float32 tensor in [0, 1].  Accepts the reference's (1, H, W) arrays and
    plain (H, W) ones."""
    m = np.load(path).astype(np.float32)
    if m.ndim == 2:
        m = m[None]
    return to_tensor(normalize_hu(m))


def save_slice(path, x01):
    """The inverse storage rule (HU + 1024, (1, H, W) float32) -- used to write test volumes."""
    x = np.asarray(x01, dtype=np.float32).reshape((1,) + tuple(np.asarray(x01).shape[-2:]))
    np.save(path, x * 3000.0 - 1000.0 + 1024.0)


def hu_window(x01, lo=-160.0, hi=240.0):
    """[0,1]-normalised image -> display window in [0,1] (torch tensor or ndarray)."""
    if isinstance(x01, torch.Tensor):
        return (torch.clip(x01 * 3000 - 1000, lo, hi) - lo) / (hi - lo)
    return (np.clip(x01 * 3000 - 1000, lo, hi) - lo) / (hi - lo)


def define_label(path, lung_quarter_is_10=True):
    """Dose label of a file path.  `lung_quarter_is_10`: the test dataset's rule (data/pdf_dataset.py:486-490);
    False gives the Dose-CLIP training dataset's (data/dose_dataset.py:106-108).  That dataset has no branch for
    the 0.25 fraction -- its `label` stays unbound there -- so the same path raises here too."""
    if "full_1mm" in path:
        return 1
    if "quarter_1mm" in path:
        return 10 if (lung_quarter_is_10 and "lung" in path) else 4
    dose = float(path.split("-")[-2])
    if dose == 0.25 and not lung_quarter_is_10:
        raise UnboundLocalError("local variable 'label' referenced before assignment "
                                "(data/dose_dataset.py:110-126 has no branch for dose 0.25)")
    if dose not in DOSE_LABELS:
        raise UnboundLocalError(f"no dose label for fraction {dose} in {path!r}")
    return DOSE_LABELS[dose]


def make_grid(t, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid for a (B, C, H, W) tensor (the reference's preview layout)."""
    t = t.detach().float().cpu()
    if t.dim() == 3:
        t = t[None]
    if t.shape[1] == 1:
        t = t.repeat(1, 3, 1, 1)
    B, C, H, W = t.shape
    xmaps = min(nrow, B)
    ymaps = (B + xmaps - 1) // xmaps
    hh, ww = H + padding, W + padding
    grid = t.new_full((C, hh * ymaps + padding, ww * xmaps + padding), pad_value)
    for k in range(B):
        y, x = divmod(k, xmaps)
        grid[:, y * hh + padding:y * hh + padding + H, x * ww + padding:x * ww + padding + W] = t[k]
    return grid


def save_image(t, path, nrow=8):
    """torchvision.utils.save_image: grid -> clamp [0,1] -> *255 + 0.5 -> uint8 PNG (src/DADiff.py:1812)."""
    from PIL import Image
    g = make_grid(t, nrow=nrow)
    arr = g.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(path)


class CTSliceDataset(torch.utils.data.Dataset):
    """Pairs of (low-dose, normal-dose) `.npy` slices -> [NDCT, LDCT] tensors (1,H,W) in [0,1]."""

    def __init__(self, ldct_paths, ndct_paths):
        assert len(ldct_paths) == len(ndct_paths)
        self.q_path_list, self.f_path_list = list(ldct_paths), list(ndct_paths)

    def __len__(self):
        return len(self.q_path_list)

    def __getitem__(self, i):
        return [load_slice(self.f_path_list[i]), load_slice(self.q_path_list[i])]

    def load_name(self, index, sub_dir=False):
        return os.path.basename(self.q_path_list[index])


class MixedDoseTestDataset(torch.utils.data.Dataset):
    """The reference's test dataset (data/pdf_dataset.py:424-510): a flat list of low-dose slices of several
    anatomies and dose levels, each paired with its full-dose partner through the file name.

    q_paths: low-dose files `<loc>-<...>-<dose>-<index>.npy` (loc = head | lung | ab);
    ndct_paths: {loc: sorted list of that anatomy's full-dose files}."""

    def __init__(self, q_paths, ndct_paths):
        self.q_path_list = list(q_paths)
        self.ndct = {k: list(v) for k, v in ndct_paths.items()}
        self.A_size = len(self.q_path_list)
        self.dataset_size = [len(self.ndct.get(k, [])) for k in ("ab", "lung", "head")]

    def __len__(self):
        return self.A_size

    def partner(self, index):
        path = self.q_path_list[index]
        loc = path.split("/")[-1].split("-")[0]
        ndct_index = int(path.split(".")[-2].split("-")[-1])
        if loc not in self.ndct:
            raise KeyError(f"no full-dose list for anatomy {loc!r} ({path})")
        f = self.ndct[loc][ndct_index]
        assert f.split("-")[-1] == path.split("-")[-1], (f, path)
        return f

    def __getitem__(self, index):
        return [load_slice(self.partner(index)), load_slice(self.q_path_list[index])]

    def load_name(self, index, sub_dir=False):
        name = self.q_path_list[index]
        if sub_dir == 0:
            return os.path.basename(name)
        return os.path.dirname(name).split("/")[-1] + "_" + os.path.basename(name)

    def define_label(self, path):
        return define_label(path, lung_quarter_is_10=True)

    def dose(self, index):
        """1 / label: the dose fraction feature of data/pdf_dataset.py:452."""
        return 1.0 / self.define_label(self.q_path_list[index])


class DoseDataset(torch.utils.data.Dataset):
    """The Dose-CLIP dataset's item layout and label scheme (data/dose_dataset.py:80-129): every slice of every
    dose level (full-dose first, then 1/2 ... 1/20), item = ([x, x], label) with x (1,H,W) in [0,1] and label the
    dose denominator as float32.  `images_list`: the concatenated per-dose file lists."""

    def __init__(self, images_list):
        self.images_list = list(images_list)

    def __len__(self):
        return len(self.images_list)

    def define_label(self, path):
        return define_label(path, lung_quarter_is_10=False)

    def __getitem__(self, index):
        path = self.images_list[index]
        x = load_slice(path)
        return [x, x.clone()], np.asarray(self.define_label(path)).astype(np.float32)


def shard_indices(n, world, rank):
    """Contiguous block of the item range for this rank (founddiff_amd.parallel.shard_range): how a mixed-dose
    volume is split over the GPUs of a node."""
    from .parallel import shard_range
    lo, hi = shard_range(n, world, rank)
    return list(range(lo, hi))


class SyntheticCTDataset(torch.utils.data.Dataset):
    """Seeded phantoms (founddiff_amd.synth.ct_phantom) with the same item layout; five noise levels cycle
    through the items like a mixed-dose list."""

    def __init__(self, n, size, seed=10):
        from .synth import ct_phantom
        self.nd, self.ld = ct_phantom(n, size, seed)

    def __len__(self):
        return self.nd.shape[0]

    def __getitem__(self, i):
        return [torch.from_numpy(self.nd[i]), torch.from_numpy(self.ld[i])]

    def load_name(self, index, sub_dir=False):
        return f"synthetic-quarter-{index:04d}.npy"
