"""CT slice I/O and the normalisation contract on either side of the sampling path.

Reference contract:
  * slices are `.npy` arrays in HU + 1024; `Normalize`: x = clip((m - 1024 + 1000) / 3000, 0, 1)
    (/root/reference/data/transforms.py:572-587), shaped (1, H, W) float32;
  * a dataset item is the pair [NDCT (target), LDCT (condition)]  (data/pdf_dataset.py:466);
  * results are saved as np.save(<name[:-4]>, out.reshape(H, W)) in [0,1]  (src/DADiff.py:1913-1914);
  * previews use the HU window clip(x*3000-1000, -160, 240), (x+160)/400  (src/DADiff.py:1794-1795).
The reference's PDFDataset globs hard-coded private paths; here the file lists are arguments.
"""
import os

import numpy as np
import torch


def normalize_hu(m, min_value=-1000.0, max_value=2000.0):
    m = np.asarray(m, dtype=np.float32) - 1024.0
    return np.clip((m - min_value) / (max_value - min_value), 0.0, 1.0).astype(np.float32)


def hu_window(x01, lo=-160.0, hi=240.0):
    """[0,1]-normalised image -> display window in [0,1] (torch tensor or ndarray)."""
    if isinstance(x01, torch.Tensor):
        return (torch.clip(x01 * 3000 - 1000, lo, hi) - lo) / (hi - lo)
    return (np.clip(x01 * 3000 - 1000, lo, hi) - lo) / (hi - lo)


class CTSliceDataset(torch.utils.data.Dataset):
    """Pairs of (low-dose, normal-dose) `.npy` slices -> [NDCT, LDCT] tensors (1,H,W) in [0,1]."""

    def __init__(self, ldct_paths, ndct_paths):
        assert len(ldct_paths) == len(ndct_paths)
        self.q_path_list, self.f_path_list = list(ldct_paths), list(ndct_paths)

    def __len__(self):
        return len(self.q_path_list)

    def __getitem__(self, i):
        a = normalize_hu(np.load(self.q_path_list[i]))[None]
        b = normalize_hu(np.load(self.f_path_list[i]))[None]
        return [torch.from_numpy(b), torch.from_numpy(a)]

    def load_name(self, index, sub_dir=False):
        return os.path.basename(self.q_path_list[index])


class SyntheticCTDataset(torch.utils.data.Dataset):
    """Seeded phantoms (founddiff_amd.synth.ct_phantom) with the same item layout."""

    def __init__(self, n, size, seed=10):
        from .synth import ct_phantom
        self.nd, self.ld = ct_phantom(n, size, seed)

    def __len__(self):
        return self.nd.shape[0]

    def __getitem__(self, i):
        return [torch.from_numpy(self.nd[i]), torch.from_numpy(self.ld[i])]

    def load_name(self, index, sub_dir=False):
        return f"synthetic-quarter-{index:04d}.npy"
