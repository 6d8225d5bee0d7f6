"""CPU tests of the data contract on either side of the sampling path (founddiff_amd/data.py):
storage / normalisation round trip, the test dataset's NDCT pairing and names, both dose-label schemes, the
preview grid writer and rank sharding of a mixed-dose list."""
import os

import numpy as np
import pytest
import torch


def _write(path, seed, shape=(1, 16, 16)):
    rng = np.random.RandomState(seed)
    hu = rng.uniform(-1200, 2400, size=shape).astype(np.float32)       # beyond the clip range on both sides
    np.save(path, hu + 1024.0)
    return hu


def test_normalisation_and_npy_round_trip(tmp_path):
    from founddiff_amd import data
    hu = _write(tmp_path / "a.npy", 0)
    x = data.load_slice(str(tmp_path / "a.npy"))
    assert x.shape == (1, 16, 16) and x.dtype == torch.float32
    ref = np.clip((hu + 1000.0) / 3000.0, 0, 1)                        # data/transforms.py:582-587
    assert np.allclose(x.numpy(), ref, atol=1e-6)
    assert float(x.min()) == 0.0 and float(x.max()) == 1.0
    # 2-D arrays are accepted too; save_slice is the inverse of load_slice inside the clip range
    np.save(tmp_path / "b.npy", (hu[0] + 1024.0))
    assert torch.equal(data.load_slice(str(tmp_path / "b.npy")), x)
    data.save_slice(str(tmp_path / "c.npy"), x.numpy())
    assert np.allclose(data.load_slice(str(tmp_path / "c.npy")).numpy(), x.numpy(), atol=1e-6)
    # Trainer.test's output rule: np.save(name[:-4], out.reshape(H, W)) in [0, 1]
    np.save(str(tmp_path / "ab-sim-0.25-0003.npy")[:-4], x.numpy().reshape(16, 16))
    assert np.load(tmp_path / "ab-sim-0.25-0003.npy").shape == (16, 16)
    # HU display window of the previews (src/DADiff.py:1794-1795)
    w = data.hu_window(torch.tensor([0.0, (40 + 1000) / 3000.0, 1.0]))
    assert torch.allclose(w, torch.tensor([0.0, 0.5, 1.0]))
    with pytest.raises(AssertionError):
        data.to_tensor(np.zeros((4, 4)))                               # ToTensor asserts ndim in (3, 4)


def test_mixed_dose_test_dataset_pairing_names_labels(tmp_path):
    from founddiff_amd import data
    nd = {"ab": [], "head": []}
    for loc in nd:
        d = tmp_path / f"Mayo2020_{loc}_2d" / "test" / "full_1mm"
        d.mkdir(parents=True)
        for i in range(3):
            p = d / f"{loc}-full_1mm-{i:04d}.npy"
            _write(p, 100 + i)
            nd[loc].append(str(p))
    q = []
    for loc, dose in (("ab", "0.25"), ("ab", "0.10"), ("head", "0.05")):
        d = tmp_path / f"Mayo2020_{loc}_2d" / "test" / f"sim-{dose}"
        d.mkdir(parents=True)
        for i in (2, 0):
            p = d / f"{loc}-sim-{dose}-{i:04d}.npy"
            _write(p, 200 + i)
            q.append(str(p))
    ds = data.MixedDoseTestDataset(q, nd)
    assert len(ds) == 6 and ds.dataset_size == [3, 0, 3]
    y, x = ds[0]                                                       # [NDCT, LDCT] (data/pdf_dataset.py:466)
    assert torch.equal(y, data.load_slice(nd["ab"][2])) and torch.equal(x, data.load_slice(q[0]))
    assert ds.partner(5) == nd["head"][0]
    assert ds.load_name(0) == "ab-sim-0.25-0002.npy"
    assert ds.load_name(0, sub_dir=1) == "sim-0.25_ab-sim-0.25-0002.npy"
    assert [ds.define_label(p) for p in (q[0], q[2], q[4])] == [4, 10, 20]
    assert ds.dose(2) == pytest.approx(0.1)
    with pytest.raises(KeyError):
        data.MixedDoseTestDataset(["/x/lung-sim-0.50-0001.npy"], nd).partner(0)
    # label schemes: test set (pdf_dataset.py:480-510) vs Dose-CLIP set (dose_dataset.py:101-129)
    assert data.define_label("/d/Mayo2020_lung_2d/test/quarter_1mm/x-0001.npy") == 10
    assert data.define_label("/d/Mayo2020_lung_2d/test/quarter_1mm/x-0001.npy", lung_quarter_is_10=False) == 4
    assert data.define_label("/d/full_1mm/ab-full_1mm-0001.npy") == 1
    for frac, lab in ((0.5, 2), (0.33, 3), (0.2, 5), (0.17, 6), (0.12, 8), (0.1, 10), (0.05, 20)):
        assert data.define_label(f"/d/sim/ab-sim-{frac:.2f}-0001.npy", lung_quarter_is_10=False) == lab
    with pytest.raises(UnboundLocalError):                             # the training set's scheme has no 0.25 branch
        data.define_label("/d/sim/ab-sim-0.25-0001.npy", lung_quarter_is_10=False)
    dd = data.DoseDataset([nd["ab"][0], q[2]])
    (a, b), lab = dd[1]
    assert torch.equal(a, b) and lab.dtype == np.float32 and float(lab) == 10.0 and float(dd[0][1]) == 1.0


def test_preview_grid_and_png(tmp_path):
    from PIL import Image
    from founddiff_amd import data
    t = torch.rand(5, 1, 6, 4)
    g = data.make_grid(t, nrow=2)                                      # torchvision layout: 3 rows x 2 columns, padding 2
    assert g.shape == (3, 3 * 8 + 2, 2 * 6 + 2)
    assert torch.equal(g[0, 2:8, 2:6], t[0, 0]) and torch.equal(g[1, 10:16, 8:12], t[3, 0])
    assert float(g[:, 0].abs().max()) == 0.0
    data.save_image(t, str(tmp_path / "p.png"), nrow=2)
    im = np.asarray(Image.open(tmp_path / "p.png"))
    assert im.shape == (26, 14, 3) and im.dtype == np.uint8
    assert np.array_equal(im[2:8, 2:6, 0], (t[0, 0] * 255 + 0.5).clamp(0, 255).to(torch.uint8).numpy())


def test_shard_indices_cover_mixed_dose_list():
    from founddiff_amd import data
    for n in (0, 5, 64):
        for w in (1, 2, 8):
            got = sum((data.shard_indices(n, w, r) for r in range(w)), [])
            assert got == list(range(n))


def test_trainer_name_rule_and_group_means():
    """src/DADiff.py:1904-1907 (png name) and 1918-1952 (per-anatomy / per-dose means) -- host logic only."""
    import logging
    from founddiff_amd.DADiff import Trainer
    assert Trainer.image_file_name("L067-quarter_1mm-0012.npy") == "L067-quarter_1mm-0012.png"
    assert Trainer.image_file_name("ab-sim-0.25-0012.npy") == "ab-sim-0.25-0012.png"
    tr = Trainer.__new__(Trainer)
    tr.test_groups = (("ab", 2), ("lung", 3), ("head", 1))
    n = (2 + 3 + 1) * 4
    tr.test_running_psnr = list(np.arange(n, dtype=np.float32))
    tr.test_running_ssim = list(np.arange(n, dtype=np.float32) / 100)
    tr.test_running_rmse = list(np.arange(n, dtype=np.float32) / 1000)
    lines = []
    tr.train_logger = logging.getLogger("t_groups")
    tr.train_logger.setLevel(logging.INFO)
    h = logging.Handler()
    h.emit = lambda rec: lines.append(rec.getMessage())
    tr.train_logger.addHandler(h)
    g = tr._log_groups()
    assert g["ab"]["mean"][0] == pytest.approx(np.mean(np.arange(0, 8)))
    assert g["lung"]["mean"][0] == pytest.approx(np.mean(np.arange(8, 20)))
    assert g["head"]["mean"][0] == pytest.approx(np.mean(np.arange(20, 24)))       # taken from the END of the list
    assert [d[0] for d in g["lung"]["dose"]] == pytest.approx([9.0, 12.0, 15.0, 18.0])
    assert len(lines) == 3 * 5 and lines[0].startswith("(ab average mean: psnr: 3.5000")
    assert "dose:  2,average: psnr: 15.0000" in lines[8]
    # a list shorter than the groups: empty slices give nan like np.mean([]) does in the reference
    tr.test_running_psnr = tr.test_running_psnr[:4]
    tr.test_running_ssim, tr.test_running_rmse = tr.test_running_ssim[:4], tr.test_running_rmse[:4]
    assert np.isnan(tr._log_groups()["lung"]["mean"][0])
