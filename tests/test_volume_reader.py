"""Volume / slice / registration readers over NRRD data (reference dataset/few_shot_reader.py:232-650).
CPU: rpnet_amd.utils.nrrd against hand-built files and round trips; FewshotVolumeReader / FewshotSliceReader (eval mode,
no registration) bit-exact against what the reference's own classes returned on the same synthetic NRRD data set
(tests/golden/volume_reader.npz, gen_golden_reader.py); the train-mode item contract.
GPU: FewshotRegReader — the item test_rpnet.py consumes, with the HIP registration pre-step — against the reference's
item (CPU registration)."""
import gzip
import os
import random

import numpy as np
import pytest
import torch

from rpnet_amd.utils import nrrd
from rpnet_amd.utils import volume_reader as VR
from tests.reader_cases import CASES, config_for


# ------------------------------------------------------------------------------------------------------------ NRRD
def test_nrrd_hand_built_raw_big_endian(tmp_path):
    """first header axis is the fastest in the file; shape == sizes; big-endian payload; comments, key/value pairs"""
    vals = np.arange(2 * 3 * 4, dtype=">i2")
    p = tmp_path / "a.nrrd"
    p.write_bytes(b"NRRD0004\n# a comment\ntype: short\ndimension: 3\nspace: left-posterior-superior\nsizes: 2 3 4\n"
                  b"endian: big\nencoding: raw\nmodality:=CT\n\n" + vals.tobytes())
    data, hdr = nrrd.read(str(p))
    assert data.shape == (2, 3, 4) and data.dtype == np.int16
    assert data[1, 0, 0] == 1 and data[0, 1, 0] == 2 and data[0, 0, 1] == 6 and data[1, 2, 3] == 23
    assert hdr["dimension"] == 3 and hdr["sizes"].tolist() == [2, 3, 4] and hdr["modality"] == "CT"
    assert hdr["space"] == "left-posterior-superior"


def test_nrrd_gzip_ascii_detached_and_byte_skip(tmp_path):
    vals = np.linspace(-3, 3, 12, dtype="<f4")
    (tmp_path / "g.nrrd").write_bytes(b"NRRD0001\ntype: float\ndimension: 2\nsizes: 4 3\nendian: little\nencoding: gz\n\n"
                                      + gzip.compress(vals.tobytes()))
    g, _ = nrrd.read(str(tmp_path / "g.nrrd"))
    assert g.shape == (4, 3) and np.array_equal(g.reshape(-1, order="F"), vals)
    (tmp_path / "t.nrrd").write_bytes(b"NRRD0004\ntype: uchar\ndimension: 2\nsizes: 3 2\nencoding: ascii\n\n1 2 3\n4 5 6\n")
    t, _ = nrrd.read(str(tmp_path / "t.nrrd"))
    assert t.dtype == np.uint8 and t.tolist() == [[1, 4], [2, 5], [3, 6]]
    (tmp_path / "d.raw").write_bytes(b"JUNKJUNK" + np.arange(6, dtype="<u2").tobytes())
    (tmp_path / "d.nhdr").write_bytes(b"NRRD0004\ntype: ushort\ndimension: 1\nsizes: 6\nendian: little\nencoding: raw\n"
                                      b"byte skip: -1\ndata file: d.raw\n")
    d, _ = nrrd.read(str(tmp_path / "d.nhdr"))
    assert d.tolist() == [0, 1, 2, 3, 4, 5]


@pytest.mark.parametrize("enc", ["raw", "gzip", "bzip2"])
@pytest.mark.parametrize("dt", [np.uint8, np.int16, np.float32, np.float64, np.int64])
def test_nrrd_round_trip(tmp_path, enc, dt):
    a = (np.random.default_rng(3).normal(0, 50, (5, 7, 3))).astype(dt)
    p = str(tmp_path / "r.nrrd")
    nrrd.write(p, a, header={"space directions": "(1,0,0) (0,1,0) (0,0,2.5)"}, encoding=enc)
    b, hdr = nrrd.read(p)
    assert b.dtype == a.dtype and np.array_equal(a, b) and hdr["space directions"].endswith("(0,0,2.5)")


def test_nrrd_errors(tmp_path):
    p = tmp_path / "x.nrrd"
    p.write_bytes(b"NOTNRRD\n")
    with pytest.raises(nrrd.NRRDError):
        nrrd.read(str(p))
    p.write_bytes(b"NRRD0004\ntype: short\ndimension: 2\nsizes: 4 4\nencoding: raw\n\n" + b"\0" * 32)
    with pytest.raises(nrrd.NRRDError, match="endian"):
        nrrd.read(str(p))
    p.write_bytes(b"NRRD0004\ntype: uchar\ndimension: 2\nsizes: 4 4\nencoding: raw\n\n" + b"\0" * 15)
    with pytest.raises(nrrd.NRRDError, match="payload"):
        nrrd.read(str(p))
    p.write_bytes(b"NRRD0004\ntype: uchar\ndimension: 3\nsizes: 4 4\nencoding: raw\n\n" + b"\0" * 16)
    with pytest.raises(nrrd.NRRDError, match="sizes"):
        nrrd.read(str(p))


# --------------------------------------------------------------------------------------------------------- readers
def _dataset(tmp_path, case):
    data_dir, set_name, csv_dir = VR.write_synthetic_dataset(str(tmp_path), **case["data"])
    return data_dir, set_name, config_for(case, csv_dir)


@pytest.mark.parametrize("tag", list(CASES))
def test_slice_reader_eval_matches_reference(golden, tmp_path, tag):
    """truncate -> pad16 -> annotated z-range -> crop/pad -> HU window -> k-block support/query matching, bit for bit"""
    g, case = golden("volume_reader"), CASES[tag]
    data_dir, set_name, cfg = _dataset(tmp_path, case)
    rd = VR.FewshotSliceReader(data_dir, set_name, dict(cfg, use_registration_loss=False), mode="eval")
    assert len(rd) == int(g[f"{tag}_len"])
    for idx in range(len(rd)):
        random.seed(case["seed"] + idx)
        it = rd[idx]
        p = f"{tag}_slice{idx}_"
        assert it["pid"] == str(g[p + "pid"]) and np.array_equal(np.array(it["supp_pids"]), g[p + "supp_pids"])
        assert tuple(it["query_images_3D"][0][0].shape) == tuple(g[p + "vol_shape"])
        for key, got in [("support_images", it["support_images"][0][0]), ("query_images", it["query_images"]),
                         ("warped_supp", it["warped_supp"])]:
            assert got.dtype == torch.float32 and np.array_equal(got.numpy(), g[p + key]), (tag, idx, key)
        for key, got in [("support_labels", it["support_labels"][0][0]), ("query_labels", it["query_labels"])]:
            assert got.dtype == torch.float32 and np.array_equal(got.numpy().astype(np.uint8), g[p + key]), (tag, idx, key)
        assert it["registration_field"] is None and it["warped_supp_label"] is None
        assert it["query_images"].min() >= -1 and it["query_images"].max() <= 1


def test_volume_reader_contract(tmp_path):
    case = CASES["cut"]
    data_dir, set_name, cfg = _dataset(tmp_path, case)
    rd = VR.FewshotVolumeReader(data_dir, set_name, cfg, mode="eval")
    assert len(rd) == 8 and rd.n_data == [4, 4] and rd.data_info[1][3]["pid"] == "case003"
    it = rd.__getitem__(5, supp_idx=0)
    assert it["class_id"] == 1 and it["pid"] == "case001" and it["supp_pids"] == [(1, 0)]
    img, msk = it["query_images"][0][0], it["query_labels"][0][0]
    assert img.shape == msk.shape and img.shape[0] == 1 and tuple(img.shape[2:]) == (32, 32) and img.dtype == torch.float32
    assert set(np.unique(msk.numpy()).tolist()) <= {0.0, 1.0} and msk.sum() > 0
    # a .npy pid list that keeps two of the volumes
    np.save(str(tmp_path / "two.npy"), np.array(["case001", "case003"]))
    rd2 = VR.FewshotVolumeReader(data_dir, str(tmp_path / "two.npy"), cfg, mode="train")
    assert rd2.n_data == [2, 2] and [r["pid"] for r in rd2.data_info[0]] == ["case001", "case003"]
    with pytest.raises(NotImplementedError):
        VR.FewshotVolumeReader(data_dir, set_name, cfg, mode="test")


def test_slice_reader_train_contract(tmp_path):
    """train mode: one augmented slice per block, k of them, 3 identical channels; seeded runs repeat"""
    case = CASES["pad"]
    data_dir, set_name, cfg = _dataset(tmp_path, case)
    cfg = dict(cfg, use_registration_loss=False, do_elastic=True)

    def item(seed):
        random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
        return VR.FewshotSliceReader(data_dir, set_name, cfg, mode="train")[1]

    a, b, c = item(5), item(5), item(6)
    k = cfg["k"]
    assert tuple(a["query_images"].shape) == (k, 3, 48, 48) and tuple(a["query_labels"].shape) == (k, 48, 48)
    assert tuple(a["support_images"][0][0].shape) == (k, 3, 48, 48) and tuple(a["support_labels"][0][0].shape) == (k, 48, 48)
    assert torch.equal(a["query_images"][:, 0], a["query_images"][:, 2])
    assert set(np.unique(a["query_labels"].numpy()).tolist()) <= {0.0, 1.0}
    assert a["query_images"].min() >= -1 - 1e-5 and a["query_images"].max() <= 1 + 1e-5
    # python/numpy/torch draws are seeded; the elastic transform is not (unseeded RandomState, as in the reference),
    # it only touches the query volume
    assert torch.equal(a["support_images"][0][0], b["support_images"][0][0])
    assert not torch.equal(a["query_images"], c["query_images"])


def test_augmentations():
    torch.manual_seed(0)
    x = torch.zeros(1, 2, 40, 40)
    x[:, :, 12:28, 14:30] = 1
    y = VR.random_affine(x, 0, translate=None, scale=(1.0, 1.0), shear=None)         # identity parameters
    assert torch.equal(x, y)
    torch.manual_seed(1)
    y = VR.random_affine(x, 5, translate=(0.2, 0.2), scale=(0.7, 1.5), shear=0)
    assert y.shape == x.shape and set(np.unique(y.numpy()).tolist()) <= {0.0, 1.0} and torch.equal(y[:, 0], y[:, 1])
    area = y[0, 0].sum().item() / x[0, 0].sum().item()
    assert 0.7 ** 2 * 0.8 < area < 1.5 ** 2 * 1.2
    np.random.seed(0)
    img = np.random.rand(1, 16, 16).astype(np.float32) * 2 - 1
    out = VR.gamma_tansform(img, [0.5, 1.5])
    assert out.shape == img.shape and abs(out.min() - img.min()) < 1e-4 and abs(out.max() - img.max()) < 1e-4
    image = np.full((1, 3, 32, 32), -1.0, np.float32)
    image[:, :, 10:22, 10:22] = 0.5
    mask = (image > 0).astype(np.float32)
    ni, nm = VR.elastic_transform_all(image, mask, random_state=np.random.RandomState(3))
    assert ni.shape == image.shape and nm.shape == mask.shape and set(np.unique(nm).tolist()) <= {0.0, 1.0}
    assert abs(nm.sum() - mask.sum()) < 0.35 * mask.sum() and ni.min() >= -1 - 1e-5


def test_reg_reader_needs_registration_and_gpu(tmp_path):
    """no CPU fallback for the registration pre-step: off a GPU box the reader fails loudly"""
    import dataset.few_shot_reader as fsr
    case = CASES["pad"]
    data_dir, set_name, cfg = _dataset(tmp_path, case)
    rd = fsr.FewshotRegReader(data_dir, set_name, dict(cfg, use_registration_loss=False), mode="eval")
    assert isinstance(rd, VR.FewshotRegReader) and rd.fewshot_reader.fewshot_volume_reader.data_info[0][1]["pid"] == "case001"
    with pytest.raises(TypeError, match="use_registration_loss"):
        rd[0]
    if not torch.cuda.is_available():
        rd = fsr.FewshotRegReader(data_dir, set_name, cfg, mode="eval")
        with pytest.raises(RuntimeError):
            rd[0]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_reg_reader_item_vs_reference(golden, tmp_path, tag):
    """the item test_rpnet.py consumes, HIP registration inside; against the reference's (CPU registration): theta
    within the reference's own cross-CPU spread (2e-3, see tests/test_registration.py), warped sources 2e-2 of a
    [-1,1] image at that theta spread (5e-3 when the base grids agree), labels < 1 % of the pixels"""
    import dataset.few_shot_reader as fsr
    from rpnet_amd.registration import base_grid
    g, case = golden("volume_reader"), CASES[tag]
    data_dir, set_name, cfg = _dataset(tmp_path, case)
    rd = fsr.FewshotRegReader(data_dir, set_name, cfg, mode="eval")
    for idx in case["registration"]:
        random.seed(case["seed"] + idx)
        it = rd[idx]
        p = f"{tag}_reg{idx}_"
        same_grid = np.array_equal(base_grid(it["grid"].shape[-1], "cpu").numpy(), g[p + "base_grid"])
        th_tol, src_tol = (1e-3, 5e-3) if same_grid else (2e-3, 2e-2)
        assert np.abs(it["registration_field"].numpy() - g[p + "theta"]).max() < th_tol
        assert np.array_equal(it["query_images"].numpy(), g[p + "query_images"])
        assert np.array_equal(it["grid"][:1].numpy(), g[p + "grid"]) and it["grid"].shape[0] == it["query_images"].shape[0]
        assert np.abs(it["support_images"][0][0].numpy() - g[p + "support_images"]).max() < src_tol
        assert np.abs(it["warped_supp"].numpy() - g[p + "warped_supp"]).max() < src_tol
        for key, got in [("support_labels", it["support_labels"][0][0]), ("appr_query_labels", it["appr_query_labels"])]:
            flips = (got.numpy().astype(np.uint8) != g[p + key]).mean()
            assert flips < 0.01, (tag, idx, key, flips)
        assert tuple(it["original_support_images"][0][0].shape) == tuple(g[p + "orig_support_images_shape"])


@pytest.mark.parametrize("dt", ["int16", "float32", "uint8"])
@pytest.mark.parametrize("endian", ["little", "big"])
@pytest.mark.parametrize("enc", ["raw", "gzip", "bzip2"])
@pytest.mark.parametrize("detached", [False, True])
def test_nrrd_read_against_independent_writer(tmp_path, dt, endian, enc, detached):
    """The reader against files produced WITHOUT this module: header lines written by hand per the published format
    (teem.sourceforge.net/nrrd/format.html), payload = numpy bytes of a 3-D volume in file order (first header axis
    fastest), byte-swapped for `endian: big`, compressed by Python's own gzip / bz2, attached or in a detached data file."""
    import bz2
    import gzip
    rs = np.random.RandomState(hash((dt, endian, enc, detached)) % (2 ** 31))
    a = (rs.randn(5, 4, 3) * 100).astype(dt)                    # shape == sizes: axis 0 fastest in the file
    names = {"int16": "short", "float32": "float", "uint8": "unsigned char"}
    wire = a.astype(np.dtype(dt).newbyteorder(">" if endian == "big" else "<")).tobytes(order="F")
    payload = {"raw": wire, "gzip": gzip.compress(wire), "bzip2": bz2.compress(wire)}[enc]
    hdr = (f"NRRD0004\n# written by the test, not by rpnet_amd.utils.nrrd\ntype: {names[dt]}\ndimension: 3\nsizes: 5 4 3\n"
           f"endian: {endian}\nencoding: {enc}\nspace: left-posterior-superior\nmodality:=CT\n")
    if detached:
        (tmp_path / "v.raw").write_bytes(payload)
        (tmp_path / "v.nhdr").write_bytes((hdr + "data file: v.raw\n").encode("ascii"))
        got, h = nrrd.read(str(tmp_path / "v.nhdr"))
    else:
        (tmp_path / "v.nrrd").write_bytes((hdr + "\n").encode("ascii") + payload)
        got, h = nrrd.read(str(tmp_path / "v.nrrd"))
    assert got.shape == (5, 4, 3) and got.dtype == np.dtype(dt) and np.array_equal(got, a)
    assert list(h["sizes"]) == [5, 4, 3] and h["space"] == "left-posterior-superior" and h["modality"] == "CT"


@pytest.mark.parametrize("enc", ["raw", "gzip"])
def test_nrrd_write_against_independent_parser(tmp_path, enc):
    """The writer against a parser that shares nothing with the reader: split at the first blank line, read the fields
    with str methods, decode the payload with zlib / numpy."""
    import zlib
    a = (np.arange(2 * 3 * 4).reshape(2, 3, 4) - 7).astype(np.int16)
    p = tmp_path / "w.nrrd"
    nrrd.write(str(p), a, encoding=enc)
    blob = p.read_bytes()
    head, _, payload = blob.partition(b"\n\n")
    lines = head.decode("ascii").split("\n")
    assert lines[0].startswith("NRRD000")
    f = dict(ln.split(": ", 1) for ln in lines[1:] if ": " in ln and not ln.startswith("#"))
    assert f["dimension"] == "3" and f["sizes"].split() == ["2", "3", "4"] and f["type"] in ("short", "int16")
    raw = payload if enc == "raw" else zlib.decompress(payload, 16 + zlib.MAX_WBITS)
    got = np.frombuffer(raw, dtype="<i2" if f["endian"] == "little" else ">i2").reshape((2, 3, 4), order="F")
    assert np.array_equal(got, a)
