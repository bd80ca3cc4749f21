"""vlatouch.h5lite — the product's HDF5 reader / writer for the reference's episode files.
Pinned to REAL h5py output: tests/golden/episodes_h5/*.h5 were written by h5py 3.3.0 / HDF5 1.10.6 exactly the way the reference
writes episodes (tools/make_h5_fixtures.py: create_dataset(..., compression='lzf'), groups per sensor).  The writer is checked
by reading its files back with the real h5py where that interpreter exists (/opt/conda/bin/python3.9 in this image)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import cases
from vlatouch import h5lite as H
from vlatouch import convert, eval as ev

EP = os.path.join(cases.GOLDEN, "episodes")
EPH = os.path.join(cases.GOLDEN, "episodes_h5")
PY39 = "/opt/conda/bin/python3.9"
have_h5py = os.path.exists(PY39) and subprocess.run([PY39, "-c", "import h5py"], capture_output=True).returncode == 0


@pytest.mark.parametrize("name", ["episode_1", "episode_2"])
def test_reads_real_h5py_lzf_episode(name):
    z = np.load(os.path.join(EP, f"{name}.npz"))
    with H.File(os.path.join(EPH, f"{name}.h5")) as f:
        assert set(f.keys()) == {"ee_poses", "gripper_pos", "vla_action", "gelsight_force", "camera1_resized", "camera2_resized", "instruct_embeddings"}
        assert isinstance(f["gelsight_force"], H.Group) and set(f["gelsight_force"].keys()) == {"forces", "displacement"}
        for k in z.files:
            d = f[k]
            assert d.compression == "lzf" and d.chunks is not None and d.shape == z[k].shape and d.dtype == z[k].dtype
            assert np.array_equal(d[...], z[k]), k
        assert np.array_equal(f["gelsight_force"]["forces"][3:5], z["gelsight_force/forces"][3:5])      # the reference's access pattern
        assert "vla_action" in f and "nope" not in f
        with pytest.raises(KeyError):
            f["nope"]
    # the evaluation harness reads the .h5 episode exactly like the .npz one
    a, b = ev.load_episode(os.path.join(EPH, f"{name}.h5")), ev.load_episode(os.path.join(EP, f"{name}.npz"))
    for k in b:
        assert np.array_equal(a[k], b[k]), k
    assert ev.episode_windows(a, 2, 8) == ev.episode_windows(b, 2, 8)


def test_reads_other_storage_forms_written_by_h5py():
    rng = np.random.default_rng(5)      # the generator sequence of tools/make_h5_fixtures.py
    want = {
        "gz_shuffle_i16": rng.integers(-300, 300, (50, 40), dtype=np.int16),
        "contiguous_f32": rng.standard_normal((7, 5)).astype(np.float32),
        "many_chunks_u8": rng.integers(0, 255, (300, 64), dtype=np.uint8),
        "incompressible_lzf": rng.integers(0, 255, (4, 4096), dtype=np.uint8),
        "edge_chunks_f64": rng.standard_normal((10, 7, 3)),
    }
    with H.File(os.path.join(EPH, "storage_forms.h5")) as f:
        for k, v in want.items():
            assert np.array_equal(f[k][...], v), k
        assert f["gz_shuffle_i16"].compression == "gzip" and f["contiguous_f32"].chunks is None
        assert f["many_chunks_u8"].chunks == (1, 64)                 # 300 chunks: a two-level chunk B-tree
        assert int(f["scalar_i64"][...]) == -42 and f["scalar_i64"].shape == ()
        assert np.array_equal(f["a/b/deep"][...], np.arange(5, dtype=np.uint32)) and f["a"].attrs["target_width"] == 28
    with pytest.raises(ValueError):
        H._Reader(b"not an hdf5 file at all" * 100)


def test_lzf_codec_round_trip_and_python_decoder():
    rng = np.random.default_rng(0)
    for raw in (bytes(1000), rng.integers(0, 4, 70000).astype(np.uint8).tobytes(), b"abcabcabcabc" * 500 + b"xyz", np.arange(300, dtype=np.float64).tobytes()):
        c = H.lzf_compress(raw)
        assert c is not None and len(c) < len(raw)
        assert H.lzf_decompress(c, len(raw)) == raw and H.lzf_decompress_py(c, len(raw)) == raw
    assert H.lzf_compress(rng.integers(0, 256, 5000).astype(np.uint8).tobytes()) is None      # incompressible: stored raw
    with pytest.raises(ValueError):
        H.lzf_decompress(b"\xff\xff\xff", 10)


def _tree():
    z = np.load(os.path.join(EP, "episode_1.npz"))
    tree = {k: z[k] for k in z.files}
    rng = np.random.default_rng(1)
    tree["big_u8"] = rng.integers(0, 3, (70, 96, 96, 3), dtype=np.uint8)          # 192 chunks
    tree["rand_f32"] = rng.standard_normal((5, 4096)).astype(np.float32)           # incompressible chunks
    tree["i16"] = np.arange(-5, 5, dtype=np.int16)
    return tree


def test_writer_round_trip(tmp_path):
    tree = _tree()
    p = str(tmp_path / "w.h5")
    H.write_file(p, tree)
    with H.File(p) as f:
        for k, v in tree.items():
            assert np.array_equal(f[k][...], v) and f[k].compression == "lzf", k


@pytest.mark.skipif(not have_h5py, reason="no interpreter with the real h5py here")
def test_real_h5py_reads_what_the_writer_wrote(tmp_path):
    tree = _tree()
    p = str(tmp_path / "w.h5")
    H.write_file(p, tree)
    np.savez(str(tmp_path / "want.npz"), **{k.replace("/", "__"): v for k, v in tree.items()})
    code = ("import h5py, numpy as np, sys\n"
            f"f = h5py.File({p!r}, 'r'); z = np.load({str(tmp_path / 'want.npz')!r})\n"
            "for k in z.files:\n"
            "    d = f[k.replace('__', '/')]\n"
            "    assert d.compression == 'lzf' and np.array_equal(d[...], z[k]), k\n"
            "assert isinstance(f['gelsight_force'], h5py.Group)\n"
            "print('H5PY_OK', len(z.files))\n")
    r = subprocess.run([PY39, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "H5PY_OK" in r.stdout, r.stderr[-2000:]


def test_convert_raw_episode_folder_and_labelled_writer(tmp_path):
    """4_convert_to_hdf5.py's layout from a raw recording folder, then create_controller_dataset_episode.py's labelled file."""
    from PIL import Image
    rng = np.random.default_rng(3)
    ep = tmp_path / "raw" / "episode_7"
    (ep / "camera1").mkdir(parents=True)
    (ep / "gelsight_force").mkdir()
    N = 5
    np.save(ep / "ee_poses.npy", rng.standard_normal((N, 7)))
    np.save(ep / "gripper_pos.npy", rng.uniform(0, 255, N))
    frames = rng.integers(0, 255, (N, 20, 24, 3), dtype=np.uint8)
    for i in (3, 0, 4, 1, 2):                                              # written out of order: sorted by the number in `rgb_<n>.jpg`
        Image.fromarray(frames[i]).save(ep / "camera1" / f"rgb_{i}.jpg", quality=95)
    frames = np.stack([np.asarray(Image.open(ep / "camera1" / f"rgb_{i}.jpg").convert("RGB")) for i in range(N)])   # what a decoder returns
    np.save(ep / "gelsight_force" / "forces.npy", rng.standard_normal((N, 3)))
    out = str(tmp_path / "h5" / "episode_7.h5")
    assert convert.convert_dataset_to_hdf5(str(tmp_path / "raw"), str(tmp_path / "h5")) == [out]
    with H.File(out) as f:
        assert np.array_equal(f["camera1"]["camera1"][...], frames)          # one array per image folder, named like the folder
        assert f["ee_poses"].shape == (N, 7) and np.array_equal(f["gelsight_force"]["forces"][...], np.load(ep / "gelsight_force" / "forces.npy"))
    vla = rng.standard_normal((N, 64, 10)).astype(np.float32)
    cam = rng.integers(0, 255, (N, 16, 16, 3), dtype=np.uint8)
    lab = str(tmp_path / "labelled" / "episode_7.h5")
    convert.write_labelled_episode(out, lab, vla, cam, cam[::-1].copy())
    e = ev.load_episode(lab)
    assert np.array_equal(e["vla_action"], vla) and np.array_equal(e["camera2_resized"], cam[::-1]) and e["gelsight_force/forces"].shape == (N, 3)
    with pytest.raises(ValueError):
        convert.write_labelled_episode(out, lab, vla[:-1], cam, cam)


# ---- property tests (hypothesis): the codec and the container on arbitrary inputs
from hypothesis import given, settings, strategies as st   # noqa: E402


@settings(max_examples=150, deadline=None)
@given(st.binary(min_size=0, max_size=4096), st.integers(1, 40), st.integers(0, 3))
def test_lzf_round_trip_property(chunk, repeat, mode):
    """decompress(compress(x)) == x for any byte string (repetitive, mixed or empty), on both decoders; a None from the encoder means
    'stored raw' (not compressible into fewer bytes), exactly liblzf's contract."""
    raw = chunk * repeat if mode < 2 else chunk + bytes(len(chunk)) * repeat + chunk[::-1]
    if mode == 3:
        raw = raw[: len(raw) // 2]
    c = H.lzf_compress(raw)
    if c is None:
        return
    assert len(c) < max(len(raw), 1)
    assert H.lzf_decompress(c, len(raw)) == raw
    assert H.lzf_decompress_py(c, len(raw)) == raw


@settings(max_examples=25, deadline=None)
@given(st.lists(st.tuples(st.sampled_from(["u1", "i2", "i4", "f4", "f8", "i8"]), st.lists(st.integers(1, 9), min_size=1, max_size=4), st.integers(0, 2 ** 31)),
                min_size=1, max_size=5))
def test_writer_reader_round_trip_property(tmp_path_factory, specs):
    """write_file -> File for arbitrary dtypes / ranks / shapes (compressible and incompressible contents, nested group)."""
    tree = {}
    for i, (dt, shape, seed) in enumerate(specs):
        rng = np.random.default_rng(seed)
        a = rng.integers(0, 5 if seed % 2 else 250, size=shape).astype(dt) if dt[0] in "ui" else rng.standard_normal(shape).astype(dt)
        tree[f"d{i}" if i % 2 == 0 else f"grp/d{i}"] = a
    p = str(tmp_path_factory.mktemp("h5p") / "t.h5")
    H.write_file(p, tree)
    with H.File(p) as f:
        for k, v in tree.items():
            got = f[k][...]
            assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
            if v.ndim >= 1 and v.shape[0] > 1:
                assert np.array_equal(f[k][1:], v[1:])
