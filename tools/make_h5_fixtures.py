#!/opt/conda/bin/python3.9
"""Writes the synthetic episodes of tests/golden/episodes/*.npz as tests/golden/episodes_h5/episode_*.h5 with the REAL h5py (3.3.0 / HDF5 1.10.6,
present in this image only under /opt/conda/bin/python3.9) exactly the way the reference writes its episodes:
`create_dataset(name, data=..., compression='lzf')`, one group per sensor folder
(/root/reference/VLA/data/franka_data/4_convert_to_hdf5.py:30-167, data/create_controller_dataset_episode.py:161-213).
They pin vlatouch/h5lite.py (the product's own HDF5 reader: no h5py on the GPU box's python) to real h5py output.
Also writes one file with the other storage forms h5py users produce: gzip + shuffle, contiguous, compact-size scalars,
a many-chunk dataset (multi-level chunk B-tree) and attributes.
    /opt/conda/bin/python3.9 tools/make_h5_fixtures.py
"""
import os

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EP = os.path.join(ROOT, "tests", "golden", "episodes")
OUT = os.path.join(ROOT, "tests", "golden", "episodes_h5")


def main():
    for name in sorted(os.listdir(EP)):
        if not name.endswith(".npz"):
            continue
        z = np.load(os.path.join(EP, name))
        out = os.path.join(OUT, name.replace(".npz", ".h5"))
        with h5py.File(out, "w") as hf:
            groups = {}
            for k in z.files:
                if "/" in k:
                    g, d = k.split("/")
                    grp = groups.get(g) or groups.setdefault(g, hf.create_group(g))
                    grp.create_dataset(d, data=z[k], compression="lzf")
                else:
                    hf.create_dataset(k, data=z[k], compression="lzf")
            hf.create_dataset("instruct_embeddings", data=np.linspace(-1, 1, 6 * 32, dtype=np.float32).reshape(1, 6, 32), compression="lzf")
        print(out, os.path.getsize(out))
    rng = np.random.default_rng(5)
    with h5py.File(os.path.join(OUT, "storage_forms.h5"), "w") as hf:
        hf.create_dataset("gz_shuffle_i16", data=rng.integers(-300, 300, (50, 40), dtype=np.int16), compression="gzip", shuffle=True)
        hf.create_dataset("contiguous_f32", data=rng.standard_normal((7, 5)).astype(np.float32))
        hf.create_dataset("scalar_i64", data=np.int64(-42))
        hf.create_dataset("many_chunks_u8", data=rng.integers(0, 255, (300, 64), dtype=np.uint8), chunks=(1, 64), compression="lzf")
        hf.create_dataset("incompressible_lzf", data=rng.integers(0, 255, (4, 4096), dtype=np.uint8), chunks=(1, 4096), compression="lzf")
        hf.create_dataset("edge_chunks_f64", data=rng.standard_normal((10, 7, 3)), chunks=(4, 4, 2), compression="lzf")
        hf.create_dataset("bool_as_u8", data=np.array([1, 0, 1], dtype=np.uint8))
        g = hf.create_group("a")
        g.create_group("b").create_dataset("deep", data=np.arange(5, dtype=np.uint32))
        hf.attrs["resized"] = True
        g.attrs["target_width"] = 28
    print("storage_forms.h5", os.path.getsize(os.path.join(OUT, "storage_forms.h5")))


if __name__ == "__main__":
    main()
