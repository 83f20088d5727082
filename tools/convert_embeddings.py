"""image_embeddings_<model>.hdf5['images'] -> image_embeddings_<model>.npy, for machines without h5py at training time
(fumi_amd/dataset/inat_anim.py reads either).  Needs h5py where it is run.

    python tools/convert_embeddings.py <data_dir>/iNat-Anim/image_embeddings_resnet-152.hdf5
"""
import sys

import numpy as np


def main(path):
    import h5py
    with h5py.File(path, "r") as f:
        d = f["images"]
        out = np.lib.format.open_memmap(path[:-len(".hdf5")] + ".npy", mode="w+", dtype=np.float32, shape=d.shape)
        step = 16384
        for s in range(0, d.shape[0], step):
            out[s:s + step] = d[s:s + step]
        out.flush()
    print("wrote", path[:-len(".hdf5")] + ".npy", d.shape)


if __name__ == "__main__":
    main(sys.argv[1])
