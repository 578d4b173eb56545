"""KITTI sequence loader (reference layout, main.cpp:20-57) on a tiny generated sequence."""
import importlib

import numpy as np


def test_kitti_loader_roundtrip(pkg, tmp_path):
    from PIL import Image
    kio = importlib.import_module("stereo_semantic_vo_amd.kitti_io")
    seq = tmp_path / "sequences" / "00"
    (seq / "image_2").mkdir(parents=True); (seq / "image_3").mkdir()
    (tmp_path / "poses").mkdir()
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (3, 2, 20, 30, 3), dtype=np.uint8)
    for k in range(3):
        Image.fromarray(imgs[k, 0]).save(seq / "image_2" / ("%06d.png" % k))
        Image.fromarray(imgs[k, 1]).save(seq / "image_3" / ("%06d.png" % k))
    (seq / "times.txt").write_text("0.000000e+00\n1.037224e-01\n2.073990e-01\n")
    rows = np.tile(np.eye(4)[:3].reshape(12), (3, 1)); rows[:, 11] = [0, 1, 2]
    np.savetxt(tmp_path / "poses" / "00.txt", rows)
    L, R = kio.load_frames(str(tmp_path), "00", 1, 2)
    assert L.shape == (2, 20, 30) and L.dtype == np.uint8
    r, g, b = (imgs[1, 0][..., i].astype(np.int64) for i in range(3))
    assert np.array_equal(L[0], ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8))
    assert np.allclose(kio.load_times(str(tmp_path), "00"), [0.0, 0.1037224, 0.207399])
    T = kio.load_poses(str(tmp_path), "00")
    assert T.shape == (3, 4, 4) and T[2, 2, 3] == 2.0
