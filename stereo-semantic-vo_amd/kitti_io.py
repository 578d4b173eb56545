"""KITTI odometry sequence I/O in the layout the reference's driver reads (main.cpp:20-57):

    <root>/sequences/<seq>/times.txt
    <root>/sequences/<seq>/image_2/NNNNNN.png   (left, colour)    or image_0 (gray)
    <root>/sequences/<seq>/image_3/NNNNNN.png   (right, colour)   or image_1 (gray)
    <root>/poses/<seq>.txt                      (optional ground truth, 12 floats per row)

Real data is opt-in (there is none offline): bench.py uses it when KITTI_ROOT is set, otherwise the
synth-kitti renderer.  Colour frames are reduced to gray with cv::cvtColor's fixed-point weights,
which is what cv::ORB does with the reference's 8UC3 images."""
import os

import numpy as np


def _read_gray(path):
    from PIL import Image
    im = np.asarray(Image.open(path))
    if im.ndim == 3:
        r, g, b = (im[..., i].astype(np.int64) for i in range(3))
        im = ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    return np.ascontiguousarray(im, np.uint8)


def sequence_dir(root, seq):
    for cand in (os.path.join(root, "sequences", seq), os.path.join(root, seq), root):
        if os.path.exists(os.path.join(cand, "times.txt")):
            return cand
    raise FileNotFoundError("no times.txt for sequence %s under %s" % (seq, root))


def load_times(root, seq):
    with open(os.path.join(sequence_dir(root, seq), "times.txt")) as f:
        return np.array([float(s) for s in f.read().split()])


def load_poses(root, seq):
    for cand in (os.path.join(root, "poses", seq + ".txt"), os.path.join(sequence_dir(root, seq), "poses.txt")):
        if os.path.exists(cand):
            rows = np.loadtxt(cand).reshape(-1, 12)
            T = np.tile(np.eye(4), (len(rows), 1, 1))
            T[:, :3, :] = rows.reshape(-1, 3, 4)
            return T
    return None


def load_frames(root, seq, start, count):
    """(L, R) uint8 arrays (count, H, W) of frames start .. start+count-1."""
    d = sequence_dir(root, seq)
    left, right = ("image_2", "image_3") if os.path.isdir(os.path.join(d, "image_2")) else ("image_0", "image_1")
    Ls, Rs = [], []
    for k in range(start, start + count):
        Ls.append(_read_gray(os.path.join(d, left, "%06d.png" % k)))
        Rs.append(_read_gray(os.path.join(d, right, "%06d.png" % k)))
    return np.stack(Ls), np.stack(Rs)
