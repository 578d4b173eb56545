#!/usr/bin/env python3
"""Trajectory evaluation for KITTI-format pose files (12 floats per row = first three rows of
T_wc, the layout of the reference's Stereo/01.txt ground truth and of the rows
Tracking::SaveTrajectoryAndDraw writes, src/Tracking.cc:129-136).

    python tools/evaluate_ate.py <estimate.txt> <ground_truth.txt>

Reports the un-aligned translation RMSE (both trajectories start at the identity), the final
position error and the per-frame step lengths - the figures SURVEY.md section 6 derives for the
reference's own shipped output."""
import sys

import numpy as np


def load_kitti(path):
    rows = np.loadtxt(path).reshape(-1, 12)
    T = np.tile(np.eye(4), (len(rows), 1, 1))
    T[:, :3, :] = rows.reshape(-1, 3, 4)
    return T


def ate(est, gt):
    n = min(len(est), len(gt))
    err = np.linalg.norm(est[:n, :3, 3] - gt[:n, :3, 3], axis=1)
    steps_e = np.linalg.norm(np.diff(est[:n, :3, 3], axis=0), axis=1)
    steps_g = np.linalg.norm(np.diff(gt[:n, :3, 3], axis=0), axis=1)
    return dict(frames=n, rmse_m=float(np.sqrt(np.mean(err ** 2))), final_error_m=float(err[-1]),
                max_error_m=float(err.max()), step_est_min_max=(float(steps_e.min()), float(steps_e.max())),
                step_gt_min_max=(float(steps_g.min()), float(steps_g.max())),
                path_length_m=float(steps_g.sum()))


if __name__ == "__main__":
    r = ate(load_kitti(sys.argv[1]), load_kitti(sys.argv[2]))
    for k, v in r.items():
        print("%-18s %s" % (k, v))
