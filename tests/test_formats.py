"""Harness / wire formats (SURVEY f-4) pinned on the data files the reference itself ships:
Stereo/01.txt (KITTI ground truth), Stereo/cameratrajectory_{kitti,tum}.txt (the first 10 poses the
reference produced on KITTI 01) - copied as fixtures under tests/golden/."""
import os
import re
import subprocess
import sys

import numpy as np

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import evaluate_ate  # noqa: E402

HOST = os.path.join(ROOT, "stereo-semantic-vo_amd", "host")
KITTI = os.path.join(util.GOLDEN, "ref_cameratrajectory_kitti_seq01.txt")
TUM = os.path.join(util.GOLDEN, "ref_cameratrajectory_tum_seq01.txt")
GT = os.path.join(util.GOLDEN, "kitti01_gt_first10.txt")


def test_evaluator_reproduces_survey_figures():
    """the reference's shipped output is 0.97 m RMSE / 1.32 m final error vs
    ground truth over its 10 frames, steps 0.89-1.38 m vs 0.99-1.00 m."""
    r = evaluate_ate.ate(evaluate_ate.load_kitti(KITTI), evaluate_ate.load_kitti(GT))
    assert r["frames"] == 10
    assert abs(r["rmse_m"] - 0.97) < 0.005 and abs(r["final_error_m"] - 1.32) < 0.005
    assert 0.88 < r["step_est_min_max"][0] < 0.90 and 1.37 < r["step_est_min_max"][1] < 1.39
    assert 0.98 < r["step_gt_min_max"][0] and r["step_gt_min_max"][1] < 1.01


def test_to_quaternion_matches_reference_output():
    """convert::toQuaternion (src/convert.cc:76-88 = Eigen::Quaterniond(R)) pinned on the reference's
    own files: the rotation written to the KITTI file must give the quaternion in the TUM file."""
    kitti = np.loadtxt(KITTI).reshape(-1, 3, 4)
    tum = np.loadtxt(TUM)
    exe = os.path.join(HOST, "stereo_kitti")
    for k in range(len(kitti)):
        R = kitti[k][:, :3]
        out = subprocess.check_output([exe, "--quat"] + ["%.9f" % v for v in R.reshape(9)], text=True)
        q = np.array([float(v) for v in out.split()])
        assert np.abs(q - tum[k, 4:8]).max() < 2e-6, (k, q, tum[k, 4:8])
        # translation columns of both files agree too (twc)
        assert np.abs(kitti[k][:, 3] - tum[k, 1:4]).max() < 1e-6


def test_trajectory_line_formats():
    """`fixed`, 9 decimals, 12 values per KITTI row; `fixed`, 6-decimal stamp + 7-decimal pose per TUM
    row (main.cpp:141-146, src/Tracking.cc:129-136)."""
    krow = re.compile(r"^(-?\d+\.\d{9} ){11}-?\d+\.\d{9}$")
    trow = re.compile(r"^\d+\.\d{6}( -?\d+\.\d{7}){7}$")
    for line in open(KITTI).read().splitlines():
        assert krow.match(line), line
    for line in open(TUM).read().splitlines():
        assert trow.match(line), line


def test_reference_frame0_pose_is_optimised_not_identity():
    """The reference's first row is NOT exactly the identity: Optimizer::PoseOptimization also runs on
    frame 0 (src/Tracking.cc:107-121) - the behaviour the tracker and the oracle reproduce."""
    first = np.loadtxt(KITTI)[0]
    assert np.abs(first - np.eye(4)[:3].reshape(12)).max() > 1e-9
    assert np.abs(first - np.eye(4)[:3].reshape(12)).max() < 1e-5
