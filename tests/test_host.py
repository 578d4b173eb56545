"""C++ host classes (stereo-semantic-vo_amd/host): the reference's class surface over the C-ABI."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "stereo-semantic-vo_amd", "host")


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def read_pgm(path):
    import util
    return util.read_pgm(path)


def test_host_binaries_exist():
    for b in ("stereo_kitti", "host_check", "elas", "libstereo_vo_host.a"):
        assert os.path.exists(os.path.join(HOST, b)), "run __graft_entry__.build()"


def test_png_decoder_matches_pillow(tmp_path):
    """The driver's PNG reader (zlib only) vs Pillow, gray and RGB with cv's BGR2GRAY weights."""
    from PIL import Image
    rng = np.random.default_rng(0)
    gray = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    rgb = rng.integers(0, 256, (29, 41, 3), dtype=np.uint8)
    Image.fromarray(gray).save(tmp_path / "g.png")
    Image.fromarray(rgb).save(tmp_path / "c.png")
    exe = os.path.join(HOST, "stereo_kitti")
    for name in ("g", "c"):
        subprocess.check_call([exe, "--decode", str(tmp_path / (name + ".png")), str(tmp_path / (name + ".pgm"))])
    assert np.array_equal(read_pgm(str(tmp_path / "g.pgm")), gray)
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    expect = ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(read_pgm(str(tmp_path / "c.pgm")), expect)


def test_settings_reader_reads_reference_yaml_keys(tmp_path):
    # the reference consumes exactly Camera.fx/fy/cx/cy/bf (src/Tracking.cc:24-38)
    y = tmp_path / "KITTI00-02.yaml"
    y.write_text("%YAML:1.0\nCamera.fx: 718.856\nCamera.fy: 718.856\nCamera.cx: 607.1928\n"
                 "Camera.cy: 185.2157\nCamera.width: 1241\nCamera.height: 376\nCamera.bf: 386.1448\n"
                 "ORBextractor.nFeatures: 2000\n")
    (tmp_path / "seq").mkdir()
    # no times.txt -> the driver must stop before touching the GPU, after parsing its arguments
    p = subprocess.run([os.path.join(HOST, "stereo_kitti"), "voc", str(y), str(tmp_path / "seq")],
                       capture_output=True, text=True)
    assert p.returncode == 1 and "no times.txt" in p.stderr


@pytest.mark.gpu
def test_host_classes_equal_device_tracker(pkg, tmp_path):
    """frame/Tracking/pnpmatch/Optimizer driven seam by seam == svo_track_frame (poses, map size)."""
    import importlib
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    n = 6
    L, R, _ = synth.render_sequence(n)
    (tmp_path / "image_0").mkdir(); (tmp_path / "image_1").mkdir()
    for k in range(n):
        write_pgm(str(tmp_path / "image_0" / ("%06d.pgm" % k)), L[k].numpy())
        write_pgm(str(tmp_path / "image_1" / ("%06d.pgm" % k)), R[k].numpy())
    p = subprocess.run([os.path.join(HOST, "host_check"), str(tmp_path), str(n)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    rows = np.loadtxt(str(tmp_path / "host_kitti.txt"))
    assert rows.shape == (n, 12)
    assert abs(rows[-1, 11] - (n - 1)) < 0.5     # ~1 m per frame forward
    ev = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("elas_valid")]
    assert ev and int(ev[0][1]) > 0.5 * int(ev[0][3])      # frame::ElasMatch: dense map mostly valid
    # the same comparison with the dense ELAS map as the depth source on both sides (Tracking::depth_source = 1
    # seam by seam vs svo_set_option("depth_source", 1))
    p = subprocess.run([os.path.join(HOST, "host_check"), str(tmp_path), str(n), "dense"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    worst = [float(ln.split()[1]) for ln in p.stdout.splitlines() if ln.startswith("worst")]
    assert worst and worst[0] < 1e-3
    # and with the MSA map, the reference's live configuration (Tracking::depth_source = 2: featuredetect, MBdense,
    # computekeypoint_r, disp2Depth vs svo_set_option("depth_source", 2))
    p = subprocess.run([os.path.join(HOST, "host_check"), str(tmp_path), "3", "msa"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    worst = [float(ln.split()[1]) for ln in p.stdout.splitlines() if ln.startswith("worst")]
    assert worst and worst[0] < 1e-3


@pytest.mark.gpu
def test_host_classes_equal_device_tracker_with_boxes(pkg, tmp_path):
    """Same comparison with offline detection boxes (creation gates + epipolar veto)."""
    import importlib
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    n = 5
    L, R, _ = synth.render_sequence(n)
    (tmp_path / "image_0").mkdir(); (tmp_path / "image_1").mkdir(); (tmp_path / "boxes").mkdir()
    for k in range(n):
        write_pgm(str(tmp_path / "image_0" / ("%06d.pgm" % k)), L[k].numpy())
        write_pgm(str(tmp_path / "image_1" / ("%06d.pgm" % k)), R[k].numpy())
        txt = "500 760 200 330\n" if k == 0 else "200 1000 195 370\n20 120 30 90\n"
        (tmp_path / "boxes" / ("%d.txt" % (k + 1))).write_text(txt)
    p = subprocess.run([os.path.join(HOST, "host_check"), str(tmp_path), str(n)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


@pytest.mark.gpu
def test_stereo_kitti_driver_runs(pkg, tmp_path):
    import importlib
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    from PIL import Image
    n = 4
    L, R, _ = synth.render_sequence(n)
    seq = tmp_path / "seq"
    (seq / "image_2").mkdir(parents=True); (seq / "image_3").mkdir()
    for k in range(n):
        Image.fromarray(np.stack([L[k].numpy()] * 3, -1)).save(seq / "image_2" / ("%06d.png" % k))
        Image.fromarray(np.stack([R[k].numpy()] * 3, -1)).save(seq / "image_3" / ("%06d.png" % k))
    (seq / "times.txt").write_text("".join("%e\n" % (0.1 * k) for k in range(n)))
    y = tmp_path / "s.yaml"
    y.write_text("%YAML:1.0\nCamera.fx: 718.856\nCamera.fy: 718.856\nCamera.cx: 607.1928\nCamera.cy: 185.2157\n"
                 "Camera.width: 1241\nCamera.height: 376\nCamera.bf: 386.1448\n")
    p = subprocess.run([os.path.join(HOST, "stereo_kitti"), "voc", str(y), str(seq)], capture_output=True,
                       text=True, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout + p.stderr
    assert "median tracking time" in p.stdout
    kitti = np.loadtxt(str(tmp_path / "cameratrajectory_kitti.txt"))
    tum = np.loadtxt(str(tmp_path / "cameratrajectory_tum.txt"))
    assert kitti.shape == (n, 12) and tum.shape == (n, 8)


@pytest.mark.gpu
def test_stereo_kitti_pipelined_writes_the_frame_by_frame_trajectory(pkg, tmp_path):
    """stereo_kitti --pipelined (Tracking::TrackBatch over svo_track_batch_host, 3 frames per call, offline boxes on some
    frames) against the frame-by-frame driver (Tracking::Track, one C-ABI call per reference seam) on the same PGM sequence:
    the two trajectory files must agree row for row (the host classes' tail and the device tail are the same computation:
    test_host_classes_equal_device_tracker)."""
    import importlib
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    n = 8
    L, R, _ = synth.render_sequence(n)
    seq = tmp_path / "seq"
    (seq / "image_0").mkdir(parents=True); (seq / "image_1").mkdir(); (seq / "boxes").mkdir()
    for k in range(n):
        write_pgm(str(seq / "image_0" / ("%06d.pgm" % k)), L[k].numpy())
        write_pgm(str(seq / "image_1" / ("%06d.pgm" % k)), R[k].numpy())
        if k in (2, 3, 6):
            (seq / "boxes" / ("%d.txt" % (k + 1))).write_text("200 600 195 370\n20 120 30 90\n")
    (seq / "times.txt").write_text("".join("%e\n" % (0.1 * k) for k in range(n)))
    y = tmp_path / "s.yaml"
    y.write_text("%YAML:1.0\nCamera.fx: 718.856\nCamera.fy: 718.856\nCamera.cx: 607.1928\nCamera.cy: 185.2157\n"
                 "Camera.width: 1241\nCamera.height: 376\nCamera.bf: 386.1448\n")
    out = {}
    for mode, extra in (("frame", []), ("pipelined", ["3"])):
        d = tmp_path / mode
        d.mkdir()
        cmd = [os.path.join(HOST, "stereo_kitti")] + (["--pipelined"] if extra else []) + ["voc", str(y), str(seq)] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, cwd=str(d))
        assert p.returncode == 0, p.stdout + p.stderr
        if extra:
            assert "frames per second" in p.stdout
        out[mode] = (np.loadtxt(str(d / "cameratrajectory_kitti.txt")), np.loadtxt(str(d / "cameratrajectory_tum.txt")))
    assert out["frame"][0].shape == (n, 12) and out["pipelined"][0].shape == (n, 12)
    assert np.abs(out["frame"][0] - out["pipelined"][0]).max() < 1e-4
    assert np.abs(out["frame"][1] - out["pipelined"][1]).max() < 1e-4
    assert abs(out["pipelined"][0][-1, 11] - (n - 1)) < 0.5
