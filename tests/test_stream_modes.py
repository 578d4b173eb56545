"""The context's stream mode and the process around it (VERDICT r5 missing #4 / #5).

* The DEFAULT mode makes the context's four main streams as dedicated hardware queues, which HIP only offers as BLOCKING streams:
  they order against the NULL stream - exactly the ordering whose absence caused round 4's hipMemset race.  A dependency the
  library forgets is therefore masked in the default mode and live in the fallback mode (pooled NON-blocking streams:
  SVO_POOLED_QUEUES=1 / SVO_CREATE_POOLED_STREAMS, also what a device without CU-masked streams gets).  So the parity tests that
  exercise every overlap the library builds - the 64-frame tracker against the free-running oracle, configs[4] (boxes + dense ELAS
  + tracker), the sharded tracker's byte identity (direct and through the pinned bounce), the host-fed entries - are re-run in a
  child process in that mode.
* Another GPU user in the process (the reference's detector thread, src/semantic.cc:13-45): the tracker's records must not
  change while a host thread keeps kernels running on the NULL stream or on a non-blocking stream of its own, in either mode;
  and two independent trackers (two contexts, two host threads) on one GPU must each reproduce the single tracker."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

POOLED_SUBSET = [
    "tests/test_configs.py::test_64_frames_gpu_tracker_equals_oracle",
    "tests/test_configs.py::test_config4_boxes_with_dense_elas_depth_equals_oracle",
    "tests/test_shard.py::test_track_sharded_contexts_equal_single_context",
    "tests/test_shard.py::test_sharded_calls_back_to_back_overlap_and_stay_identical",
    "tests/test_shard.py::test_closing_a_producer_right_after_an_unsynced_sharded_call",
    "tests/test_track.py::test_gpu_batch_tracker_equals_frame_by_frame",
    "tests/test_track.py::test_gpu_batch_tracker_with_dense_depth_equals_frame_by_frame",
    "tests/test_track.py::test_multi_sequence_tracker_equals_independent_chains",
    "tests/test_hostfeed.py",
]


@pytest.mark.gpu
def test_overlap_parity_tests_in_the_pooled_non_blocking_stream_mode():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, SVO_POOLED_QUEUES="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + POOLED_SUBSET,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-3000:] + out.stderr[-1000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail
    print("POOLED_MODE " + out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_tracker_records_do_not_change_beside_another_gpu_user_of_the_process(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    import bench
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    dev = torch.device("cuda", 0)
    B, calls = 24, 3
    dL, dR, _ = bench.render_frames(synth, 2 * B * calls, dev, synth.BASE_SEED)
    cam = pkg.Camera(**pkg.KITTI_00_02)
    rec = pkg.TRACK_DTYPE.itemsize
    ref = pkg.Svo(bench.W, bench.H, max_batch=B)
    ref.track_reset(cam)
    want = torch.zeros((B * calls, rec), dtype=torch.uint8, device=dev)
    fb = bench.H * bench.PITCH
    for c in range(calls):
        ref.track_batch_dev(dL.data_ptr() + c * B * fb, dR.data_ptr() + c * B * fb, bench.PITCH, B, want.data_ptr() + c * B * rec)
    ref.sync(); ref.close()
    leg = bench.cotenant_leg(pkg, cam, dL, dR, dev, 0, want.cpu().numpy(), B, calls=calls, warm=1)
    print("COTENANT " + repr({k: v for k, v in leg.items() if isinstance(v, dict)}))
    assert leg["records_identical"] is True, leg
    for mode in ("dedicated_queues", "pooled_streams"):
        assert leg[mode]["cotenant_kernels_null"] > 0 and leg[mode]["cotenant_kernels_pooled"] > 0      # the co-tenant did run
        assert min(leg[mode]["alone"], leg[mode]["beside_null_stream_cotenant"], leg[mode]["beside_pooled_stream_cotenant"]) > 0
    assert leg["two_contexts_one_gpu"]["first_equals_the_single_tracker"] is True
