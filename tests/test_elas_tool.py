"""The `elas` command-line tool (host/elas_main.cc) against the reference's own demo program
(Thirdparty/libelas/src/main.cpp compiled unmodified by oracle/Makefile.ref into oracle/_ref/ref_elas_demo):
same command line, same `<name>_disp.pgm` files.  Headers must be identical; pixel bytes may differ only
where Triangle's pool-order artefact decides (see tests/test_elas_delaunay.py) - a 1e-4 fraction at most."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "stereo-semantic-vo_amd", "host", "elas")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_elas_demo")

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")]


@pytest.mark.parametrize("name", ["urban1", "cones"])
def test_tool_writes_the_reference_programs_files(tmp_path, name):
    out = {}
    for who, exe in (("ref", REF), ("gpu", TOOL)):
        d = tmp_path / who
        d.mkdir()
        for side in ("left", "right"):
            shutil.copy(os.path.join(util.GOLDEN, "%s_%s.pgm" % (name, side)), str(d / ("%s_%s.pgm" % (name, side))))
        p = subprocess.run([exe, str(d / (name + "_left.pgm")), str(d / (name + "_right.pgm"))],
                           capture_output=True, text=True)
        assert p.returncode == 0 and "... done!" in p.stdout, p.stdout + p.stderr
        out[who] = [open(str(d / ("%s_%s_disp.pgm" % (name, side))), "rb").read() for side in ("left", "right")]
    for a, b in zip(out["ref"], out["gpu"]):
        assert len(a) == len(b)
        hdr = a.index(b"255\n") + 4
        assert a[:hdr] == b[:hdr]
        pa = np.frombuffer(a, np.uint8, offset=hdr); pb = np.frombuffer(b, np.uint8, offset=hdr)
        assert (pa > 0).mean() > 0.5
        assert (pa != pb).mean() < 1e-4, int((pa != pb).sum())


def test_usage_text():
    p = subprocess.run([TOOL], capture_output=True, text=True)
    assert p.returncode == 0 and "./elas left.pgm right.pgm .. process a single stereo pair" in p.stdout
