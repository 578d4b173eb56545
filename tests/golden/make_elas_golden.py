"""Golden vectors for the dense ELAS row, produced by the REFERENCE ITSELF: the vendored
libelas of /root/reference compiled by oracle/Makefile.ref (oracle/_ref/libref_elas.so) and called through
oracle/ref_elas_wrap.cpp.  Run in the build container (needs /root/reference at build time):
    python tests/golden/make_elas_golden.py
Writes tests/golden/elas_vectors.npz: for crops of the urban1 pair (data fixture in this directory) and both
parameter sets, the support points, the triangle lists (canonical order, see tests/test_elas_delaunay.py), and
the final disparity maps D1/D2 of the reference's stages run on those lists, plus (as a sparse difference)
the maps of the untouched Elas::process.  Inputs are re-derived from urban1_*.pgm by tests/util.py, so only outputs are stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import util  # noqa: E402
from oracle import binding as ob  # noqa: E402

CROPS = {"small": (640, 240, 300, 60), "tiny": (97, 64, 600, 150)}


def main():
    out = {}
    for name, crop in CROPS.items():
        L, R = util.urban_pair(*crop)
        for mb in (0, 1):
            p = ob.ref_elas_params(bool(mb))
            r0 = ob.ref_elas_staged(L, R, p)
            t1, t2 = ob.canonical_triangles(r0["tri1"]), ob.canonical_triangles(r0["tri2"])
            r = ob.ref_elas_staged(L, R, p, tri1=t1, tri2=t2)
            D1, D2 = ob.ref_elas(L, R, p)
            k = "%s_%d_" % (name, mb)
            out[k + "crop"] = np.array(crop, np.int32)
            out[k + "support"] = r["support"]
            out[k + "tri1"] = t1; out[k + "tri2"] = t2
            out[k + "D1"] = r["D1"]; out[k + "D2"] = r["D2"]
            for side, (a, b) in (("1", (r["D1"], D1)), ("2", (r["D2"], D2))):   # untouched Elas::process: sparse diff
                idx = np.flatnonzero(a.ravel() != b.ravel()).astype(np.int32)
                out[k + "process_diff_idx" + side] = idx
                out[k + "process_diff_val" + side] = b.ravel()[idx]
            print(name, mb, len(r["support"]), len(t1), len(t2), int((r["D1"] != D1).sum()), int((r["D2"] != D2).sum()))
    np.savez_compressed(os.path.join(HERE, "elas_vectors.npz"), **out)


if __name__ == "__main__":
    main()
