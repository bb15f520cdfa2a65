#!/usr/bin/env python3
"""Generates tests/golden/costmap_track_converter.npz with the REFERENCE's own costmap writer.

Run ONCE in the build container (needs /root/reference and PIL); the outputs are committed.
It imports /root/reference/autorally_control/src/path_integral/scripts/track_converter.py and
calls gen_costmap(costmap_txt, image_name, output_name) (:6-34) on a small synthetic costmap in the
old text format ("x_min x_max y_min y_max pixelsPerMeter v0 v1 ... vN " -- the element after the last
space is dropped by the reference, :10).  Committed: the input text (costmap_input.txt) and the
.npz the reference wrote (data only; no reference source travels).  The file pins the costmap
*format* read by MPPICosts::loadTrackData (costs.cu:190-232): key names, dtypes, row-major order.
"""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
REF = "/root/reference/autorally_control/src/path_integral"
sys.path.insert(0, os.path.join(REF, "scripts"))

import numpy as np

import track_converter as ref_tc  # the reference's module

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    x_min, x_max, y_min, y_max, ppm = -3.0, 5.0, -2.0, 2.5, 4
    W, H = int((x_max - x_min) * ppm), int((y_max - y_min) * ppm)  # 32 x 18, costs.cu:204-205
    yy, xx = np.mgrid[0:H, 0:W]
    # an off-centre ring: low cost on the ring, 1 off it (row-major, x fastest)
    px = x_min + (xx + 0.5) / ppm
    py = y_min + (yy + 0.5) / ppm
    r = np.sqrt((px - 1.0) ** 2 + (py - 0.25) ** 2)
    ch0 = np.clip(np.abs(r - 1.5) / 0.8, 0.0, 1.0).astype(np.float32)
    txt = os.path.join(HERE, "costmap_input.txt")
    with open(txt, "w") as f:
        f.write("%r %r %r %r %r " % (x_min, x_max, y_min, y_max, ppm))
        f.write(" ".join("%.6f" % v for v in ch0.reshape(-1)))
        f.write(" ")  # the reference drops whatever follows the last space
    with tempfile.TemporaryDirectory() as d:
        ref_tc.gen_costmap(txt, os.path.join(d, "display.png"), os.path.join(HERE, "costmap_track_converter.npz"))
    z = np.load(os.path.join(HERE, "costmap_track_converter.npz"))
    print({k: (z[k].shape, str(z[k].dtype)) for k in z.files})


if __name__ == "__main__":
    main()
