#!/usr/bin/env python3
"""Generates tests/golden/costmap_track_generator.npz with the REFERENCE's image-to-costmap writer.

Run ONCE in the build container (needs /root/reference and PIL); inputs and output are committed.
It imports /root/reference/autorally_control/src/path_integral/scripts/track_generator.py and calls
gen_costmap(input_img, config_file, output_name) (:6-42) on a small synthetic RGBA image with a
configuration that exercises every knob of the script (rotation, per-channel offset / normaliser,
channel permutation, vertical flip).  Committed: the input image (costmap_image.png), the
configuration (costmap_image_config.txt) and the .npz the reference wrote -- data only.  Unlike the
text converter's output, all four channels of this file are populated, which pins the channel order
of the float4 texture MPPICosts::loadTrackData builds (costs.cu:207-222).
"""
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference/autorally_control/src/path_integral"
sys.path.insert(0, os.path.join(REF, "scripts"))

import numpy as np
from PIL import Image

import track_generator as ref_tg  # the reference's module

HERE = os.path.dirname(os.path.abspath(__file__))

CONFIG = {
    "imageRotation": 180,
    "rOffset": 0.0, "rNormalizer": 255.0,
    "gOffset": -10.0, "gNormalizer": 100.0,
    "bOffset": 5.0, "bNormalizer": 50.0,
    "aOffset": 0.0, "aNormalizer": 1.0,
    "channelMap": [2, 0, 3, 1],
    "flip": True,
    "xBounds": [-2.0, 3.0],
    "yBounds": [-1.0, 2.0],
    "pixelsPerMeter": 4,
}


def main():
    W = int((CONFIG["xBounds"][1] - CONFIG["xBounds"][0]) * CONFIG["pixelsPerMeter"])  # 20
    H = int((CONFIG["yBounds"][1] - CONFIG["yBounds"][0]) * CONFIG["pixelsPerMeter"])  # 12
    rng = np.random.RandomState(11)
    img = rng.randint(0, 256, size=(H, W, 4)).astype(np.uint8)
    png = os.path.join(HERE, "costmap_image.png")
    Image.fromarray(img, mode="RGBA").save(png)
    cfg = os.path.join(HERE, "costmap_image_config.txt")
    with open(cfg, "w") as f:
        f.write(repr(CONFIG) + "\n")
    out = os.path.join(HERE, "costmap_track_generator.npz")
    ref_tg.gen_costmap(png, cfg, out)
    z = np.load(out)
    print({k: (z[k].shape, str(z[k].dtype)) for k in z.files})


if __name__ == "__main__":
    main()
