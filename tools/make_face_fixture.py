#!/usr/bin/env python3
"""Regenerates ugproj-detector_amd/data/face64.u8: a 64x64 8-bit gray crop of the face in the
public-domain NASA astronaut portrait (Eileen Collins) that ships with scikit-image as
skimage/data/astronaut.png.  Used only as an input stimulus for synthetic 'planted' frames
(ugproj-detector_amd/synth.py); it is not an oracle and not part of the reference."""
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/opt/conda/lib/python3.9/site-packages/skimage/data/astronaut.png"
im = Image.open(src).convert("L")
# detector box on this image is about (176, 65, 98, 98); keep a 12 % margin around it
x, y, s = 176, 65, 98
m = 12
box = (x - m, y - m, x + s + m, y + s + m)
face = im.crop(box).resize((64, 64), Image.BOX)
np.asarray(face, np.uint8).tofile("ugproj-detector_amd/data/face64.u8")
print("wrote ugproj-detector_amd/data/face64.u8", np.asarray(face).mean())
