#!/usr/bin/env python3
"""Writes a "retrained" cascade for the plug-in tests: the packed haarcascade_frontalface_alt2 with the last leaf value
of its last stage scaled by 1.5 -- every result-deciding number but one equals the stock file, the hash differs, so the
library's built-in compiled stages must not be used for it.
usage: make_modified_cascade.py <stock.hcc> <out.hcc>"""
import sys

import numpy as np

raw = bytearray(open(sys.argv[1], "rb").read())
raw[-4:] = np.float32(np.frombuffer(bytes(raw[-4:]), np.float32)[0] * np.float32(1.5)).tobytes()
open(sys.argv[2], "wb").write(bytes(raw))
