"""Writes tests/golden/tiny_420.jpg, tiny_gray.jpg and tiny_prog.jpg (seeds of the sanitizer fuzz of the JPEG decoder): Pillow-
encoded JPEGs of a synthetic 40x27 image — baseline 4:2:0 with restart markers, baseline greyscale, progressive 4:2:0.
Data, not reference material."""
import os
import numpy as np
from PIL import Image
here = os.path.dirname(os.path.abspath(__file__))
yy, xx = np.mgrid[0:27, 0:40]
img = np.stack([128 + 100 * np.sin(xx / 5.0) * np.cos(yy / 4.0), 128 + 90 * np.cos(xx / 7.0 + yy / 3.0), 40 + 5 * xx + yy], -1).clip(0, 255).astype(np.uint8)
Image.fromarray(img).save(os.path.join(here, "tiny_420.jpg"), "JPEG", quality=80, subsampling=2, restart_marker_blocks=2)
Image.fromarray(img[..., 1]).save(os.path.join(here, "tiny_gray.jpg"), "JPEG", quality=70, optimize=True)
Image.fromarray(img).save(os.path.join(here, "tiny_prog.jpg"), "JPEG", quality=75, subsampling=2, progressive=True)
