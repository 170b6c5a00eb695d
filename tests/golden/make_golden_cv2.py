#!/usr/bin/env python3
"""Golden vectors for the two OpenCV steps either side of the hot path in the reference's video loop,
produced by CALLING cv2 ITSELF.  ``cv2`` is NOT in the build image, so this script cannot run there: it is committed so that the day
an image has OpenCV, ``python tests/golden/make_golden_cv2.py`` writes ``tests/golden/cv2.npz`` and the tests that today say
"parity unpinned" (tests/test_oracle_golden.py::test_frameio_restatement_vs_cv2_fixture, tests/test_hip_stream.py::
test_frameio_kernels_vs_cv2_fixture) start comparing oracle/frameio_ref.py and csrc/frameio.hip with OpenCV's own bytes.

cv2 calls made (exactly the reference's, main_new.py):
  * ``cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)``                                   :639, :654, :665
  * ``cv2.resize(gray, (256, 256), interpolation=cv2.INTER_AREA)``                :640, :655, :666
  * ``cv2.resize(out, (640, 360), interpolation=cv2.INTER_AREA)`` from 720p / 1080p / 854x480      :723
  * ``cv2.GaussianBlur(img, (3, 3), 0.2)`` on uint8 -- checked to be the identity at 8 bits       :724 (the build skips it)
Inputs are regenerated from seeds (pwstablenet_amd/synth.py, numpy RandomState); only seeds, shapes and cv2's OUTPUT bytes are
stored.  No reference source is copied.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SIZES = {"720p": (720, 1280), "1080p": (1080, 1920), "480p": (480, 854)}


def frame_u8(name, seed):
    """A decoded BGR frame: smooth structure + noise, so that area averages land on every rounding case."""
    h, w = SIZES[name]
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 127 + 90 * np.sin(xx / 37.0 + seed) * np.cos(yy / 23.0) + 30 * np.sin((xx + yy) / 11.0)
    img = base[..., None] + rs.randint(-40, 41, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    try:
        import cv2
    except ImportError:
        print("make_golden_cv2.py: cv2 is not importable in this image -- nothing written (the frame-io oracle stays 'parity unpinned')")
        return 1
    out = {"cv2_version": np.array(cv2.__version__), "seeds": np.array([11, 12, 13])}
    for (name, seed) in (("720p", 11), ("1080p", 12), ("480p", 13)):
        fr = frame_u8(name, seed)
        gray = cv2.cvtColor(fr, cv2.COLOR_BGR2GRAY)
        out["gray_" + name] = gray
        out["plane256_" + name] = cv2.resize(gray, (256, 256), interpolation=cv2.INTER_AREA)
        small = cv2.resize(fr, (640, 360), interpolation=cv2.INTER_AREA)
        out["out640x360_" + name] = small
        blur = cv2.GaussianBlur(small, (3, 3), 0.2)
        out["blur_is_identity_" + name] = np.array(bool(np.array_equal(blur, small)))
    np.savez_compressed(os.path.join(HERE, "cv2.npz"), **out)
    print("wrote", os.path.join(HERE, "cv2.npz"), "with cv2", cv2.__version__)
    return 0


if __name__ == "__main__":
    sys.exit(main())
