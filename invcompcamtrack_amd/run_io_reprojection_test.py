"""One-frame-pair driver, argv- and file-compatible with the reference's run_io_reprojection_test
(run_io_reprojection_test.cpp:99-334):

  python -m invcompcamtrack_amd.run_io_reprojection_test imgA imgB infile outfile lv_f lv_l psz maxiter \\
         normdp_ratio donorm dopatchnorm maxpttrack verbosity

infile = binary point/cam file, outfile = 6 x f64 (io_formats.py). verbosity == 1 repeats the tracking 1000 times
and prints the reference's timing line (run_io_reprojection_test.cpp:206-231); the pyramids stay on the GPU.
"""
from __future__ import annotations

import sys
import time

import numpy as np

from . import io_formats as iof
from .tracker import CamClass, OdometerClass, PoseClass, optparam, util_constructpyramide


def main(argv=None):
    a = sys.argv[1:] if argv is None else list(argv)
    if len(a) != 13:
        print(__doc__)
        return 2
    img_a, img_b, infile, outfile = a[0:4]
    lv_f, lv_l, psz, maxiter = int(a[4]), int(a[5]), int(a[6]), int(a[7])
    ratio, donorm, dpn, maxpt, verbosity = float(a[8]), int(a[9]), int(a[10]), int(a[11]), int(a[12])
    op = optparam(lv_f, lv_l, psz, maxiter, ratio, donorm, dpn, maxpt, verbosity)

    fa, fb = iof.read_image_gray(img_a), iof.read_image_gray(img_b)
    pyr_a = util_constructpyramide(fa, lv_f, 1, psz)  # run_io_reprojection_test.cpp:157-158
    pyr_b = util_constructpyramide(fb, lv_f, 1, psz)
    d = iof.read_pointcam_file(infile)
    cam = CamClass(lv_f + 1, d["fc"], d["cc"], d["wh"], psz)
    pose = PoseClass(cam, op)
    odo = OdometerClass(pose, op)
    reps = 1000 if verbosity == 1 else 1
    t0 = time.perf_counter()
    for _ in range(reps):
        pts = d["pts3d"].copy()  # the reference normalises its input array in place on every call as well
        odo.Set3Dpoints(pts)
        odo.SetPose(d["pose"], pyr_a, pyr_b)
        out = odo.TrackPose()
    if verbosity == 1:
        print("TIME (pose tracking) (musec): %3g" % ((time.perf_counter() - t0) * 1e3))  # ms for 1000 runs = us per run
    iof.write_pose_result(outfile, out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
