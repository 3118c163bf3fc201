"""MI355X-native Gauss-Newton photometric camera tracker (drop-in for catree/InvCompCamTrack's
per-frame tracking hot path). See DESIGN.md. The HIP library is loaded lazily by ``_lib.load()``;
importing this package never touches the GPU and never falls back to a CPU implementation."""
from . import _lib  # noqa: F401
from ._lib import IctrError  # noqa: F401
from .tracker import (CamClass, OdometerClass, PoseClass, Pyramid, TrackBatch, device_count, locality_order, ncc_score, optparam,  # noqa: F401
                      timebase_mark,
                      solve6,
                      util_constructpyramide, util_getPatch, util_getPatch_grad, util_SE3_coeff_to_group,
                      util_SE3_group_to_coeff)

__version__ = "0.1.0"
