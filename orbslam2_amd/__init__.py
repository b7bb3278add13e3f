"""MI355X-native ORB front end of ORB-SLAM2: extractor + Hamming matcher behind a C ABI.

`ORBextractor` and `ORBmatcher` mirror the reference's classes of the same names
(include/ORBextractor.h, include/ORBmatcher.h of lianbin/orbslam2); all computation runs in
liborbx.so (hand-written HIP kernels for gfx950).
"""
from ._lib import KEYPOINT_DTYPE, OrbxError, lib  # noqa: F401
from .extractor import ORBextractor, Pipeline, synth_image  # noqa: F401
from .matcher import ORBmatcher, synth_descriptors  # noqa: F401
from .stereo import ComputeStereoMatches, compute_stereo_matches_device  # noqa: F401
from .vocabulary import ORBVocabulary  # noqa: F401
