import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


# ORB parameter sets of the reference's shipped YAMLs (Examples/Monocular/TUM1.yaml:30-43,
# Examples/Stereo/KITTI00-02.yaml:38-51, Examples/Stereo/KITTI04-12.yaml:50, Examples/Stereo/EuRoC.yaml:88)
TUM = dict(nfeatures=1000, scale=1.2, nlevels=8, ini=20, minth=7)
KITTI = dict(nfeatures=2000, scale=1.2, nlevels=8, ini=20, minth=7)
KITTI_LOW = dict(nfeatures=2000, scale=1.2, nlevels=8, ini=12, minth=7)
EUROC_STEREO = dict(nfeatures=1200, scale=1.2, nlevels=8, ini=20, minth=7)
