"""The CPU oracle (test infrastructure, plain C) under AddressSanitizer and UndefinedBehaviorSanitizer: a native
driver runs the extraction at five image sizes with both blur kernels, a top-2 match and the frame-level matchers
(initialisation search, grid queries, undistortion, distinctive descriptor, the three projection forms); any report fails."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_runs_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    build = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", "-std=c11", os.path.join(ROOT, "tests", "cpp", "oracle_san.c"),
                            os.path.join(ROOT, "oracle", "orb_oracle.c"), "-lm", "-o", exe],
                           capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    if "does not come first in initial library list" in run.stderr:
        pytest.skip("another preloaded library precedes the sanitizer runtime in this environment")
    assert run.returncode == 0, run.stdout + run.stderr
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr
    assert "1241x376 blur 0: rc 0, 2005 keypoints" in run.stdout
