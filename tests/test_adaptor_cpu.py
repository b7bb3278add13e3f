"""The cv:: adaptor (orbslam2_amd/cpp/ORBextractor.{h,cc}) is the one product file that needs OpenCV, which
this image lacks.  These tests type-check and link it against the test-only cv:: stand-in under
tests/cpp/cv_stub (declarations of Mat / KeyPoint / InputArray / OutputArray, nothing more), with the language
level of the reference's own build (-std=c++11, CMakeLists.txt:10-18), and check that the code the integration
guide shows is the code that is compiled."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I", os.path.join(ROOT, "tests", "cpp", "cv_stub"), "-I", os.path.join(ROOT, "include"),
       "-I", os.path.join(ROOT, "orbslam2_amd", "cpp")]


def run(cmd):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, " ".join(cmd) + "\n" + out.stdout + out.stderr


def test_adaptor_type_checks_as_cxx11():
    # includes static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint)) and ::orbx_extractor *handle()
    run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", *INC,
         os.path.join(ROOT, "orbslam2_amd", "cpp", "ORBextractor.cc")])
    # the header alone, as Frame.h / Tracking.h include it
    run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", *INC,
         os.path.join(ROOT, "orbslam2_amd", "cpp", "ORBextractor.h")])


def test_frame_style_caller_and_integration_snippets_compile_and_link(tmp_path):
    lib = os.path.join(ROOT, "orbslam2_amd", "liborbx.so")
    assert os.path.exists(lib), "build liborbx.so first (__graft_entry__.build())"
    run(["g++", "-std=c++11", "-Wall", *INC, os.path.join(ROOT, "tests", "cpp", "test_adaptor.cc"),
         os.path.join(ROOT, "orbslam2_amd", "cpp", "ORBextractor.cc"), "-L", os.path.dirname(lib), "-lorbx", "-pthread",
         "-o", str(tmp_path / "test_adaptor")])


GUIDE = "INTEGRATION.md"  # the integration guide at the repository root


def test_integration_guide_shows_the_compiled_snippets():
    norm = lambda t: re.sub(r"\s+", " ", t).strip()
    compiled = norm(open(os.path.join(ROOT, "tests", "cpp", "integration_snippets.inc")).read())
    md = open(os.path.join(ROOT, GUIDE)).read()
    blocks = re.findall(r"<!-- compiled: tests/cpp/integration_snippets.inc -->\s*```cpp\n(.*?)```", md, re.S)
    assert len(blocks) >= 2, GUIDE + " must mark the DescriptorDistance and ComputeStereoMatches bodies"
    for b in blocks:
        assert norm(b) in compiled, GUIDE + " snippet differs from the compiled one:\n" + b
