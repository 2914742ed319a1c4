"""The CMake route (CMakeLists.txt; reference: CMakeLists.txt:36-44, apps/CMakeLists.txt:1-6): configure + build in a scratch
directory (hipcc cross-compiles gfx950 without a GPU, ~20 s), the exported INTERFACE target `raytracer::raytracer` links the
applications, and - where the reference is mounted - the reference's OWN apps/ directory (its CMakeLists.txt and app.cu,
unchanged) builds against this tree."""
import ctypes
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_APPS = "/root/reference/apps"

pytestmark = pytest.mark.skipif(shutil.which("cmake") is None or shutil.which("ninja") is None or not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="needs cmake, ninja and the ROCm toolchain")


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    return p.stdout


def test_cmake_builds_libmirt_and_the_apps(tmp_path):
    from raytracer_amd import capi

    b = str(tmp_path / "b")
    _run(["cmake", "-S", ROOT, "-B", b, "-G", "Ninja"])
    ninja = open(os.path.join(b, "build.ninja")).read()
    assert "--offload-arch=gfx950" in ninja and "-ffp-contract=off" in ninja and "gfx906" not in ninja
    _run(["cmake", "--build", b, "-j", "6"])
    lib = ctypes.CDLL(os.path.join(b, "libmirt.so"))
    assert not [s for s in capi.SYMBOLS if not hasattr(lib, s)]
    for exe in ("app", "render", "obj_info"):
        assert os.access(os.path.join(b, "apps", exe), os.X_OK)
    if os.path.exists(os.path.join(REF_APPS, "CMakeLists.txt")):
        # the reference's apps directory, as it lies: `add_executable(app app.cu)` + `target_link_libraries(app PRIVATE raytracer fmt::fmt)`
        os.remove(os.path.join(b, "apps", "app"))
        _run(["cmake", "-S", ROOT, "-B", b, "-G", "Ninja", f"-DMIRT_APPS_DIR={REF_APPS}"])
        _run(["cmake", "--build", b, "-j", "6", "--target", "app"])
        assert os.access(os.path.join(b, "apps", "app"), os.X_OK)
        needed = subprocess.run(["ldd", os.path.join(b, "apps", "app")], stdout=subprocess.PIPE, text=True).stdout
        assert "libmirt.so" in needed
