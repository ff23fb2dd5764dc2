"""SURVEY section 5 ("ASAN build of the extension"), CPU box only: the HOST side of every .hip source - argument validation,
workspace plans, launch sequences - compiled with `hipcc -fsanitize=address,undefined -fno-gpu-sanitize` and driven through
tests/host_cases.py in a child interpreter with the sanitizer runtime preloaded.  (GPU AddressSanitizer is not available on
this pool; device code is covered by the parity tests.)"""
import glob
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ("api", "knn", "edgeconv", "proto", "pointwise", "attention", "head", "kmeanspp", "optim")


def _asan_runtime():
    found = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return found[-1] if found else None


def _build(outdir):
    os.makedirs(outdir, exist_ok=True)
    so = os.path.join(outdir, "libgfs3d_hip_san.so")
    srcs = [os.path.join(PKG, "csrc", s + ".hip") for s in SOURCES]
    deps = srcs + [os.path.join(PKG, "csrc", "common.h"), os.path.join(ROOT, "include", "gfs3d.h")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
        return so
    # host code instrumented, device code not (-fno-gpu-sanitize: GPU AddressSanitizer needs xnack+, unavailable here); the device
    # side is still compiled so that the module's fat binary exists and the library loads like the shipped one
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
             "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    jobs = []
    for src in srcs:
        obj = os.path.join(outdir, os.path.basename(src) + ".o")
        jobs.append(subprocess.Popen([HIPCC, *flags, "-c", src, "-o", obj]))
    assert all(j.wait() == 0 for j in jobs)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan", "-shared",
                           *[os.path.join(outdir, os.path.basename(s_) + ".o") for s_ in srcs], "-o", so])
    return so


def test_host_launch_code_is_clean_under_asan_and_ubsan():
    rt = _asan_runtime()
    if rt is None or not os.path.exists(HIPCC):
        pytest.skip("no hipcc / clang AddressSanitizer runtime on this box")
    so = _build(os.path.join(PKG, "lib", "san"))
    env = dict(os.environ, GFS3D_SAN_LIB=so, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=77",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=78")
    env.pop("PYTHONMALLOC", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san_driver.py")], env=env, capture_output=True, text=True, timeout=600)
    report = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "SAN_DRIVER_OK" in r.stdout, report
    assert "ERROR: AddressSanitizer" not in report and "runtime error:" not in report, report
