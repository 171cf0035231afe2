"""CPU: ASan + UBSan over the oracle and over the kernel source compiled for the host (the GPU pool has no
device AddressSanitizer; the kernel's table/ring/LUT indexing is this same source)."""
import os
import subprocess

import pytest

from conftest import GOLDEN_REF, REF_CASES, ROOT
from test_oracle_golden import load_vectors


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    out = str(d / "sanitize_main")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-Wno-unknown-pragmas", "-x", "c", os.path.join(ROOT, "oracle", "pz_oracle.c"), "-x", "c++",
           os.path.join(ROOT, "tests", "model", "model_harness.cpp"), os.path.join(ROOT, "tests", "cxx", "sanitize_main.cpp"),
           "-o", out]
    subprocess.check_call(cmd)
    return out, d


@pytest.mark.parametrize("rb,cap", [(15, 1 << 21), (11, 1 << 21), (12, 1000), (11, 0)])
def test_oracle_and_kernel_model_under_asan_ubsan(exe, rb, cap):
    out, d = exe
    files = [os.path.join(GOLDEN_REF, n + ".z") for n in REF_CASES]
    for v in load_vectors():
        p = d / (v["name"] + ".z")
        if not p.exists():
            p.write_bytes(bytes.fromhex(v["z"]))
        files.append(str(p))
    r = subprocess.run([out, str(cap), str(rb)] + files, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 mismatches" in r.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_helper_thread_pool_under_sanitizers(tmp_path, san):
    """The persistent helper pool of the host-pointer paths (pure_zlib_amd/csrc/pzg_helpers.h; round 4): four threads call run()
    at once with jobs of 1-37 parts, every part runs exactly once and its writes are visible when run() returns; under
    ThreadSanitizer and under ASan + UBSan."""
    out = str(tmp_path / "helpers_stress")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-pthread",
                           os.path.join(ROOT, "tests", "cxx", "helpers_stress.cpp"), "-o", out])
    for workers in ("8", "1", "24"):
        r = subprocess.run([out, workers, "150"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "0 error(s)" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout[-500:] + r.stderr[-3000:]
