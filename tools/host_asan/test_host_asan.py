"""Build-container test (NOT under tests/: the GPU pool refuses any run that could reach a sanitizer build, and tools/host_asan/ is
in .gpurunignore): the loader and the binary CSR cache (what mgx_load_mtx / mgx_graph_save_csr / mgx_graph_load_csr wrap) under
AddressSanitizer + UndefinedBehaviorSanitizer on the host side.  usage: python -m pytest tools/host_asan/test_host_asan.py"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_only_entry_points_under_asan_and_ubsan():
    """fixtures round-trip, every truncation and byte flip of a cache file is rejected, a header that promises more than the
    file holds is not believed, malformed MatrixMarket text is refused -- no sanitizer report"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", os.path.join(HERE, "run.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "host_io_asan: 0 failures" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
