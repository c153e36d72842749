"""The oracle's C code under AddressSanitizer + UBSan (the reference has
no sanitizer builds; the build plan asks for them on the C host code).  CPU only."""
import os
import subprocess

import hdtest


def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-s", "-C", hdtest.ORACLE_DIR, "san_harness"], check=True)
    p = subprocess.run([os.path.join(hdtest.ORACLE_DIR, "san_harness")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 failures" in p.stdout
