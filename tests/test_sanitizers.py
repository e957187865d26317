"""Sanitizer leg for the CPU pieces (the oracle, scene_prep.cpp, host_helpers.cpp): built with
-fsanitize=address,undefined and driven by tests/sanitize/host_san_main.cpp.  CPU only — never on the GPU box's device."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "sanitize"), "run", "OUT=%s" % tmp_path],
                         capture_output=True, text=True, timeout=900)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-4000:]
    assert "sanitizer driver: 0 failed checks" in text
    assert "ERROR: AddressSanitizer" not in text and "runtime error" not in text and "LeakSanitizer" not in text, text[-4000:]
