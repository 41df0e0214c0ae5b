"""The host-side native code that parses untrusted text (clip_amd/csrc/tokenizer.cpp, include/clip_tok.h) under
AddressSanitizer + UndefinedBehaviorSanitizer: a standalone fuzz driver (tests/helpers/tok_fuzz.cpp) is compiled
together with the tokenizer source by g++ and run as a plain executable (host code only; nothing here touches a GPU
or the product .so)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_tokenizer_fuzz_under_asan_ubsan(tmp_path):
    exe = tmp_path / "tok_fuzz"
    cmd = ["g++", "-O1", "-g", "-std=c++20", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "helpers", "tok_fuzz.cpp"),
           os.path.join(ROOT, "clip_amd", "csrc", "tokenizer.cpp"), "-o", str(exe)]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ)
    # the environment's own preloads stay; the sanitizer runtime need not be first in the library list
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0:verify_asan_link_order=0"
    r = subprocess.run([str(exe), os.path.join(ROOT, "tests", "golden", "vocab.txt"), "400"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "no finding" in r.stdout
