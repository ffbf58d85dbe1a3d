"""The C++ WordPiece tokenizer under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on
the pool; the tokenizer is the part of the C ABI that parses untrusted bytes): tests/native/tokenizer_fuzz_main.cpp feeds it ~3 000
random mixtures of valid text, invalid / truncated UTF-8, NULs and over-long words at every max_len edge through both entry points."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_tokenizer_fuzz_under_asan_ubsan(tmp_path):
    from rassengine_amd.encoder import synthetic_vocab
    vocab = tmp_path / "vocab.txt"
    vocab.write_text("\n".join(synthetic_vocab(3000)) + "\n", encoding="utf-8")
    exe = str(tmp_path / "tokenizer_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-pthread", "-o", exe, os.path.join(ROOT, "tests", "native", "tokenizer_fuzz_main.cpp"),
           os.path.join(ROOT, "rassengine_amd", "csrc", "tokenizer.cpp")]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe, str(vocab)], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert run.stdout.startswith("ok ")
