"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (`make asan`; CPU only, no GPU sanitizer exists on
this pool): the block allocator and the tokenizer of the library, the CPU checker, and the command line's JSON
reader / argument handling, on randomized traces and on short or damaged input files.  The reference reads its
tokenizer file unchecked (include/core/tokenizer.hpp:31-61); the library must fail cleanly instead."""
import json
import os
import shutil
import subprocess

import pytest

from golden_defs import write_tokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "build", "asan", "asan_driver")
RUN = os.path.join(ROOT, "build", "asan", "nvk_run_asan")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0",
           UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def built():
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host compiler")
    r = subprocess.run(["make", "-s", "-C", ROOT, "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return True


def clean(r, ok_codes=(0,)):
    text = (r.stderr or b"").decode(errors="replace")
    assert "Sanitizer" not in text and "runtime error" not in text, text[-3000:]
    assert r.returncode in ok_codes, (r.returncode, text[-1000:])


def test_block_manager_and_checker_under_sanitizers(built):
    clean(subprocess.run([DRIVER, "bm"], capture_output=True, env=ENV, timeout=300))
    clean(subprocess.run([DRIVER, "attn"], capture_output=True, env=ENV, timeout=300))


def test_tokenizer_under_sanitizers_good_and_damaged_files(built, tmp_path):
    good = str(tmp_path / "tokenizer.bin")
    write_tokenizer(good, 512)
    clean(subprocess.run([DRIVER, "tok", good, "512"], capture_output=True, env=ENV, timeout=300))
    data = open(good, "rb").read()
    cases = {"empty": b"", "header_only": data[:4], "cut_in_a_length": data[:4 + 4 + 2], "cut_in_a_piece": data[:40],
             "half": data[: len(data) // 2], "one_byte_short": data[:-1],
             "huge_length": data[:8] + (2 ** 31 - 1).to_bytes(4, "little") + data[12:],
             "negative_length": data[:8] + (-5).to_bytes(4, "little", signed=True) + data[12:],
             "negative_max_len": (-1).to_bytes(4, "little", signed=True) + data[4:]}
    for name, blob in cases.items():
        path = str(tmp_path / f"{name}.bin")
        with open(path, "wb") as fh:
            fh.write(blob)
        clean(subprocess.run([DRIVER, "tokbad", path, "512"], capture_output=True, env=ENV, timeout=120))
    clean(subprocess.run([DRIVER, "tokbad", good, "100000"], capture_output=True, env=ENV, timeout=120))  # asks for more pieces than the file has


def test_command_line_json_reader_under_sanitizers(built, tmp_path):
    """Malformed request files and arguments: an error exit (the reference's messages), no sanitizer report.  The
    JSON is read before any device is touched, so this runs without a GPU."""
    model_dir = tmp_path / "m"
    model_dir.mkdir()
    (model_dir / "model.bin").write_bytes(b"\0" * 28)
    write_tokenizer(str(model_dir / "tokenizer.bin"), 512)
    corpus = ["", "{", "}", "[]", "null", '{"requests": 5}', '{"requests": [}', '{"requests": [{"prompt": "a" "max_tokens": 3}]}',
              '{"requests": [{"prompt": "", "max_tokens": 3}]}', '{"requests": [{"prompt": "x", "max_tokens": -7}]}',
              '{"requests": [{"prompt": "\\u12', '{"requests": [{"prompt": "\\', '{"requests": [{"prompt": "x", "temperature": 1e999}]}',
              '{"requests": [' + '[' * 5000, '{"requests": [{"prompt": "' + "y" * 100000 + '"}]}', '{"a": {"b": {"c": [1, 2, {"d": null}]}}}',
              '{"requests": [{"prompt": "ok", "max_tokens": 3, "temperature": 0}]} trailing', "\xff\xfe\x00garbage", '{"requests": [{"prompt": 5}]}',
              json.dumps({"requests": [{"prompt": "p", "temperature": 0.7, "max_tokens": 4}]})]
    for i, text in enumerate(corpus):
        path = tmp_path / f"req{i}.json"
        path.write_bytes(text.encode("utf-8", errors="surrogateescape") if isinstance(text, str) else text)
        r = subprocess.run([RUN, str(model_dir), "--input-json", str(path)], capture_output=True, env=ENV, timeout=120)
        clean(r, ok_codes=(1, 2))  # every one of these is an error of some kind: bad JSON, no sampler, no device
    for args in ([], [str(model_dir)], [str(model_dir), "-i"], [str(model_dir), "-i", "x", "-n"], ["-t", "0"],
                 [str(tmp_path / "nowhere"), "-i", "x"], [str(model_dir), "-i", "x", "--bogus"]):
        clean(subprocess.run([RUN, *args], capture_output=True, env=ENV, timeout=120), ok_codes=(1, 2))
