"""The tokenizer behind the C ABI (nvk_tokenizer_*) against the reference's Tokenizer:
bit-exact token ids and pieces on the committed fixture F8 (recorded from include/core/tokenizer.hpp through
oracle/ref_shim.cpp) and, where the reference library is built, on seeded random texts.  Host-only: no GPU."""
import json
import os

import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from golden_defs import TOKENIZER_TEXTS, tokenizer_spec, write_tokenizer
from nano_vllm_amd.workloads import splitmix64


@pytest.fixture(scope="module")
def tokenizer_path(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("tok") / "tokenizer.bin")
    write_tokenizer(path, 512)
    return path


def test_fixture_f8_encode_and_decode(tokenizer_path, golden_dir):
    with open(os.path.join(golden_dir, "tokenizer_cases.json")) as fh:
        gold = json.load(fh)
    tok = nv.Tokenizer(tokenizer_path, gold["vocab_size"])
    pieces, _, max_len = tokenizer_spec(512)
    assert tok.vocab_size == 512 and tok.max_token_length == max_len
    for case in gold["encode"]:
        text = bytes.fromhex(case["text_hex"])
        assert tok.encode(text, case["bos"], case["eos"]) == case["ids"], text
    for i, want in zip(range(-1, 513), gold["decode_hex"]):
        assert tok.decode_bytes(i).hex() == want, i
    assert tok.decode_bytes(3 + 0x41) == b"A"  # "<0x41>" is the byte itself (tokenizer.hpp:70-73)
    assert tok.decode_bytes(-5) == b"" and tok.decode_bytes(512) == b""
    assert tok.encode("hello world") == tok.encode(b"hello world")


def test_round_trip_of_ascii_text(tokenizer_path):
    """decode(encode(text)) gives back ' ' + text for text made of pieces the vocabulary has (the dummy prefix)."""
    tok = nv.Tokenizer(tokenizer_path, 512)
    for text in TOKENIZER_TEXTS:
        if not text or any(c >= 0x80 for c in text):
            continue
        ids = tok.encode(text, bos=False)
        assert b"".join(tok.decode_bytes(i) for i in ids) == b" " + text


def test_errors(tokenizer_path, tmp_path):
    with pytest.raises(nv.NvkError):
        nv.Tokenizer(str(tmp_path / "missing.bin"), 512)
    with pytest.raises(nv.NvkError):
        nv.Tokenizer(tokenizer_path, 100000)  # the file ends before that many pieces


@pytest.mark.skipif(ol.REF is None or not hasattr(ol.REF, "ref_tokenizer_create"), reason="oracle/_ref not built here")
def test_random_texts_against_the_reference(tokenizer_path):
    tok = nv.Tokenizer(tokenizer_path, 512)
    ref = ol.REF.ref_tokenizer_create(tokenizer_path.encode(), 512)
    assert ref
    alphabet = b" etaoinshrdlu .,?!\n" + bytes([0xC3, 0xA9, 0x00, 0x7F])
    for seed in range(200):
        n = int(splitmix64(5000 + seed, np.array([0], np.uint64))[0] % 60)
        draws = splitmix64(6000 + seed, np.arange(n, dtype=np.uint64))
        text = bytes(alphabet[int(d % len(alphabet))] for d in draws)
        for bos, eos in ((True, False), (False, True)):
            if not bos and n == 0:
                continue  # the reference's merge loop underflows on an empty list (tokenizer.hpp:111)
            assert tok.encode(text, bos, eos) == ol.ref_tokenizer_encode(ref, text, bos, eos), (seed, text)
    ol.REF.ref_tokenizer_destroy(ref)


@pytest.mark.skipif(ol.REF is None or not hasattr(ol.REF, "ref_tokenizer_create"), reason="oracle/_ref not built here")
def test_tied_scores_and_floor_scores_against_the_reference(tmp_path):
    """The merge order where it is decided by POSITION: a vocabulary whose merged pieces all share two scores (so most
    rounds are ties, resolved to the leftmost pair, tokenizer.hpp:116 compares with >), overlapping candidates
    ("aa" in "aaaa..."), and pieces scored at / below the -1e10 floor, which the reference never merges."""
    import struct
    singles = [bytes([c]) for c in b" ab"]
    merged = [b"aa", b"ab", b"ba", b"bb", b" a", b"a ", b"aaa", b"aba", b"bab", b"aab", b"abab", b"aaaa", b" b", b"b "]
    pieces = [b"<unk>", b"<s>", b"</s>"] + singles + merged + [b"bbb", b"  "]
    scores = [0.0] * (3 + len(singles)) + [(-1.0 if i % 3 else -2.0) for i in range(len(merged))] + [-1e10, -3e10]
    path = str(tmp_path / "ties.bin")
    with open(path, "wb") as fh:
        fh.write(struct.pack("i", max(len(p) for p in pieces)))
        for piece, score in zip(pieces, scores):
            fh.write(struct.pack("fi", score, len(piece)))
            fh.write(piece)
    tok = nv.Tokenizer(path, len(pieces))
    ref = ol.REF.ref_tokenizer_create(path.encode(), len(pieces))
    assert ref
    for seed in range(300):
        n = 1 + int(splitmix64(7000 + seed, np.array([0], np.uint64))[0] % 48)
        draws = splitmix64(8000 + seed, np.arange(n, dtype=np.uint64))
        text = bytes(b" aabb"[int(d % 5)] for d in draws)
        assert tok.encode(text, True, False) == ol.ref_tokenizer_encode(ref, text, True, False), (seed, text)
    for text in (b"a" * 33, b"ab" * 20, b"b" * 9, b"  " * 5, b"abba" * 7):
        assert tok.encode(text, True, True) == ol.ref_tokenizer_encode(ref, text, True, True), text
    ol.REF.ref_tokenizer_destroy(ref)
