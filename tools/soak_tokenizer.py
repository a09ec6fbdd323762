"""Random vocabularies (pieces over a small alphabet, scores with many ties, some at the reference's -1e10 floor) and
random texts through the product's tokenizer (nvk_tokenizer_*) against the reference's own class
(oracle/_ref/libnvk_ref.so, built from /root/reference where that exists): identical token ids, identical pieces.
CPU only:    python tools/soak_tokenizer.py --first 0 --minutes 2"""
import argparse
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import nano_vllm_amd as nv  # noqa: E402
import oracle_lib as ol  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=0)
ap.add_argument("--minutes", type=float, default=2.0)
args = ap.parse_args()
if ol.REF is None or not hasattr(ol.REF, "ref_tokenizer_create"):
    sys.exit("oracle/_ref/libnvk_ref.so is not built here (it needs /root/reference)")

t0, seed, texts, bad = time.perf_counter(), args.first, 0, 0
with tempfile.TemporaryDirectory() as d:
    while time.perf_counter() - t0 < args.minutes * 60:
        rng = np.random.default_rng(seed)
        alphabet = bytes(rng.choice(np.frombuffer(b" abcdexyz.,\n\xc3\xa9\x7f", np.uint8), size=int(rng.integers(2, 9)), replace=False).tolist())
        # the 256 byte pieces <0x00>..<0xFF> at ids 3..258, as every llama2.c vocabulary has (without them the reference's
        # byte fallback indexes past its tables and faults: nothing to compare with)
        pieces = [b"<unk>", b"<s>", b"</s>"] + [b"<0x%02X>" % c for c in range(256)]
        pieces += [bytes([c]) for c in alphabet if rng.random() < 0.9]  # (a letter without a piece falls back to its byte, or to <unk>)
        merged = set()
        for _ in range(int(rng.integers(0, 60))):
            merged.add(bytes(rng.choice(np.frombuffer(alphabet, np.uint8), size=int(rng.integers(2, 7))).tolist()))
        first_merged = len(pieces)
        pieces += sorted(merged, key=lambda p: (rng.random(), p))
        levels = [-1.0, -2.0, -2.0, -3.5, -1e10, -3e10] if rng.random() < 0.7 else None
        scores = [0.0] * first_merged + [float(rng.choice(levels)) if levels else float(-rng.random() * 10) for _ in pieces[first_merged:]]
        path = os.path.join(d, "t.bin")
        with open(path, "wb") as fh:
            fh.write(struct.pack("i", max(len(p) for p in pieces)))
            for piece, score in zip(pieces, scores):
                fh.write(struct.pack("fi", score, len(piece)))
                fh.write(piece)
        tok = nv.Tokenizer(path, len(pieces))
        ref = ol.REF.ref_tokenizer_create(path.encode(), len(pieces))
        assert ref
        for i in range(-1, len(pieces) + 1):
            if tok.decode_bytes(i) != ol.ref_tokenizer_decode(ref, i):
                print(f"seed {seed}: decode({i}) differs", flush=True)
                bad += 1
        for _ in range(40):
            n = int(rng.integers(1, 90))
            text = bytes(rng.choice(np.frombuffer(alphabet + b"q", np.uint8), size=n).tolist())  # 'q': never in the vocabulary
            for bos, eos in ((True, False), (True, True), (False, True)):
                if not bos and not tok.encode(text, False, False):
                    continue  # no piece for any character and no BOS: the reference's merge loop underflows (tokenizer.hpp:111)
                got, want = tok.encode(text, bos, eos), ol.ref_tokenizer_encode(ref, text, bos, eos)
                texts += 1
                if got != want:
                    print(f"seed {seed}: {text!r} bos={bos} eos={eos}: product {got}, reference {want}", flush=True)
                    bad += 1
        ol.REF.ref_tokenizer_destroy(ref)
        seed += 1
print(f"seeds {args.first}..{seed - 1}: {seed - args.first} vocabularies, {texts} encodings against the reference's Tokenizer: {bad} differences "
      f"({time.perf_counter() - t0:.0f} s)")
sys.exit(1 if bad else 0)
