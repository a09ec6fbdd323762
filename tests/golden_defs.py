"""Definitions shared by oracle/make_golden.py (which records the reference's
answers) and the tests (which replay them): the list of cases, how their inputs
are derived from seeds, and the BlockManager op scripts.  Data and seeds only.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Dict, List

import numpy as np

from nano_vllm_amd.workloads import shuffled_permutation, splitmix64, uniform

# ------------------------------------------------------------------ F2 cases
# (n_heads, n_kv_heads, head_dim, block_size, num_tokens[, q_gain])
_RAW = [
    # G-tiny geometry, the default stories15M shape
    (6, 6, 48, 16, 1), (6, 6, 48, 16, 15), (6, 6, 48, 16, 16), (6, 6, 48, 16, 17), (6, 6, 48, 16, 37),
    (6, 6, 48, 16, 256),
    # G-llama
    (32, 8, 128, 16, 1), (32, 8, 128, 16, 16), (32, 8, 128, 16, 2048),
    # G-mha
    (32, 32, 128, 16, 513),
    (8, 2, 64, 32, 100),
    # beyond the survey's list: other block sizes (incl. non power of two), head dims, group sizes, peaked softmax
    (8, 4, 128, 8, 77), (4, 4, 32, 4, 9), (12, 3, 64, 12, 50), (4, 2, 256, 16, 40), (8, 8, 80, 16, 33),
    (16, 2, 64, 16, 70), (6, 2, 16, 16, 45), (32, 8, 128, 16, 300, 6.0), (6, 6, 48, 16, 200, 8.0),
    (32, 8, 128, 16, 0), (5, 5, 20, 16, 23), (16, 1, 128, 16, 129),
]

ATTENTION_CASES: List[Dict] = []
for _i, _c in enumerate(_RAW):
    H, KV, hd, bs, T = _c[:5]
    ATTENTION_CASES.append({
        "name": f"h{H}_kv{KV}_d{hd}_bs{bs}_t{T}" + (f"_g{int(_c[5])}" if len(_c) > 5 else ""),
        "index": _i, "n_heads": H, "n_kv_heads": KV, "head_dim": hd, "block_size": bs, "num_tokens": T,
        "q_gain": float(_c[5]) if len(_c) > 5 else 1.0,
    })


def attention_case_inputs(case):
    """q [H, hd], k_pool/v_pool [num_blocks, bs, KV, hd], table [ceil(T/bs)] for a case.

    The pool has a few more blocks than the sequence needs and the table is a
    seeded shuffle of it, so logical order != physical order; every pool element
    (used or not) holds a distinct seeded value."""
    i, H, KV, hd, bs, T = case["index"], case["n_heads"], case["n_kv_heads"], case["head_dim"], \
        case["block_size"], case["num_tokens"]
    need = (T + bs - 1) // bs
    num_blocks = need + 3
    perm = shuffled_permutation(num_blocks, 7000 + i)
    table = perm[:max(need, 1)].astype(np.int32)
    n = num_blocks * bs * KV * hd
    k_pool = uniform(1000 + i, 0, n).reshape(num_blocks, bs, KV, hd)
    v_pool = uniform(2000 + i, 0, n).reshape(num_blocks, bs, KV, hd)
    q = (uniform(3000 + i, 0, H * hd) * np.float32(case["q_gain"])).reshape(H, hd)
    return q, k_pool, v_pool, table


# ------------------------------------------------------------------ F6 cases
# chunked-prefill attention: (n_heads, n_kv_heads, head_dim, start_pos, chunk_size)
PREFILL_CASES = []
for _i, (_H, _KV, _hd, _start, _C) in enumerate([(6, 6, 48, 0, 37), (6, 6, 48, 40, 16), (32, 8, 128, 0, 64),
                                                  (32, 8, 128, 100, 33), (8, 2, 64, 7, 70), (4, 4, 32, 0, 1),
                                                  (16, 1, 128, 31, 34), (12, 3, 64, 64, 64)]):
    PREFILL_CASES.append({"name": f"h{_H}_kv{_KV}_d{_hd}_s{_start}_c{_C}", "index": _i, "n_heads": _H,
                          "n_kv_heads": _KV, "head_dim": _hd, "start_pos": _start, "chunk_size": _C})


def prefill_case_inputs(case):
    """q [C, H, hd] and the contiguous cache rows k_lin/v_lin [start+C, KV, hd] of a chunked-prefill case."""
    i, H, KV, hd = case["index"], case["n_heads"], case["n_kv_heads"], case["head_dim"]
    T = case["start_pos"] + case["chunk_size"]
    q = uniform(6000 + i, 0, case["chunk_size"] * H * hd).reshape(case["chunk_size"], H, hd)
    k_lin = uniform(6100 + i, 0, T * KV * hd).reshape(T, KV, hd)
    v_lin = uniform(6200 + i, 0, T * KV * hd).reshape(T, KV, hd)
    return q, k_lin, v_lin


# F7: the head of a layer (model.hpp:161-166): rms_norm, then the q / k / v projections.  (batch, dim, q_dim, kv_dim)
QKV_CASES = []
for _i, (_B, _D, _QD, _KD) in enumerate([(1, 288, 288, 288), (5, 288, 288, 288), (3, 64, 40, 8), (33, 128, 96, 32),
                                          (32, 512, 256, 64), (70, 256, 384, 128), (200, 96, 130, 30)]):
    QKV_CASES.append({"name": f"b{_B}_d{_D}_q{_QD}_kv{_KD}", "index": _i, "batch": _B, "dim": _D, "q_dim": _QD,
                      "kv_dim": _KD})


def qkv_case_inputs(case):
    """x [B, dim], rms weight [dim], wq [q_dim, dim], wk / wv [kv_dim, dim] of a QKV case (weights ~ U(-1,1)/sqrt(dim))."""
    i, B, D, QD, KD = case["index"], case["batch"], case["dim"], case["q_dim"], case["kv_dim"]
    x = (uniform(7000 + i, 0, B * D) * np.float32(2)).reshape(B, D)
    g = uniform(7100 + i, 0, D) * np.float32(0.5) + np.float32(1)
    sc = np.float32(1.0 / np.sqrt(D))
    wq = (uniform(7200 + i, 0, QD * D) * sc).reshape(QD, D)
    wk = (uniform(7300 + i, 0, KD * D) * sc).reshape(KD, D)
    wv = (uniform(7400 + i, 0, KD * D) * sc).reshape(KD, D)
    return x, g, wq, wk, wv


def paged_from_contiguous(k_lin, v_lin, block_size, seed, spare=3):
    """The same rows behind a shuffled block table: returns (k_pool, v_pool, table)."""
    T, KV, hd = k_lin.shape
    need = (T + block_size - 1) // block_size
    nb = need + spare
    table = shuffled_permutation(nb, seed)[:max(need, 1)].astype(np.int32)
    k_pool = np.zeros((nb, block_size, KV, hd), np.float32)
    v_pool = np.zeros_like(k_pool)
    for t in range(T):
        k_pool[table[t // block_size], t % block_size] = k_lin[t]
        v_pool[table[t // block_size], t % block_size] = v_lin[t]
    return k_pool, v_pool, table


# ------------------------------------------------------------ F1 BM scripts
def bm_trace_script():
    scripts = []
    ops = [["alloc"]] * 9 + [["num_free"], ["util"], ["free", 3], ["free", 3], ["free", 5], ["is_free", 3],
                              ["is_free", 4], ["is_free", -1], ["is_free", 8], ["alloc"], ["alloc"], ["alloc"],
                              ["free", -1], ["free", 8], ["free", 0], ["free", 7], ["alloc"], ["alloc"], ["util"],
                              ["num_free"]]
    scripts.append({"name": "fill_exhaust_reuse", "num_blocks": 8, "block_size": 16, "ops": ops})

    ops = [["alloc_seq", 9], ["num_free"], ["alloc_seq", 30], ["num_free"], ["alloc_seq", 0], ["free_seq", [1]],
           ["alloc_seq", 28], ["num_free"], ["alloc_seq", 1], ["free_seq", [0, 2]], ["alloc_seq", 5],
           ["free_seq", [3, 99, 4]], ["num_free"], ["is_free", 3], ["is_free", 4], ["free_seq", []], ["alloc_seq", 4],
           ["alloc_seq", 4], ["alloc_seq", 4], ["util"]]
    scripts.append({"name": "sequences", "num_blocks": 10, "block_size": 4, "ops": ops})

    ops = [["req_alloc", 7, 40], ["req_alloc", 9, 17], ["active"], ["req_count", 7], ["req_blocks", 9],
           ["req_alloc_block", 7], ["req_alloc_block", 11], ["req_blocks", 7], ["req_alloc", 12, 1000],
           ["active"], ["req_count", 12], ["req_free", 9], ["active"], ["num_free"], ["req_alloc", 9, 16],
           ["req_blocks", 9], ["req_free", 404], ["req_alloc", 5, 0], ["active"], ["req_count", 5],
           ["req_free", 7], ["req_free", 11], ["req_free", 9], ["num_free"], ["active"], ["alloc"],
           ["req_alloc_block", 1], ["req_blocks", 1]]
    scripts.append({"name": "per_request", "num_blocks": 12, "block_size": 16, "ops": ops})

    # a long pseudo-random mix on an odd-sized pool (crosses a 64-bit word boundary of the bitmap)
    nb = 131
    draws = splitmix64(555, np.arange(1200, dtype=np.uint64))
    ops = []
    for d in draws.tolist():
        r = d % 100
        arg = (d >> 8) % (nb + 6) - 3
        if r < 45:
            ops.append(["alloc"])
        elif r < 75:
            ops.append(["free", int(arg)])
        elif r < 85:
            ops.append(["alloc_seq", int((d >> 20) % 200)])
        elif r < 92:
            ops.append(["is_free", int(arg)])
        elif r < 96:
            ops.append(["num_free"])
        else:
            ops.append(["req_alloc", int((d >> 30) % 5), int((d >> 20) % 90)] if r < 98
                       else ["req_free", int((d >> 30) % 5)])
    scripts.append({"name": "random_mix_131", "num_blocks": nb, "block_size": 8, "ops": ops})
    return scripts


class BmCLib:
    """Adapter: a C library exposing <prefix>bm_* (the oracle or the reference shim)."""

    def __init__(self, lib, prefix, num_blocks, block_size):
        self.lib, self.prefix = lib, prefix
        self.bm = self._f("create")(num_blocks, block_size)
        self.cap = max(num_blocks, 1)
        self.buf = (C.c_int * self.cap)()

    def _f(self, name):
        return getattr(self.lib, self.prefix + "bm_" + name)

    def close(self):
        self._f("destroy")(self.bm)

    def do(self, op):
        kind = op[0]
        if kind == "alloc":
            return int(self._f("allocate_block")(self.bm))
        if kind == "free":
            return 0 if self._f("free_block")(self.bm, op[1]) == 0 else "throw"
        if kind == "alloc_seq":
            n = int(self._f("allocate_sequence")(self.bm, op[1], self.buf, self.cap))
            return list(self.buf[:n]) if n >= 0 else "throw"
        if kind == "free_seq":
            arr = (C.c_int * max(len(op[1]), 1))(*op[1])
            return 0 if self._f("free_sequence")(self.bm, arr, len(op[1])) == 0 else "throw"
        if kind == "req_alloc":
            n = int(self._f("allocate_for_request")(self.bm, op[1], op[2], self.buf, self.cap))
            return list(self.buf[:n])
        if kind == "req_alloc_block":
            return int(self._f("allocate_block_for_request")(self.bm, op[1]))
        if kind == "req_free":
            self._f("free_request")(self.bm, op[1])
            return None
        if kind == "req_blocks":
            n = int(self._f("get_request_blocks")(self.bm, op[1], self.buf, self.cap))
            return list(self.buf[:n])
        if kind == "req_count":
            return int(self._f("get_request_block_count")(self.bm, op[1]))
        if kind == "active":
            return int(self._f("num_active_requests")(self.bm))
        if kind == "num_free":
            return int(self._f("num_free")(self.bm))
        if kind == "is_free":
            return int(self._f("is_free")(self.bm, op[1]))
        if kind == "util":
            return float(self._f("utilization")(self.bm))
        raise ValueError(kind)


class BmProduct:
    """Adapter: nano_vllm_amd.BlockManager (the C ABI through the host mirror)."""

    def __init__(self, num_blocks, block_size):
        from nano_vllm_amd import BlockManager
        self.bm = BlockManager(num_blocks, block_size)

    def close(self):
        self.bm = None

    def do(self, op):
        kind, bm = op[0], self.bm
        try:
            if kind == "alloc":
                return bm.allocate_block()
            if kind == "free":
                bm.free_block(op[1])
                return 0
            if kind == "alloc_seq":
                return bm.allocate_sequence(op[1])
            if kind == "free_seq":
                bm.free_sequence(op[1])
                return 0
        except RuntimeError:
            return "throw"
        if kind == "req_alloc":
            return bm.allocate_for_request(op[1], op[2])
        if kind == "req_alloc_block":
            return bm.allocate_block_for_request(op[1])
        if kind == "req_free":
            bm.free_request(op[1])
            return None
        if kind == "req_blocks":
            return bm.get_request_blocks(op[1])
        if kind == "req_count":
            return bm.get_request_block_count(op[1])
        if kind == "active":
            return bm.get_num_active_requests()
        if kind == "num_free":
            return bm.get_num_free_blocks()
        if kind == "is_free":
            return int(bm.is_free(op[1]))
        if kind == "util":
            return bm.get_utilization()
        raise ValueError(kind)


def replay(adapter, ops):
    out = [adapter.do(op) for op in ops]
    adapter.close()
    return out


# --------------------------------------------------------------- F4 models
def _tokens(seed, n, vocab):
    return [int(x) for x in (splitmix64(seed, np.arange(n, dtype=np.uint64)) % np.uint64(vocab)).tolist()]


TINY_MODELS = [
    # a GQA miniature: 6 q-heads over 2 kv-heads (kv_mul 3), head_dim 16, 3 layers
    dict(name="mini_gqa", dim=96, hidden_dim=256, n_layers=3, n_heads=6, n_kv_heads=2, vocab_size=97,
         max_seq_len=96, block_size=16, num_blocks=64, weight_seed=11, weight_gain=0.35, shared=True,
         forced_tokens=_tokens(21, 50, 97), prompt_tokens=[1] + _tokens(22, 5, 97), gen_steps=50),
    # the default checkpoint's geometry (stories15M: 6 layers, 6/6 heads, head_dim 48) with a small vocabulary;
    # block_size/num_blocks are the reference's compile-time defaults (model.hpp:35-36)
    dict(name="g_tiny", dim=288, hidden_dim=768, n_layers=6, n_heads=6, n_kv_heads=6, vocab_size=512,
         max_seq_len=256, block_size=16, num_blocks=256, weight_seed=12, weight_gain=0.30, shared=False,
         forced_tokens=_tokens(23, 40, 512), prompt_tokens=[1] + _tokens(24, 7, 512), gen_steps=50),
]


def checkpoint_tensors(spec):
    """Weights of the synthetic checkpoint, in llama2.c file order (reference model.hpp:322-347)."""
    D, Hd, L = spec["dim"], spec["hidden_dim"], spec["n_layers"]
    hd = D // spec["n_heads"]
    QD, KD, V = spec["n_heads"] * hd, spec["n_kv_heads"] * hd, spec["vocab_size"]
    sizes = [("tok_emb", V * D), ("rms_att", L * D), ("wq", L * D * QD), ("wk", L * D * KD), ("wv", L * D * KD),
             ("wo", L * QD * D), ("rms_ffn", L * D), ("w_gate", L * D * Hd), ("w_down", L * Hd * D),
             ("w_up", L * D * Hd), ("rms_final", D)]
    if not spec["shared"]:
        sizes.append(("lm_head", V * D))
    out, at = [], 0
    for name, n in sizes:
        w = uniform(spec["weight_seed"], at, n)
        at += n
        if name.startswith("rms"):
            w = (np.float32(1.0) + np.float32(0.1) * w).astype(np.float32)
        elif name in ("tok_emb", "lm_head"):
            w = w.astype(np.float32)
        else:
            w = (w * np.float32(spec["weight_gain"])).astype(np.float32)
        out.append((name, w))
    return out


def write_checkpoint(path, spec):
    """llama2.c format: 7 int32 header (dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, max_seq_len)."""
    with open(path, "wb") as fh:
        fh.write(struct.pack("7i", spec["dim"], spec["hidden_dim"], spec["n_layers"], spec["n_heads"],
                             spec["n_kv_heads"], spec["vocab_size"], spec["max_seq_len"]))
        for _, w in checkpoint_tensors(spec):
            fh.write(w.tobytes())


# ---- F8: a synthetic tokenizer.bin (reference format, tokenizer.hpp:31-61) and texts to encode ----------------
def tokenizer_spec(vocab_size=512):
    """(pieces, scores, max_token_length): ids 0-2 <unk> <s> </s>, 3-258 the byte pieces <0x00>..<0xFF>, then single
    printable characters, then multi-character pieces (merges) with descending scores -- the shape of llama2.c's
    tokenizer, small.  Deterministic; no duplicates."""
    pieces = [b"<unk>", b"<s>", b"</s>"] + [b"<0x%02X>" % i for i in range(256)]
    singles = [bytes([c]) for c in b" etaoinshrdlucmfwypvbgkqjxzETAOINSHRDLUCMFWYPVBGKQJXZ0123456789.,!?'\"-:;()\n"]
    pieces += singles
    scores = [0.0] * len(pieces)
    merges = [b" t", b"he", b" a", b"in", b" the", b"er", b" s", b"on", b"re", b" w", b"at", b"nd", b"en", b" o", b"ou",
              b"ing", b" c", b"es", b"or", b" b", b"it", b"an", b" and", b"ll", b" h", b"is", b" f", b"ed", b" to", b"ar",
              b" m", b" p", b"le", b" d", b"as", b"om", b" in", b"st", b" l", b"ow", b" you", b"ve", b"ot", b" how", b"ay",
              b"hel", b"hell", b"hello", b" hello", b" are", b"wor", b"world", b" world", b"?!", b"...", b" on", b" once",
              b" upon", b"time", b" time", b" was", b" there", b"ce", b"up", b"th", b" th", b"ere", b" is", b" it"]
    seen = set(pieces)
    for k, mrg in enumerate(merges):
        if len(pieces) >= vocab_size:
            break
        if mrg not in seen:
            pieces.append(mrg)
            seen.add(mrg)
            scores.append(-1.0 - 0.5 * k)
    k = 0
    while len(pieces) < vocab_size:  # filler pieces that never match ASCII text
        pieces.append(b"\xf0\x9f" + bytes([0x80 + k // 64, 0x80 + k % 64]))
        scores.append(-1000.0 - k)
        k += 1
    return pieces[:vocab_size], scores[:vocab_size], max(len(p) for p in pieces[:vocab_size])


def write_tokenizer(path, vocab_size=512):
    pieces, scores, max_len = tokenizer_spec(vocab_size)
    with open(path, "wb") as fh:
        fh.write(struct.pack("i", max_len))
        for piece, score in zip(pieces, scores):
            fh.write(struct.pack("fi", score, len(piece)))
            fh.write(piece)


TOKENIZER_TEXTS = [b"", b"hello world", b"Hello, how are you?", b"the theory of the thing", b"once upon a time there was",
                   b"  two  spaces", b"caf\xc3\xa9 \xe2\x82\xac5", b"a", b" ", b"?!...?!", b"tttttt", b"in the inn, an ant and a hen",
                   b"line one\nline two", b"x" * 40, b"hellohellohello", b"0123456789"]
