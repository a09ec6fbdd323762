"""The reference's tokenizer.bin + BPE behind the C ABI: ``class Tokenizer``,
``include/core/tokenizer.hpp`` -- same file format, same token ids.  Host-only: works without a GPU."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Union

from . import _capi


class Tokenizer:
    def __init__(self, path: str, vocab_size: int):
        h = C.c_void_p()
        _capi.check(_capi.lib.nvk_tokenizer_load(os.fsencode(path), int(vocab_size), C.byref(h)))
        self._h = h
        self.vocab_size = int(_capi.lib.nvk_tokenizer_vocab_size(h))
        self.max_token_length = int(_capi.lib.nvk_tokenizer_max_token_length(h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            _capi.lib.nvk_tokenizer_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def encode(self, text: Union[str, bytes], bos: bool = True, eos: bool = False) -> List[int]:
        """Tokenizer::encode(text, bos, eos) (tokenizer.hpp:77-131)."""
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        cap = len(raw) + 8
        ids = (C.c_int * cap)()
        n = _capi.check(_capi.lib.nvk_tokenizer_encode(self._h, raw, len(raw), int(bos), int(eos), ids, cap))
        return list(ids[:n])

    def decode_bytes(self, token: int) -> bytes:
        """Tokenizer::decode(token) (tokenizer.hpp:63-75) as raw bytes."""
        buf = C.create_string_buffer(max(self.max_token_length, 8) + 16)
        n = _capi.check(_capi.lib.nvk_tokenizer_decode(self._h, int(token), buf, len(buf)))
        return buf.raw[:min(n, len(buf) - 1)]

    def decode(self, token: int) -> str:
        return self.decode_bytes(token).decode("utf-8", errors="replace")
