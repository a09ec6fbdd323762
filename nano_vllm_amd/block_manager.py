"""BlockManager: host-side mirror of the reference class, over the C ABI.

Same method names, argument meaning and error behaviour as
``BlockManager`` in the reference's ``include/scheduler/block_manager.hpp:14-283``
(RuntimeError where it throws ``std::runtime_error``, ``-1`` / ``[]`` where it
returns them), so tests written against the reference read the same here.
The allocator itself is native (``nvk_bm_*`` in ``csrc/nvk_block_manager.cpp``).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

from . import _capi


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int):
        handle = C.c_void_p()
        _capi.check(_capi.lib.nvk_bm_create(int(num_blocks), int(block_size), C.byref(handle)))
        self._h = handle
        self._cap = max(int(num_blocks), 1)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _capi.lib.nvk_bm_destroy(h)
            self._h = None

    # ---- plain (unlocked) API, ref :28-124 --------------------------------
    def allocate_block(self) -> int:
        return int(_capi.lib.nvk_bm_allocate_block(self._h))

    def free_block(self, block_id: int) -> None:
        rc = _capi.lib.nvk_bm_free_block(self._h, int(block_id))
        if rc == _capi.ERR_BAD_BLOCK:
            raise RuntimeError("Invalid block_id")  # ref :52
        _capi.check(rc)

    def allocate_sequence(self, num_tokens: int) -> List[int]:
        ids = (C.c_int * self._cap)()
        n = _capi.lib.nvk_bm_allocate_sequence(self._h, int(num_tokens), ids, self._cap)
        if n == _capi.ERR_OOM:
            raise RuntimeError("Out of memory")  # ref :73
        _capi.check(n)
        return list(ids[:n])

    def free_sequence(self, block_ids: Sequence[int]) -> None:
        n = len(block_ids)
        arr = (C.c_int * max(n, 1))(*block_ids)
        rc = _capi.lib.nvk_bm_free_sequence(self._h, arr, n)
        if rc == _capi.ERR_BAD_BLOCK:
            raise RuntimeError("Invalid block_id")
        _capi.check(rc)

    def get_num_free_blocks(self) -> int:
        return int(_capi.lib.nvk_bm_num_free_blocks(self._h))

    def get_num_blocks(self) -> int:
        return int(_capi.lib.nvk_bm_num_blocks(self._h))

    def get_block_size(self) -> int:
        return int(_capi.lib.nvk_bm_block_size(self._h))

    def is_free(self, block_id: int) -> bool:
        return bool(_capi.lib.nvk_bm_is_free(self._h, int(block_id)))

    def get_utilization(self) -> float:
        return float(_capi.lib.nvk_bm_utilization(self._h))

    # ---- per-request (locked, non-throwing) API, ref :132-216 -------------
    def allocate_for_request(self, request_id: int, num_tokens: int) -> List[int]:
        ids = (C.c_int * self._cap)()
        n = _capi.lib.nvk_bm_allocate_for_request(self._h, int(request_id), int(num_tokens), ids, self._cap)
        return list(ids[: max(n, 0)])

    def allocate_block_for_request(self, request_id: int) -> int:
        return int(_capi.lib.nvk_bm_allocate_block_for_request(self._h, int(request_id)))

    def free_request(self, request_id: int) -> None:
        _capi.lib.nvk_bm_free_request(self._h, int(request_id))

    def release_block_for_request(self, request_id: int, block_id: int) -> None:
        """One block of a request back to the free set, out of the request's list in the same locked step."""
        _capi.check(_capi.lib.nvk_bm_release_block_for_request(self._h, int(request_id), int(block_id)))

    def get_request_blocks(self, request_id: int) -> List[int]:
        n = _capi.lib.nvk_bm_get_request_block_count(self._h, int(request_id))
        ids = (C.c_int * max(n, 1))()
        n = _capi.lib.nvk_bm_get_request_blocks(self._h, int(request_id), ids, max(n, 1))
        return list(ids[:n])

    def get_request_block_count(self, request_id: int) -> int:
        return int(_capi.lib.nvk_bm_get_request_block_count(self._h, int(request_id)))

    def get_num_active_requests(self) -> int:
        return int(_capi.lib.nvk_bm_num_active_requests(self._h))

    def allocate_block_safe(self) -> int:
        return int(_capi.lib.nvk_bm_allocate_block_safe(self._h))

    def free_block_safe(self, block_id: int) -> None:
        _capi.lib.nvk_bm_free_block_safe(self._h, int(block_id))
