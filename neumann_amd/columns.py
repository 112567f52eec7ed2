"""Columnar metadata + WHERE-predicate programs on the GPU (include/neumann_gpu.h, `nmn_columns_*`).

`GpuColumns` is the thin host wrapper of the C ABI the engine's pre-filter path drives
(neumann_amd/csrc/nmn_engine.cpp: columns_build / compile_filter / pre_filter_search): typed
(kind, payload) cells per row and field in HBM, a postfix predicate program evaluated by one kernel,
the selection bitmap left in device memory for the masked scan.  It replaces the per-key
`store.get` + `evaluate_filter` of the reference (vector_engine/src/lib.rs:3526-3530, 3582-3630).
"""
import ctypes as C
import struct

import numpy as np

from . import _capi
from ._capi import (CELL_ABSENT, CELL_BOOL, CELL_FLOAT, CELL_INT, CELL_NULL, CELL_STRING,  # noqa: F401
                    CMP_EQ, CMP_GE, CMP_GT, CMP_LE, CMP_LT, CMP_NE, PRED_AND, PRED_CMP, PRED_EXISTS,  # noqa: F401
                    PRED_FALSE, PRED_IN, PRED_OR, PRED_STRSET, PRED_TRUE, PredOp)  # noqa: F401


def f64_bits(x):
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def i64_bits(x):
    return int(x) & 0xFFFFFFFFFFFFFFFF


class GpuColumns:
    def __init__(self, capacity_rows, device=-1):
        self._lib = _capi.load()
        h = C.c_void_p()
        _capi.check(self._lib.nmn_columns_create(int(device), int(capacity_rows), C.byref(h)))
        self._h = h
        self.capacity_rows = int(capacity_rows)

    def close(self):
        if self._h:
            self._lib.nmn_columns_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_column(self):
        cid = C.c_uint32()
        _capi.check(self._lib.nmn_columns_add(self._h, C.byref(cid)))
        return int(cid.value)

    @property
    def n_columns(self):
        return int(self._lib.nmn_columns_count(self._h))

    def write(self, column, row0, kinds, payloads):
        kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
        payloads = np.ascontiguousarray(payloads, dtype=np.uint64)
        assert kinds.size == payloads.size
        _capi.check(self._lib.nmn_columns_write(self._h, int(column), int(row0), kinds.size,
                                                C.c_void_p(kinds.ctypes.data), C.c_void_p(payloads.ctypes.data)))

    def clear_row(self, row):
        _capi.check(self._lib.nmn_columns_clear_row(self._h, int(row)))

    def write_valid(self, word0, words):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        _capi.check(self._lib.nmn_columns_write_valid(self._h, int(word0), words.size, C.c_void_p(words.ctypes.data)))

    def eval(self, ops, consts, n_rows):
        """ops: list of (op, cmp, vkind, column, a, b) tuples (or PredOp); consts: iterable of u64.
        Returns the number of selected rows; the bitmap stays on the device (`mask_device`, `read_mask`)."""
        arr = (PredOp * max(len(ops), 1))()
        for i, o in enumerate(ops):
            arr[i] = o if isinstance(o, PredOp) else PredOp(*o)
        cs = np.ascontiguousarray(np.asarray(list(consts), dtype=np.uint64))
        cnt = C.c_uint64()
        _capi.check(self._lib.nmn_columns_eval(self._h, arr, len(ops), C.c_void_p(cs.ctypes.data) if cs.size else None,
                                               cs.size, int(n_rows), C.byref(cnt)))
        return int(cnt.value)

    @property
    def mask_device(self):
        return self._lib.nmn_columns_mask_device(self._h)

    @property
    def valid_device(self):
        return self._lib.nmn_columns_valid_device(self._h)

    def read_mask(self, n_rows):
        words = (int(n_rows) + 63) // 64
        out = np.empty(words, dtype=np.uint64)
        _capi.check(self._lib.nmn_columns_read_mask(self._h, C.c_void_p(out.ctypes.data), words))
        return out
