"""Executor for a stand-alone `GraphQuery.edge_fold` system over arbitrary components (dsl.GraphFold): the generated
kernels of codegen.generate_graph_fold_source, driven through the HIP runtime with ctypes.  GPU only — like the rest of
the product there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Sequence

import numpy as np

from ._lib import BackendError
from . import codegen, dsl

_H2D, _D2H = 1, 2


def _hip():
    try:
        lib = C.CDLL("libamdhip64.so")
    except OSError as e:                                    # pragma: no cover - no ROCm runtime at all
        raise BackendError(f"HIP runtime not found: {e}") from None
    n = C.c_int(0)
    if lib.hipGetDeviceCount(C.byref(n)) != 0 or n.value < 1:
        raise BackendError("no HIP device: edge_fold systems run on the GPU only")
    lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.hipFree.argtypes = [C.c_void_p]
    return lib


class _Params(C.Structure):   # GraphFoldParams of the generated translation unit
    _fields_ = [("left", C.c_void_p * 8), ("right", C.c_void_p * 8), ("scratch", C.c_void_p), ("out", C.c_void_p),
                ("row_start", C.c_void_p), ("dst", C.c_void_p), ("src_rows", C.c_void_p), ("n_src", C.c_uint32)]


class GraphFoldExec:
    """columns: {component: (rows [n,w], entity ids [n])}; edges: (from_ids, to_ids) in spawn order."""

    def __init__(self, fold: dsl.GraphFold, columns: Mapping[str, tuple], edges: Sequence[np.ndarray], device: int = 0):
        self._hip = _hip()
        self._check(self._hip.hipSetDevice(device), "hipSetDevice")
        names = list(dict.fromkeys(fold.left + fold.right + (fold.out,)))
        if len(fold.left) > 8 or len(fold.right) > 8:
            raise ValueError("edge_fold queries are limited to 8 components each")
        self.names = names
        self._own = {n: (np.array(columns[n][0], dtype=np.float64, order="C"), np.asarray(columns[n][1], dtype=np.uint64)) for n in names}
        widths = {n: self._own[n][0].shape[1] for n in names}
        traced = fold.trace(widths)
        so = codegen.build_graph_fold(traced)
        self._dl = C.CDLL(str(so))
        if self._dl.graph_fold_abi() != C.sizeof(_Params):
            raise BackendError("generated edge_fold object does not match this build (GraphFoldParams layout)")
        # common row set: every entity carrying one of the components, ascending id (query.rs:136-208)
        self.row_ids = np.unique(np.concatenate([ids for _, ids in self._own.values()]))
        row_of = {int(e): k for k, e in enumerate(self.row_ids)}
        self._at = {n: np.array([row_of[int(e)] for e in ids], dtype=np.int64) for n, (_, ids) in self._own.items()}
        has = {n: np.isin(self.row_ids, ids) for n, (_, ids) in self._own.items()}
        frm, to = (np.asarray(e, dtype=np.uint64) for e in edges)
        keep = [k for k in range(len(frm)) if int(frm[k]) in row_of and int(to[k]) in row_of
                and all(has[n][row_of[int(frm[k])]] for n in fold.left + (fold.out,))
                and all(has[n][row_of[int(to[k])]] for n in fold.right)]
        src = np.array([row_of[int(frm[k])] for k in keep], dtype=np.int64)
        dst = np.array([row_of[int(to[k])] for k in keep], dtype=np.int64)
        order = np.argsort(src, kind="stable")               # CSR by source, spawn order inside a source
        self.src_rows, counts = np.unique(src, return_counts=True)
        self._row_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
        self._dst = dst[order].astype(np.uint32)
        self._w = widths
        self._dense = {}
        for n in names:
            d = np.zeros((len(self.row_ids), widths[n]))
            d[self._at[n]] = self._own[n][0]
            self._dense[n] = d
        self._dev: Dict[str, C.c_void_p] = {}
        self._bufs = []
        p = _Params()
        for n in names:
            self._dev[n] = self._upload(self._dense[n])
        for i, n in enumerate(fold.left):
            p.left[i] = self._dev[n]
        for i, n in enumerate(fold.right):
            p.right[i] = self._dev[n]
        p.out = self._dev[fold.out]
        p.scratch = self._alloc(max(1, len(self.src_rows)) * widths[fold.out] * 8)
        p.row_start = self._upload(self._row_start)
        p.dst = self._upload(self._dst if len(self._dst) else np.zeros(1, np.uint32))
        p.src_rows = self._upload(self.src_rows.astype(np.uint32) if len(self.src_rows) else np.zeros(1, np.uint32))
        p.n_src = len(self.src_rows)
        self._p, self._out, self.tick = p, fold.out, 0

    def _check(self, rc, what):
        if rc != 0:
            raise BackendError(f"{what} failed with HIP error {rc}")

    def _alloc(self, nbytes):
        ptr = C.c_void_p()
        self._check(self._hip.hipMalloc(C.byref(ptr), max(int(nbytes), 8)), "hipMalloc")
        self._bufs.append(ptr)
        return ptr

    def _upload(self, arr):
        arr = np.ascontiguousarray(arr)
        ptr = self._alloc(arr.nbytes)
        self._check(self._hip.hipMemcpy(ptr, arr.ctypes.data, arr.nbytes, _H2D), "hipMemcpy H2D")
        return ptr

    def run(self, ticks: int = 1):
        self._check(self._dl.graph_fold_launch(C.byref(self._p), C.c_uint(int(ticks)), None), "graph_fold_launch")
        self._check(self._hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        d = self._dense[self._out]
        self._check(self._hip.hipMemcpy(d.ctypes.data, self._dev[self._out], d.nbytes, _D2H), "hipMemcpy D2H")
        self.tick += int(ticks)

    def column_array(self, name: str) -> np.ndarray:
        """The component's own rows in its own entity order (what `exec.history` would show last)."""
        return self._dense[name][self._at[name]]

    def column_ids(self, name: str) -> np.ndarray:
        return self._own[name][1]

    def close(self):
        for ptr in self._bufs:
            self._hip.hipFree(ptr)
        self._bufs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
