"""ORACLE — test infrastructure: ctypes binding of oracle/n2v_oracle.c + sgns_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libn2v_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("n2v_oracle.c", "sgns_oracle.c", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs if os.path.exists(s))):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_walk_tables.restype = C.c_int64
        _lib.orc_walk_on_the_fly.restype = C.c_int64
        _lib.orc_sgns_train.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def mt19937_fill(seed, n, skip=0):
    out = np.empty(n, dtype=np.float64)
    lib().orc_mt19937_fill(C.c_uint32(seed), C.c_int64(skip), C.c_int64(n), _p(out))
    return out


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(o))
    return tuple(int(x) for x in o)


def alias_setup(probs):
    probs = np.ascontiguousarray(probs, dtype=np.float64)
    K = len(probs)
    J = np.zeros(K, dtype=np.int32)
    q = np.zeros(K, dtype=np.float64)
    rc = lib().orc_alias_setup(_p(probs), C.c_int64(K), _p(J), _p(q))
    assert rc == 0
    return J, q


class CsrOracle:
    """Tables + walks on a dense CSR (see n2v_oracle.py:to_csr)."""

    def __init__(self, row_ptr, col, w, p, q):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        self.N = len(self.row_ptr) - 1
        self.nnz = int(self.row_ptr[-1])
        self.p, self.q = float(p), float(q)
        self.nodeJ = self.nodeq = self.edge_off = self.edgeJ = self.edgeq = None

    def build_node_tables(self):
        self.nodeJ = np.zeros(self.nnz, dtype=np.int32)
        self.nodeq = np.zeros(self.nnz, dtype=np.float64)
        rc = lib().orc_build_node_tables(C.c_int64(self.N), _p(self.row_ptr), _p(self.col), _p(self.w),
                                         _p(self.nodeJ), _p(self.nodeq))
        if rc == -1:
            raise ZeroDivisionError("float division by zero")
        assert rc == 0

    def build_edge_tables(self):
        self.edge_off = np.zeros(self.nnz + 1, dtype=np.int64)
        lib().orc_edge_offsets(C.c_int64(self.N), _p(self.row_ptr), _p(self.col), _p(self.edge_off))
        T = int(self.edge_off[-1])
        self.edgeJ = np.zeros(T, dtype=np.int32)
        self.edgeq = np.zeros(T, dtype=np.float64)
        rc = lib().orc_build_edge_tables(C.c_int64(self.N), _p(self.row_ptr), _p(self.col), _p(self.w),
                                         C.c_double(self.p), C.c_double(self.q), _p(self.edge_off),
                                         _p(self.edgeJ), _p(self.edgeq))
        if rc == -1:
            raise ZeroDivisionError("float division by zero")
        assert rc == 0

    def preprocess(self, first_order_shortcut=False):
        self.build_node_tables()
        if not first_order_shortcut:
            self.build_edge_tables()

    def walk(self, starts, num_walks, L, mode="mt", seed=0, uniforms=None, walk_index_base=0,
             on_the_fly=False):
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        W = len(starts) * num_walks
        walks = np.empty((W, L), dtype=np.int32)
        lens = np.zeros(W, dtype=np.int32)
        m = {"mt": 0, "buffer": 1, "philox": 2}[mode]
        if uniforms is not None:
            uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        if on_the_fly:
            n = lib().orc_walk_on_the_fly(
                C.c_int64(self.N), _p(self.row_ptr), _p(self.col), _p(self.w),
                C.c_double(self.p), C.c_double(self.q), _p(starts), C.c_int64(len(starts)),
                C.c_int64(num_walks), C.c_int64(L), C.c_int(m), C.c_uint64(seed), _p(uniforms),
                C.c_uint64(walk_index_base), _p(walks), _p(lens))
        else:
            n = lib().orc_walk_tables(
                C.c_int64(self.N), _p(self.row_ptr), _p(self.col), _p(self.nodeJ), _p(self.nodeq),
                _p(self.edge_off), _p(self.edgeJ), _p(self.edgeq), _p(starts), C.c_int64(len(starts)),
                C.c_int64(num_walks), C.c_int64(L), C.c_int(m), C.c_uint64(seed), _p(uniforms),
                C.c_uint64(walk_index_base), _p(walks), _p(lens))
        if n == -1:
            raise ZeroDivisionError("float division by zero")
        assert n >= 0
        return walks, lens, int(n)


# ---------------------------------------------------------------------------- SGNS (sgns_oracle.c)
def randint_fill(seed, high, n):
    out = np.empty(n, dtype=np.int64)
    lib().orc_randint_fill(C.c_uint32(seed), C.c_uint32(high), C.c_int64(n), _p(out))
    return out


def sgns_init(n_words, dim, stride, seed):
    syn0 = np.empty((n_words, stride), dtype=np.float32)
    syn1 = np.empty((n_words, stride), dtype=np.float32)
    lib().orc_sgns_init(_p(syn0), _p(syn1), C.c_int64(n_words), C.c_int32(dim), C.c_int32(stride), C.c_uint64(seed))
    return syn0, syn1


def sgns_train(walks, lens, syn0, syn1neg, dim, window, negative, sample_int, cum_table, alpha=0.025,
               min_alpha=1e-4, epochs=1, seed=1, n_threads=1):
    """In-place training of syn0/syn1neg (float32 [n_words, stride]); returns pairs trained."""
    walks = np.ascontiguousarray(walks, dtype=np.int32)
    lens = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    assert syn0.dtype == np.float32 and syn0.flags.c_contiguous and syn1neg.flags.c_contiguous
    si = None if sample_int is None else np.ascontiguousarray(sample_int, dtype=np.uint32)
    ct = np.ascontiguousarray(cum_table, dtype=np.uint32)
    n = lib().orc_sgns_train(
        _p(walks), _p(lens), C.c_int64(walks.shape[0]), C.c_int32(walks.shape[1]), _p(syn0), _p(syn1neg),
        C.c_int64(syn0.shape[0]), C.c_int32(dim), C.c_int32(syn0.shape[1]), C.c_int32(window), C.c_int32(negative),
        _p(si), _p(ct), C.c_float(alpha), C.c_float(min_alpha), C.c_int32(epochs), C.c_uint32(seed),
        C.c_int32(n_threads))
    return int(n)
