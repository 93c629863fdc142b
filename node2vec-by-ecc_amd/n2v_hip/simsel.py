"""Host side of the all-pairs similarity + selection kernels (include/n2v_sim.h, csrc/n2v_sim.hip):
SURVEY.md 8(f-1) top-k link prediction and 8(f-3) similarity-driven edge selection.

torch holds the buffers and runs the small glue (prefix sums of per-row counts, the final sort of a few
thousand candidates); every score and every selection decision is made by the HIP kernels.  No CPU path."""
import numpy as np
import torch

from . import _lib

METHODS = {"cos": 0, "pearson": 1, "jsd": 2}   # N2V_SIM_* of include/n2v_sim.h


def _dpad(dim):
    return -(-int(dim) // 32) * 32


def prepare(vectors, method="cos", rows=None, dim=None):
    """vectors: float32 device tensor [n, stride] (padding columns allowed when `dim` is given);
    rows: optional int64 index tensor.  Returns fp32 [n_rows, dpad] in the form whose dot product
    (or rel_entr sum, "jsd") is the similarity."""
    if not vectors.is_cuda:
        raise RuntimeError("n2v_hip: similarity kernels need device tensors; there is no CPU fallback")
    lib = _lib.load()
    v = vectors if vectors.dtype == torch.float32 and vectors.is_contiguous() else vectors.float().contiguous()
    dim = int(v.shape[1]) if dim is None else int(dim)
    n_rows = int(v.shape[0]) if rows is None else int(rows.numel())
    r = None if rows is None else rows.to(device=v.device, dtype=torch.int64).contiguous()
    out = torch.empty((n_rows, _dpad(dim)), dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(lib.n2v_sim_prepare(_lib.ptr(v), int(v.shape[1]), dim, _lib.ptr(r), n_rows, METHODS[method],
                                       _lib.ptr(out), int(out.shape[1]), _lib.stream_ptr(v.device)))
    return out


def score_block(A, row_begin, n_rows, B, method="cos", zero_diag_off=-1, out=None):
    """Scores of rows [row_begin, row_begin+n_rows) of A against all rows of B -> fp32 [n_rows, n_cols]."""
    lib = _lib.load()
    n_cols = int(B.shape[0])
    if out is None:
        out = torch.empty((n_rows, n_cols), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        _lib.check(lib.n2v_sim_block(_lib.ptr(A), int(row_begin), int(n_rows), _lib.ptr(B), n_cols, int(A.shape[1]),
                                     METHODS[method], int(zero_diag_off), _lib.ptr(out), int(out.stride(0)),
                                     _lib.stream_ptr(A.device)))
    return out


def global_topk(A, B, k, method="cos", upper_triangle=False, exclude_keys=None, capacity=1 << 20, first_rows=64):
    """The k best-scoring (row, col) pairs of A x B (col > row only if upper_triangle), pairs whose key
    row * n_cols + col is in `exclude_keys` (sorted int64 device tensor) left out.  Streaming: row blocks of
    doubling size are scanned against a running threshold tau = the k-th best score seen so far; only scores
    above it reach the candidate buffer.  Returns (scores, rows, cols) sorted by descending score (ties by
    ascending (row, col))."""
    lib = _lib.load()
    dev = A.device
    n_rows, n_cols = int(A.shape[0]), int(B.shape[0])
    k = int(k)
    tau = torch.full((1,), -float("inf"), dtype=torch.float32, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    cs = torch.empty(capacity, dtype=torch.float32, device=dev)
    cr = torch.empty(capacity, dtype=torch.int32, device=dev)
    cc = torch.empty(capacity, dtype=torch.int32, device=dev)
    excl = None if exclude_keys is None or exclude_keys.numel() == 0 else exclude_keys.contiguous()
    best = (torch.empty(0, dtype=torch.float32, device=dev), torch.empty(0, dtype=torch.int32, device=dev),
            torch.empty(0, dtype=torch.int32, device=dev))

    def scan(b, e):
        counter.zero_()
        with torch.cuda.device(dev):
            _lib.check(lib.n2v_sim_topk_scan(
                _lib.ptr(A), b, e, _lib.ptr(B), n_cols, int(A.shape[1]), METHODS[method], 1 if upper_triangle else 0,
                _lib.ptr(tau), _lib.ptr(excl), 0 if excl is None else int(excl.numel()), _lib.ptr(cs), _lib.ptr(cr),
                _lib.ptr(cc), capacity, _lib.ptr(counter), _lib.stream_ptr(dev)))
        return int(counter.item())

    def fold(n):
        nonlocal best
        s = torch.cat([best[0], cs[:n]])
        r = torch.cat([best[1], cr[:n]])
        c = torch.cat([best[2], cc[:n]])
        if s.numel() > k:
            top = torch.topk(s, k, sorted=False).indices
            s, r, c = s[top], r[top], c[top]
        best = (s, r, c)
        if s.numel() >= k:
            tau.fill_(float(s.min().item()))   # strict '>' in the kernel: later ties of the k-th score are dropped

    b, step = 0, max(1, int(first_rows))
    while b < n_rows:
        e = min(n_rows, b + step)
        n = scan(b, e)
        while n > capacity:
            # the block holds more candidates than the buffer: raise tau to the k-th best of what was kept (a valid
            # lower bound of the final threshold) and rescan the same rows
            if capacity < k or e - b == 0:
                raise RuntimeError("global_topk: candidate buffer smaller than k")
            keep = torch.topk(torch.cat([best[0], cs[:capacity]]), k, sorted=False).values.min()
            if not (float(keep.item()) > float(tau.item())):
                if e - b == 1:
                    raise RuntimeError("global_topk: more than %d ties above the threshold" % capacity)
                e = b + max(1, (e - b) // 2)
            else:
                tau.fill_(float(keep.item()))
            n = scan(b, e)
        fold(n)
        b, step = e, min(step * 2, 1 << 21)      # one launch covers at most 65 535 row tiles
    s, r, c = best
    # descending score; ties in (row, col) order — a deterministic stand-in for the reference's set/argsort order
    key = r.to(torch.int64) * n_cols + c.to(torch.int64)
    o = torch.argsort(key)
    s, r, c = s[o], r[o], c[o]
    o = torch.argsort(s, descending=True, stable=True)
    return s[o], r[o], c[o]


def rows_above(scores, n_cols, thre):
    """Per row of the score block, the columns with score > thre in column order.
    Returns (row_idx int64, col int32, val fp32) concatenated row by row."""
    lib = _lib.load()
    dev = scores.device
    n_rows = int(scores.shape[0])
    counts = torch.empty(n_rows, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.n2v_sim_rows_count(_lib.ptr(scores), n_rows, int(n_cols), int(scores.stride(0)), float(thre),
                                          _lib.ptr(counts), _lib.stream_ptr(dev)))
        off = torch.cumsum(counts, 0) - counts
        total = int(counts.sum().item())
        cols = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        vals = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
        _lib.check(lib.n2v_sim_rows_fill(_lib.ptr(scores), n_rows, int(n_cols), int(scores.stride(0)), float(thre),
                                         _lib.ptr(off.contiguous()), _lib.ptr(cols), _lib.ptr(vals),
                                         _lib.stream_ptr(dev)))
    rows = torch.repeat_interleave(torch.arange(n_rows, device=dev), counts)
    return rows, cols[:total], vals[:total]


def rows_topk(scores, n_cols, k):
    """Per row the k largest scores as the reference's sorted(..., key=-score)[:k] gives them: descending,
    ties in column order.  Returns (cols int32 [n_rows, k], vals fp32 [n_rows, k])."""
    lib = _lib.load()
    dev = scores.device
    n_rows = int(scores.shape[0])
    cols = torch.empty((n_rows, k), dtype=torch.int32, device=dev)
    vals = torch.empty((n_rows, k), dtype=torch.float32, device=dev)
    if k == 0 or n_rows == 0:
        return cols, vals
    with torch.cuda.device(dev):
        _lib.check(lib.n2v_sim_rows_topk(_lib.ptr(scores), n_rows, int(n_cols), int(scores.stride(0)), int(k),
                                         _lib.ptr(cols), _lib.ptr(vals), _lib.stream_ptr(dev)))
    # the kernel selected the SET (column order); order it like the reference's stable descending sort
    o = torch.argsort(vals, dim=1, descending=True, stable=True)
    return torch.gather(cols, 1, o), torch.gather(vals, 1, o)
