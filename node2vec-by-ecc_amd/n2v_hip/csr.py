"""Host-side graph container: networkx graph or edge arrays -> dense sorted CSR.

Dense id = rank of the node label in ascending order, rows sorted ascending, so the
k-th alias slot of a node refers to ``col[row_ptr[v] + k]`` exactly as it refers to
``sorted(G.neighbors(v))[k]`` in the reference (src/node2vec.py:67,142,185).  The start
order of ``simulate_walks`` — ``list(G.nodes())``, insertion order (src/node2vec.py:88) —
is kept separately as ``start_order``.  Labels are int64 (item ids like 9999999<id>
exceed int32, src/utils.py:392); dense ids are int32.
"""
import numpy as np


class CsrGraph:
    """labels int64[N] ascending; row_ptr int64[N+1]; col int32[nnz]; w float64[nnz] or
    None (all weights 1); start_order int32[N]; directed flag."""

    def __init__(self, labels, row_ptr, col, w, start_order, directed):
        self.labels = np.ascontiguousarray(labels, dtype=np.int64)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        self.start_order = np.ascontiguousarray(start_order, dtype=np.int32)
        self.directed = bool(directed)
        self.n_nodes = len(self.labels)
        self.nnz = int(self.row_ptr[-1]) if len(self.row_ptr) else 0

    @property
    def degrees(self):
        return np.diff(self.row_ptr)

    def dense_of(self, node_labels):
        """Label(s) -> dense id(s); KeyError for a label that is not a node."""
        arr = np.asarray(node_labels, dtype=np.int64).reshape(-1)
        idx = np.searchsorted(self.labels, arr)
        idx = np.minimum(idx, max(self.n_nodes - 1, 0))
        bad = (self.n_nodes == 0) | (self.labels[idx] != arr) if len(arr) else np.zeros(0, bool)
        if np.any(bad):
            raise KeyError(int(arr[np.argmax(bad)]))
        return idx.astype(np.int32)

    def src_of(self):
        """Row of every CSR entry (int32[nnz])."""
        return np.repeat(np.arange(self.n_nodes, dtype=np.int32), np.diff(self.row_ptr))


def _assemble(labels_in_order, src, dst, w, directed):
    """src/dst are LABELS of the final adjacency entries (already symmetrised, unique)."""
    labels_in_order = np.asarray(labels_in_order, dtype=np.int64)
    labels = np.sort(labels_in_order)
    if len(labels) > 1 and np.any(labels[1:] == labels[:-1]):
        raise ValueError("duplicate node labels")
    if len(labels) >= 2**31:
        raise ValueError("more than 2^31-1 nodes")
    s = np.searchsorted(labels, src)
    d = np.searchsorted(labels, dst)
    perm = np.lexsort((d, s))
    s, d = s[perm], d[perm]
    ww = None if w is None else np.asarray(w, dtype=np.float64)[perm]
    row_ptr = np.zeros(len(labels) + 1, dtype=np.int64)
    np.cumsum(np.bincount(s, minlength=len(labels)), out=row_ptr[1:])
    start_order = np.searchsorted(labels, labels_in_order).astype(np.int32)
    return CsrGraph(labels, row_ptr, d.astype(np.int32), ww, start_order, directed)


def from_networkx(G, is_directed=None):
    """Extract from a networkx Graph/DiGraph whose nodes are ints and whose edges carry
    'weight' (what src/main.py:66-80 builds).  A missing 'weight' raises KeyError, as
    G[u][v]['weight'] does in the reference."""
    directed = G.is_directed()
    nodes = list(G.nodes())
    src, dst, w = [], [], []
    all_one = True
    for u, nbrs in G.adjacency():
        for v, data in nbrs.items():
            wt = data["weight"]
            src.append(u)
            dst.append(v)
            w.append(wt)
            if all_one and not (wt == 1):
                all_one = False
    src = np.array(src, dtype=np.int64)
    dst = np.array(dst, dtype=np.int64)
    return _assemble(np.array(nodes, dtype=np.int64), src, dst,
                     None if all_one else np.array(w, dtype=np.float64), directed)


def from_edges(src, dst, weights=None, directed=False):
    """Edge arrays in file order -> the graph ``read_graph`` of src/main.py:66-80 would
    build with networkx (without instantiating networkx objects):
      * node order = first appearance, source before target on each line;
      * a repeated (u, v) line keeps its LAST weight;
      * undirected: DiGraph.to_undirected() — when both (u,v) and (v,u) lines exist the
        weight of the direction whose SOURCE comes later in node order wins.
    """
    src = np.asarray(src, dtype=np.int64).reshape(-1)
    dst = np.asarray(dst, dtype=np.int64).reshape(-1)
    if len(src) != len(dst):
        raise ValueError("src and dst differ in length")
    m = len(src)
    w = None if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1)
    if w is None and m > 0:
        return _from_edges_unweighted(src, dst, directed)
    inter = np.empty(2 * m, dtype=np.int64)
    inter[0::2], inter[1::2] = src, dst
    uniq, first = np.unique(inter, return_index=True)
    nodes = uniq[np.argsort(first, kind="stable")]
    pos_of = np.empty(len(uniq), dtype=np.int64)  # node-order position by sorted-rank
    pos_of[np.searchsorted(uniq, nodes)] = np.arange(len(nodes))

    # directed multigraph lines -> DiGraph entries (last line wins)
    su, sv = np.searchsorted(uniq, src), np.searchsorted(uniq, dst)
    key = su * np.int64(len(uniq)) + sv
    _, last_rev = np.unique(key[::-1], return_index=True)
    keep = np.sort(m - 1 - last_rev)
    su, sv = su[keep], sv[keep]
    wk = None if w is None else w[keep]
    if directed:
        return _assemble(nodes, uniq[su], uniq[sv], wk, True)

    # to_undirected(): entries visited in (source node order, insertion order); each visit
    # assigns both und[u][v] and und[v][u]; the last visit of an unordered pair wins.
    a, b = np.minimum(su, sv), np.maximum(su, sv)
    pkey = a * np.int64(len(uniq)) + b
    visit = np.lexsort((keep, pos_of[su]))  # visiting order of the DiGraph entries
    pk_v = pkey[visit]
    _, last_rev = np.unique(pk_v[::-1], return_index=True)
    win = visit[len(visit) - 1 - last_rev]  # winning entry per unordered pair
    a, b = a[win], b[win]
    ww = None if wk is None else wk[win]
    loop = a == b
    s2 = np.concatenate([a, b[~loop]])
    d2 = np.concatenate([b, a[~loop]])
    w2 = None if ww is None else np.concatenate([ww, ww[~loop]])
    return _assemble(nodes, uniq[s2], uniq[d2], w2, False)


def _from_edges_unweighted(src, dst, directed):
    """Unweighted fast path (sort-only, no argsort): duplicates and reciprocal lines collapse
    to one adjacency entry of weight 1 whatever their order."""
    m = len(src)
    inter = np.empty(2 * m, dtype=np.int64)
    inter[0::2], inter[1::2] = src, dst
    uniq = np.unique(inter)
    n = len(uniq)
    if n >= 2**31 or 2 * m >= 2**32:
        raise ValueError("graph too large for int32 dense ids")
    if uniq[0] == 0 and uniq[-1] == n - 1:
        dense = inter  # labels are already 0..n-1
    else:
        dense = np.searchsorted(uniq, inter)
    # first appearance of every node: sort (dense id, position) pairs packed in one int64
    packed = np.sort((dense << np.int64(32)) | np.arange(2 * m, dtype=np.int64))
    ids = packed >> np.int64(32)
    head = np.ones(2 * m, dtype=bool)
    head[1:] = ids[1:] != ids[:-1]
    first_pos = packed[head] & np.int64(0xFFFFFFFF)
    start_order = np.argsort(first_pos, kind="stable").astype(np.int32)
    su, sv = dense[0::2], dense[1::2]
    key = su * np.int64(n) + sv
    if not directed:
        key = np.concatenate([key, sv * np.int64(n) + su])
    key = np.unique(key)
    s = key // n
    d = (key - s * n).astype(np.int32)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(s, minlength=n), out=row_ptr[1:])
    return CsrGraph(uniq, row_ptr, d, None, start_order, directed)


def read_edgelist(path, weighted=False, directed=False):
    """Text edgelist as nx.read_edgelist(nodetype=int[, data=(('weight',float),)]) reads it
    in src/main.py:70-76: whitespace-separated, '#' starts a comment."""
    src, dst, w = [], [], []
    with open(path, "r") as f:
        for line in f:
            p = line.find("#")
            if p >= 0:
                line = line[:p]
            tok = line.split()
            if len(tok) < 2:
                continue
            src.append(int(tok[0]))
            dst.append(int(tok[1]))
            if weighted:
                w.append(float(tok[2]))
    return from_edges(src, dst, w if weighted else None, directed)


def degree_cut_for_budget(csr, budget_bytes, bytes_per_slot=32):
    """Tables under a memory budget (WalkEngine.preprocess(budget_bytes=...)): the largest degree D such that the
    edge tables of all entries (src -> dst) with deg(dst) <= D fit in `budget_bytes`, and the number of slots they
    take.  A table (src -> dst) has deg(dst) slots and there is one per in-edge of dst, so destinations of degree d
    cost sum over them of deg * indeg slots.  Host arithmetic only."""
    hdeg = np.diff(csr.row_ptr).astype(np.int64)
    indeg = hdeg if not csr.directed else np.bincount(csr.col, minlength=csr.n_nodes).astype(np.int64)
    max_degree = int(hdeg.max()) if csr.n_nodes else 0
    per_deg = np.zeros(max_degree + 1, dtype=np.int64)
    np.add.at(per_deg, hdeg, hdeg * indeg)
    cum = np.cumsum(per_deg)
    fit = np.nonzero(cum * int(bytes_per_slot) <= int(budget_bytes))[0]
    cut = int(fit[-1]) if len(fit) else 0
    return cut, int(cum[cut])
