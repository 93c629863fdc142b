"""BiNE path on MI355X: bipartite graph in HBM, every arithmetic step in the HIP kernels behind
include/n2v_bine.h (BASELINE config 5).

Replaces the pipeline of the reference's src/bine_train.py:433-515 (`train`):
  GraphUtils.construct_training_graph   (src/bine_graph_utils.py:32-58)   -> BipartiteGraph (host, numpy)
  calculate_centrality -> nx.hits       (src/bine_graph_utils.py:60-86)   -> BineEngine.calculate_centrality
  homogeneous_graph_random_walks_for_large_bipartite_graph (:112-131)     -> BineEngine.generate_walks
  get_negs (LSH pools, src/bine_lsh.py)                                   -> BineEngine.build_negative_pools
  get_context_and_negatives (:150-191)                                    -> BineEngine.build_occurrences (index only;
                                                                             windows/negatives are formed in the kernel)
  init_embedding_vectors, the max_iter loop (src/bine_train.py:183-206,454-504) -> init_embeddings / train

torch is memory, streams and index plumbing (prefix sums, sort of token positions by vertex); there is no
CPU path: without a GPU every device entry point raises.
"""
import numpy as np
import torch

from . import _lib

AUTO_STORE_MIN_VERTICES = 131072  # mode='parallel': context rows by atomics below, by whole-row stores above
MAX_WALK_LEN = 256  # cap of the geometric walk length (P(len > 256) = 0.85^255 ~ 1e-18 at the default 0.15)


def derive_seed(seed, k):
    """Independent 64-bit Philox keys for the pipeline's stages (splitmix64 of seed + k)."""
    x = (int(seed) + int(k) * 0x9E3779B97F4A7C15 + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


SEED_WALK_U, SEED_WALK_V, SEED_POOL_U, SEED_POOL_V, SEED_INIT, SEED_OCC, SEED_NEG = range(1, 8)


class BineConvergenceError(RuntimeError):
    """networkx 1.11 raises NetworkXError when hits() does not converge in max_iter iterations."""


def _unique_inverse(labels):
    """np.unique(labels, return_inverse=True); integer labels of moderate range skip the sort."""
    labels = np.asarray(labels)
    if labels.dtype.kind in "iu" and labels.size and labels.min() >= 0 and labels.max() < 8 * labels.size + 1024:
        present = np.bincount(labels) > 0
        rank = np.cumsum(present) - 1
        return np.nonzero(present)[0].astype(labels.dtype), rank[labels]
    return np.unique(labels, return_inverse=True)


class BipartiteGraph:
    """Host-side bipartite rating graph (src/bine_graph_utils.py:32-58).

    Vertices: users 0..n_u-1 and items n_u..n_u+n_v-1, each side in ascending label order
    (`node_u.sort()`, :53-54).  One symmetric CSR (rows ascending) with fp64 ratings; the rating list keeps
    file order and duplicates (`edge_list`, :45,58) while a repeated (user, item) pair takes its LAST rating
    everywhere (`edge_dict_u[user][item] = rating`, :46; add_weighted_edges_from, :57)."""

    def __init__(self, users, items, ratings):
        users = np.asarray(users)
        items = np.asarray(items)
        ratings = np.asarray(ratings, dtype=np.float64)
        if not (len(users) == len(items) == len(ratings)):
            raise ValueError("users, items and ratings must have the same length")
        self.user_labels, eu = _unique_inverse(users)
        self.item_labels, ev = _unique_inverse(items)
        self.n_u, self.n_v = len(self.user_labels), len(self.item_labels)
        self.n = self.n_u + self.n_v
        E = len(eu)
        # last rating of every distinct pair
        key = eu.astype(np.int64) * max(self.n_v, 1) + ev
        order = np.argsort(key, kind="stable")
        ks = key[order]
        last = np.ones(E, dtype=bool)
        last[:-1] = ks[1:] != ks[:-1]
        pair_key = ks[last]
        pair_w = ratings[order][last]
        group = np.empty(E, dtype=np.int64)                  # index of each rating's pair in the distinct list
        group[order] = np.cumsum(np.concatenate([[0], last[:-1]]))
        self.edge_u = eu.astype(np.int32)
        self.edge_v = (ev + self.n_u).astype(np.int32)
        self.edge_w = pair_w[group]
        # visited_u / visited_v (src/bine_train.py:458-487): a vertex is handled at its first rating
        first = np.zeros(E, dtype=np.uint8)
        if E:
            back = np.arange(E - 1, -1, -1)
            for ids, n_side, bit in ((eu, self.n_u, 1), (ev, self.n_v, 2)):
                pos = np.empty(n_side, dtype=np.int64)
                pos[ids[::-1]] = back                        # the last write per vertex is its smallest index
                first[pos] |= bit
        self.first = first
        # symmetric CSR over users + items.  The distinct pairs are already ordered by (user, item): that IS the
        # users' half; the items' half is the same list stably re-ordered by item (users stay ascending).
        pu = (pair_key // max(self.n_v, 1)).astype(np.int64)
        pv = (pair_key % max(self.n_v, 1)).astype(np.int64)
        by_item = np.argsort(pv, kind="stable")
        self.col = np.concatenate([pv + self.n_u, pu[by_item]]).astype(np.int32)
        self.w = np.concatenate([pair_w, pair_w[by_item]])
        self.row_ptr = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum(np.concatenate([np.bincount(pu, minlength=self.n_u), np.bincount(pv, minlength=self.n_v)]),
                  out=self.row_ptr[1:])

    @property
    def n_ratings(self):
        return len(self.edge_u)

    @classmethod
    def read(cls, filename):
        """`user<TAB>item<TAB>rating` lines (src/bine_graph_utils.py:37-49)."""
        users, items, ratings = [], [], []
        with open(filename, encoding="UTF-8") as fin:
            for line in fin:
                if not line.strip():
                    continue
                user, item, rating = line.strip().split("\t")
                users.append(user)
                items.append(item)
                ratings.append(float(rating))
        return cls(users, items, ratings)

    def label_of(self, v):
        return self.user_labels[v] if v < self.n_u else self.item_labels[v - self.n_u]


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise RuntimeError("n2v_hip.bine: no GPU visible (torch.cuda.is_available() is False); "
                           "this engine has no CPU fallback")
    return torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())


class BineEngine:
    """HBM-resident BiNE state for one bipartite graph."""

    def __init__(self, graph: BipartiteGraph, device=None, seed=0):
        self.lib = _lib.load()
        self.device = d = _require_gpu(device)
        self.g = graph
        self.seed = int(seed)
        self.row_ptr = torch.from_numpy(graph.row_ptr).to(d)
        self.col = torch.from_numpy(graph.col).to(d)
        self.w = torch.from_numpy(graph.w).to(d)
        self.edge_u = torch.from_numpy(graph.edge_u).to(d)
        self.edge_v = torch.from_numpy(graph.edge_v).to(d)
        self.edge_w = torch.from_numpy(graph.edge_w).to(d)
        self.first = torch.from_numpy(graph.first).to(d)
        deg = self.row_ptr[1:] - self.row_ptr[:-1]
        # two-hop path prefix: cum2[e] = number of two-hop paths through CSR entries before e
        self.cum2 = torch.zeros(graph.col.shape[0] + 1, dtype=torch.int64, device=d)
        if graph.col.shape[0]:
            torch.cumsum(deg[self.col.long()], 0, out=self.cum2[1:])
        self.authority = self.auth_scaled = self.counts = None
        self.tokens = self.tok_walk = self.walk_off = self.walk_node = None
        self.n_walks = (0, 0)
        self.pool = None
        self.occ_ptr = self.occ_pos = None
        self.emb = self.ctx = None
        self.state = torch.zeros(8, dtype=torch.float64, device=d)
        self.losses = []

    def _stream(self):
        return _lib.stream_ptr(self.device)

    def _seed(self, k):
        return derive_seed(self.seed, k)

    # ------------------------------------------------------------------ user-user edges (HITS input only)
    def add_user_edges(self, src_user, dst_user, weight):
        """`gul.G.add_weighted_edges_from(add_edges)` of src/bine_train.py:620: weighted user-user edges join the
        networkx graph that calculate_centrality() hands to hits().  Nothing else of the pipeline reads them
        (the projections come from the biadjacency of node_u x node_v, :114; edge_list / edge_dict_u are not
        rebuilt), so only the HITS matrix changes.  Edges are applied in order; a repeated unordered pair keeps
        its last weight (networkx semantics).  src_user / dst_user: user indices (0..n_u-1)."""
        g = self.g
        a = np.asarray(src_user, dtype=np.int64)
        b = np.asarray(dst_user, dtype=np.int64)
        w = np.asarray(weight, dtype=np.float64)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        key = lo * g.n + hi
        _, last_rev = np.unique(key[::-1], return_index=True)
        win = len(key) - 1 - last_rev
        lo, hi, w = lo[win], hi[win], w[win]
        loop = lo == hi
        src = np.concatenate([np.repeat(np.arange(g.n), np.diff(g.row_ptr)), lo, hi[~loop]])
        dst = np.concatenate([g.col.astype(np.int64), hi, lo[~loop]])
        ww = np.concatenate([g.w, w, w[~loop]])
        o = np.lexsort((dst, src))
        rp = np.zeros(g.n + 1, dtype=np.int64)
        np.cumsum(np.bincount(src, minlength=g.n), out=rp[1:])
        d = self.device
        self.hits_csr = (torch.from_numpy(rp).to(d), torch.from_numpy(dst[o].astype(np.int32)).to(d),
                         torch.from_numpy(ww[o]).to(d))
        self.authority = None
        return int(len(lo))

    hits_csr = None  # (row_ptr, col, w) of the graph handed to hits() when user-user edges were added

    # ------------------------------------------------------------------ centrality
    def calculate_centrality(self, max_iter=100, tol=1.0e-8):
        """nx.hits(G) of networkx 1.11 (src/bine_graph_utils.py:61): authority scores by power iteration from
        h = 1/n, max-normalised every iteration, stop at sum|h - h_last| < tol, error after max_iter."""
        n, d, lib = self.g.n, self.device, self.lib
        with torch.cuda.device(d):
            h = torch.full((n,), 1.0 / n, dtype=torch.float64, device=d)
            hn = torch.empty_like(h)
            a = torch.empty_like(h)
            st = torch.zeros(1, dtype=torch.float64, device=d)
            p = _lib.ptr
            rp, col, w = self.hits_csr if self.hits_csr is not None else (self.row_ptr, self.col, self.w)
            for it in range(max_iter):
                _lib.check(lib.n2v_bine_spmv(n, p(rp), p(col), p(w), p(h), p(a), self._stream()))
                _lib.check(lib.n2v_bine_spmv(n, p(rp), p(col), p(w), p(a), p(hn), self._stream()))
                _lib.check(lib.n2v_bine_hits_normalise(n, p(hn), p(a), p(h), p(st), self._stream()))
                h, hn = hn, h
                if float(st.item()) < tol:
                    self.authority = a
                    self.hits_iterations = it + 1
                    return a
        raise BineConvergenceError("HITS: power iteration failed to converge in %d iterations" % max_iter)

    # ------------------------------------------------------------------ walks
    def generate_walks(self, percentage=0.15, maxT=32, minT=1, max_len=MAX_WALK_LEN):
        """walk_generator (src/bine_train.py:210-222) for --large 1: max(ceil(maxT * authority), minT) restart
        walks from every vertex of each side, on that side's projection.  Result: one ragged token array
        (users' walks, then items'), `walk_off`, `walk_node`, `tok_walk`."""
        if self.authority is None:
            self.calculate_centrality()
        g, d, lib, p = self.g, self.device, self.lib, _lib.ptr
        with torch.cuda.device(d):
            counts = torch.zeros(g.n, dtype=torch.int32, device=d)
            auth = torch.zeros(g.n, dtype=torch.float64, device=d)
            for lo, hi in ((0, g.n_u), (g.n_u, g.n)):
                _lib.check(lib.n2v_bine_walk_counts(p(self.authority), lo, hi, int(maxT), int(minT), p(counts), p(auth),
                                                    self._stream()))
            self.counts, self.auth_scaled = counts, auth
            ids = torch.arange(g.n, dtype=torch.int32, device=d)
            walk_node = torch.repeat_interleave(ids, counts.long())
            nw_u = int(counts[: g.n_u].sum().item())
            nw = int(walk_node.shape[0])
            self.n_walks = (nw_u, nw - nw_u)
            lens = torch.empty(nw, dtype=torch.int32, device=d)
            sides = ((0, nw_u, self._seed(SEED_WALK_U)), (nw_u, nw - nw_u, self._seed(SEED_WALK_V)))
            for base, cnt, seed in sides:
                if cnt:
                    _lib.check(lib.n2v_bine_walk_lengths(p(self.row_ptr), p(self.cum2), walk_node[base:].data_ptr(), cnt, 0,
                                                         float(percentage), int(max_len), seed, lens[base:].data_ptr(),
                                                         self._stream()))
            walk_off = torch.zeros(nw + 1, dtype=torch.int64, device=d)
            torch.cumsum(lens.long(), 0, out=walk_off[1:])
            n_tok = int(walk_off[-1].item())
            tokens = torch.empty(n_tok, dtype=torch.int32, device=d)
            for base, cnt, seed in sides:
                if cnt:
                    _lib.check(lib.n2v_bine_walk(p(self.row_ptr), p(self.col), p(self.cum2), walk_node[base:].data_ptr(),
                                                 walk_off[base:].data_ptr(), cnt, 0, seed, p(tokens), self._stream()))
            self.walk_node, self.walk_off, self.tokens = walk_node, walk_off, tokens
            self.tok_walk = torch.repeat_interleave(torch.arange(nw, dtype=torch.int32, device=d), lens.long())
            self.occ_ptr = self.occ_pos = None
        return self

    def walks_as_lists(self, side):
        """The walks of one side ('u' or 'v') as lists of labels (gul.walks_u / gul.walks_v)."""
        nw_u, nw_v = self.n_walks
        lo, hi = (0, nw_u) if side == "u" else (nw_u, nw_u + nw_v)
        off = self.walk_off.cpu().numpy()
        tok = self.tokens.cpu().numpy()
        return [[self.g.label_of(int(t)) for t in tok[off[i]:off[i + 1]]] for i in range(lo, hi)]

    # ------------------------------------------------------------------ negatives, contexts
    def build_negative_pools(self, pool_size=200, max_jaccard=None, method=None, k=200):
        """get_negs (src/bine_graph_utils.py:145-148 -> src/bine_lsh.py:22-51): per vertex up to `pool_size` vertices of
        its own side that are not similar to it.

        method="lsh" (default): the reference's pipeline — MinHash(128) signatures of the neighbour labels, a
        MinHashLSHForest(l=8), `forest.query(ms[i], k)`, clusters of the `visted` sweep sharing one pool, the pool a
        sample of the side minus sim(i) and sim(j), j in sim(i) (datasketch 1.2.5 restated).
        method="jaccard" (chosen when `max_jaccard` is given): independent pools, a candidate refused when its exact
        Jaccard similarity exceeds `max_jaccard` — the round-1 stand-in, kept for comparison."""
        if method is None:
            method = "jaccard" if max_jaccard is not None else "lsh"
        if method == "lsh":
            return self._build_lsh_pools(int(pool_size), int(k))
        if method != "jaccard":
            raise ValueError("method must be 'lsh' or 'jaccard'")
        max_jaccard = 1.0 / 128.0 if max_jaccard is None else max_jaccard
        g, d, lib, p = self.g, self.device, self.lib, _lib.ptr
        with torch.cuda.device(d):
            pool = torch.empty((g.n, int(pool_size)), dtype=torch.int32, device=d)
            for lo, hi, k_ in ((0, g.n_u, SEED_POOL_U), (g.n_u, g.n, SEED_POOL_V)):
                if hi > lo:
                    _lib.check(lib.n2v_bine_neg_pools(p(self.row_ptr), p(self.col), lo, hi, lo, hi, int(pool_size),
                                                      float(max_jaccard), self._seed(k_), pool[lo:].data_ptr(),
                                                      self._stream()))
            self.pool = pool
        return self

    # -- MinHash LSH forest (include/n2v_bine.h, csrc/n2v_lsh.hip)
    def label_hashes(self):
        """hv[v] = datasketch's 32-bit SHA-1 value of vertex v's label as the reference feeds it to MinHash.update
        (`d.encode('utf8')`, src/bine_lsh.py:16), computed on the device from the packed label bytes."""
        g, d = self.g, self.device
        labels = np.concatenate([np.asarray(g.user_labels).astype(str), np.asarray(g.item_labels).astype(str)])
        raw = np.char.encode(labels, "utf8")                     # fixed-width 'S' array, zero padded
        width = max(raw.dtype.itemsize, 1)
        lens = np.char.str_len(raw).astype(np.int32)
        mat = np.frombuffer(raw.tobytes(), dtype=np.uint8).reshape(len(labels), width) if len(labels) else np.zeros((0, 1), np.uint8)
        with torch.cuda.device(d):
            b = torch.from_numpy(np.array(mat)).to(d)
            ln = torch.from_numpy(lens).to(d)
            hv = torch.empty(g.n, dtype=torch.int32, device=d)   # uint32 bits
            _lib.check(self.lib.n2v_lsh_sha1_labels(_lib.ptr(b), int(width), _lib.ptr(ln), g.n, _lib.ptr(hv), self._stream()))
        return hv

    @staticmethod
    def minhash_permutations(num_perm=128, seed=1):
        """datasketch 1.2.5 MinHash.__init__: (a_j, b_j) from numpy's legacy RandomState(seed), drawn alternately."""
        gen = np.random.RandomState(seed)
        m = (1 << 61) - 1
        ab = np.array([(gen.randint(1, m, dtype=np.uint64), gen.randint(0, m, dtype=np.uint64)) for _ in range(num_perm)],
                      dtype=np.uint64)
        return np.ascontiguousarray(ab[:, 0]), np.ascontiguousarray(ab[:, 1])

    def minhash_signatures(self, hv=None):
        """int32 (uint32 bits) [n][128]: the MinHash of every vertex's neighbour set (src/bine_lsh.py:13-17)."""
        g, d, p = self.g, self.device, _lib.ptr
        hv = self.label_hashes() if hv is None else hv
        a, b = self.minhash_permutations()
        with torch.cuda.device(d):
            pa = torch.from_numpy(a.view(np.int64)).to(d)
            pb = torch.from_numpy(b.view(np.int64)).to(d)
            sig = torch.empty((g.n, 128), dtype=torch.int32, device=d)
            _lib.check(self.lib.n2v_lsh_minhash(p(self.row_ptr), p(self.col), p(hv), p(pa), p(pb), 0, g.n, p(sig), self._stream()))
        return sig

    def _forest_side(self, sig_side, k):
        """One side's forest and all its queries.  torch: the eight sorted orders and the prefix ranges (sorts, scans);
        HIP: the queries.  Returns (sim int32[n][k], sim_n int32[n])."""
        d, p, lib = self.device, _lib.ptr, self.lib
        n = int(sig_side.shape[0])
        depth = 16
        vals = sig_side.long() & 0xFFFFFFFF                       # unsigned order
        idx = torch.arange(n, device=d)
        order = torch.empty((8, n), dtype=torch.int32, device=d)
        lo = torch.empty((n, 8, depth), dtype=torch.int32, device=d)
        hi = torch.empty((n, 8, depth), dtype=torch.int32, device=d)
        levels = torch.arange(1, depth + 1, device=d).unsqueeze(1)
        for t in range(8):
            s = vals[:, t * depth:(t + 1) * depth]
            # two 32-bit values per signed 64-bit sort key, order preserved: ((a - 2^31) << 32) | b
            pairs = ((s[:, 0::2] - (1 << 31)) << 32) | s[:, 1::2]
            o = idx
            for c in range(depth // 2 - 1, -1, -1):                 # lexicographic, stable: ties keep insertion order
                o = o[torch.sort(pairs[o, c], stable=True).indices]
            order[t] = o.int()
            so = s[o]
            lcp = torch.zeros(n, dtype=torch.int64, device=d)
            if n > 1:
                lcp[1:] = torch.cumprod((so[1:] == so[:-1]).long(), dim=1).sum(dim=1)
            pos = torch.empty(n, dtype=torch.int64, device=d)
            pos[o] = idx
            cut = lcp.unsqueeze(0) < levels                         # [r-1][i]: a new prefix-r group starts at i (always at 0)
            start = torch.cummax(torch.where(cut, idx.unsqueeze(0), torch.zeros_like(cut, dtype=torch.int64)), 1).values
            nxt = torch.where(cut, idx.unsqueeze(0), torch.full_like(cut, n, dtype=torch.int64))
            after = torch.flip(torch.cummin(torch.flip(nxt, [1]), 1).values, [1])       # first cut at or after i
            end = torch.cat([after[:, 1:], torch.full((depth, 1), n, dtype=torch.int64, device=d)], 1)
            lo[:, t, :] = start[:, pos].t().int()
            hi[:, t, :] = end[:, pos].t().int()
        sim = torch.empty((n, k), dtype=torch.int32, device=d)
        sim_n = torch.empty(n, dtype=torch.int32, device=d)
        _lib.check(lib.n2v_lsh_forest_query(p(order), p(lo), p(hi), n, k, p(sim), p(sim_n), self._stream()))
        return sim, sim_n

    def _cluster_owners(self, sim, sim_n):
        """The `visted` sweep (src/bine_lsh.py:32-36,41-45): owner[i] = the vertex whose turn produced i's pool."""
        d, p, lib = self.device, _lib.ptr, self.lib
        n, k = int(sim.shape[0]), int(sim.shape[1])
        src = torch.arange(n, device=d).unsqueeze(1).expand(n, k)
        valid = (torch.arange(k, device=d).unsqueeze(0) < sim_n.unsqueeze(1)) & (sim.long() > src)
        l, j = src[valid], sim.long()[valid]
        key = torch.sort(j * n + l).values
        rev_src = (key % n).int()
        rev_ptr = torch.zeros(n + 1, dtype=torch.int64, device=d)
        torch.cumsum(torch.bincount(key // n, minlength=n), 0, out=rev_ptr[1:])
        owner = torch.full((n,), -1, dtype=torch.int32, device=d)
        open_ = torch.zeros(1, dtype=torch.int32, device=d)
        for _ in range(n + 1):
            open_.zero_()
            _lib.check(lib.n2v_lsh_leader_round(p(rev_ptr), p(rev_src), n, p(owner), p(open_), self._stream()))
            if int(open_.item()) == 0:
                break
        else:
            raise RuntimeError("n2v_lsh_leader_round: the sweep did not resolve")
        return owner

    def _build_lsh_pools(self, pool_size, k, groups=1024):
        import time
        g, d, p, lib = self.g, self.device, _lib.ptr, self.lib
        secs = {"signatures": 0.0, "forest": 0.0, "clusters": 0.0, "pools": 0.0}

        def lap(name, t0):
            torch.cuda.synchronize(d)
            secs[name] += time.perf_counter() - t0
            return time.perf_counter()

        with torch.cuda.device(d):
            t = time.perf_counter()
            sig = self.minhash_signatures()
            t = lap("signatures", t)
            pool = torch.full((g.n, pool_size), -1, dtype=torch.int32, device=d)
            self.lsh = {}
            for name, lo, hi, kseed in (("u", 0, g.n_u, SEED_POOL_U), ("v", g.n_u, g.n, SEED_POOL_V)):
                n_side = hi - lo
                if n_side <= 0:
                    continue
                t = time.perf_counter()
                sim, sim_n = self._forest_side(sig[lo:hi], k)
                t = lap("forest", t)
                owner = self._cluster_owners(sim, sim_n)
                lead = torch.nonzero(owner == torch.arange(n_side, device=d, dtype=torch.int32)).flatten().int()
                t = lap("clusters", t)
                words = (n_side + 31) // 32
                n_groups = int(min(groups, max(int(lead.shape[0]), 1)))
                bitmap = torch.zeros((n_groups, words), dtype=torch.int32, device=d)
                side_pool = pool[lo:hi]
                _lib.check(lib.n2v_lsh_pools(p(sim), p(sim_n), k, p(lead), int(lead.shape[0]), n_side, pool_size,
                                             self._seed(kseed), lo, p(bitmap), words, n_groups, side_pool.data_ptr(),
                                             self._stream()))
                side_pool.copy_(side_pool[owner.long()])            # every member of a cluster holds its owner's pool
                t = lap("pools", t)
                self.lsh[name] = dict(sim=sim, sim_n=sim_n, owner=owner, clusters=int(lead.shape[0]))
            self.lsh["signatures"] = sig
            self.lsh["seconds"] = secs
            self.pool = pool
        return self

    def build_occurrences(self):
        """Index of every vertex's occurrences in the walks (the keys and list positions of the reference's
        context_dict, src/bine_graph_utils.py:161-187): token positions grouped by vertex, ascending."""
        if self.tokens is None:
            raise RuntimeError("generate_walks() first")
        with torch.cuda.device(self.device):
            tok = self.tokens.long()
            self.occ_pos = torch.argsort(tok, stable=True)
            self.occ_ptr = torch.zeros(self.g.n + 1, dtype=torch.int64, device=self.device)
            torch.cumsum(torch.bincount(tok, minlength=self.g.n), 0, out=self.occ_ptr[1:])
        return self

    # ------------------------------------------------------------------ embeddings
    def init_embeddings(self, d=128):
        """init_embedding_vectors (src/bine_train.py:183-206): U[0,1)^d rows scaled to unit l2 norm for the
        embedding and the context table of every vertex."""
        self.dim = int(d)
        self.stride = next(s for s in (64, 128, 256, 512) if s >= self.dim) if self.dim <= 512 else None
        if self.stride is None:
            raise ValueError("d must be <= 512")
        with torch.cuda.device(self.device):
            self.emb = torch.empty((self.g.n, self.stride), dtype=torch.float64, device=self.device)
            self.ctx = torch.empty_like(self.emb)
            _lib.check(self.lib.n2v_bine_init(_lib.ptr(self.emb), _lib.ptr(self.ctx), self.g.n, self.dim, self.stride,
                                              self._seed(SEED_INIT), self._stream()))
        return self

    def _prepare(self, mode):
        if self.emb is None:
            self.init_embeddings()
        if self.pool is None:
            self.build_negative_pools()
        if self.occ_ptr is None:
            self.build_occurrences()
        if mode == "parallel":   # like the SGNS row-sharing policy: lossless atomics while every row is hot
            mode = "atomic" if self.g.n < AUTO_STORE_MIN_VERTICES else "store"
        self.mode_used = mode
        return {"sequential": _lib.BINE_SEQUENTIAL, "atomic": _lib.BINE_PARALLEL, "store": _lib.BINE_PARALLEL_STORE}[mode]

    def reset_schedule(self, lam=0.01):
        """Start of `train`: learning rate lam, last_loss = 0 (src/bine_train.py:434,452)."""
        self.state.copy_(torch.tensor([lam, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float64))
        self.losses = []

    def train_pass(self, iteration, alpha=0.01, beta=0.01, gamma=0.1, ws=5, ns=4, mode="parallel", max_blocks=0,
                   e_range=None):
        """One pass over the rating list (or this rank's range of it): src/bine_train.py:461-494."""
        md = self._prepare(mode)
        p = _lib.ptr
        e0, e1 = e_range if e_range is not None else (0, self.g.n_ratings)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.n2v_bine_train_pass(
                p(self.edge_u), p(self.edge_v), p(self.edge_w), p(self.first), e0, e1, p(self.emb), p(self.ctx),
                self.dim, self.stride, p(self.occ_ptr), p(self.occ_pos), p(self.tokens), p(self.tok_walk),
                p(self.walk_off), p(self.pool), int(self.pool.shape[1]), int(ws), int(ns), float(alpha), float(beta),
                float(gamma), p(self.state), int(iteration), self._seed(SEED_OCC), self._seed(SEED_NEG), md,
                int(max_blocks), self._stream()))

    def finish_iteration(self, epsilon=1e-3):
        """src/bine_train.py:495-502: learning-rate step on the pass's loss; returns (loss, stop)."""
        with torch.cuda.device(self.device):
            loss_now = self.state[1:2].clone()
            _lib.check(self.lib.n2v_bine_lambda_step(_lib.ptr(self.state), float(epsilon), self._stream()))
            stop = self.state[3].item() != 0.0
        self.losses.append(float(loss_now.item()))
        return self.losses[-1], stop

    def train(self, max_iter=50, alpha=0.01, beta=0.01, gamma=0.1, lam=0.01, ws=5, ns=4, epsilon=1e-3,
              mode="parallel", max_blocks=0, e_range=None, first_iteration=0, merge=None):
        """The iteration loop of src/bine_train.py:452-504.  mode='sequential' reproduces the reference's update
        order with one wavefront (parity tests); 'atomic' / 'store' are the two parallel variants of
        include/n2v_bine.h, 'parallel' picks between them by graph size.  `merge(engine)` runs between a pass and
        its learning-rate step (multi-GPU: ReplicaMerge).  Returns the per-iteration losses; `self.lam` is the
        final learning rate."""
        self._prepare(mode)
        if first_iteration == 0:
            self.reset_schedule(lam)
        for it in range(first_iteration, first_iteration + max_iter):
            self.train_pass(it, alpha, beta, gamma, ws, ns, mode, max_blocks, e_range)
            if merge is not None:
                merge(self)
            if self.finish_iteration(epsilon)[1]:
                break
        if merge is not None and hasattr(merge, "flush"):
            merge.flush(self)          # an overlapped merge still owes the last pass's foreign changes
        self.lam = float(self.state[0].item())
        return self.losses

    def train_sharded(self, comm, rank, world, overlap=False, **kw):
        """One process per GPU (BASELINE config 5 on 8 GPUs): every rank holds the whole graph, walks, pools and
        a replica of both tables (all derived from the same seeds, so identical without communication), passes
        over its contiguous range of the rating list, and the replicas' changes are summed over RCCL before the
        learning-rate step — every rank then holds what one shared table would have received.  overlap=True: the
        2 GB all-reduce of a pass's changes runs UNDER the next pass (OverlappedReplicaMerge: the other ranks' changes
        arrive one pass late; sums commute, so nothing is lost or applied twice — but a pass then reads rows that lack
        the others' previous pass, so this is an opt-in whose loss curve has to be compared, not a bit-equal variant)."""
        per = -(-self.g.n_ratings // world)
        e0 = min(rank * per, self.g.n_ratings)
        self._prepare(kw.get("mode", "parallel"))
        merge = OverlappedReplicaMerge(self, comm) if overlap else ReplicaMerge(self, comm)
        return self.train(e_range=(e0, min(e0 + per, self.g.n_ratings)), merge=merge, **kw)

    # ------------------------------------------------------------------ results
    def vectors(self, side, which="embedding"):
        """float64 [n_side, d] (node_list_u[u]['embedding_vectors'] stacked in label order)."""
        t = self.emb if which == "embedding" else self.ctx
        lo, hi = (0, self.g.n_u) if side == "u" else (self.g.n_u, self.g.n)
        return t[lo:hi, : self.dim].cpu().numpy()


class ReplicaMerge:
    """Sum of the replicas' changes since the last merge, applied to every replica; the pass's loss (and the
    traffic counters) are summed too, so the learning-rate rule sees the whole list's loss.  `comm` needs
    all_reduce_sum(tensor) (n2v_hip.dist._Comm over RCCL, or gloo in rehearsals)."""

    def __init__(self, engine, comm):
        self.comm = comm
        self.base = [engine.emb.clone(), engine.ctx.clone()]

    def __call__(self, engine):
        wire = getattr(self.comm, "wire_dtype_f64", None)   # fp32 over RCCL: the CHANGES travel, half the bytes
        for t, base in zip((engine.emb, engine.ctx), self.base):
            t.sub_(base)
            if wire is not None:
                d = t.to(wire)
                self.comm.all_reduce_sum(d)
                t.copy_(d)
            else:
                self.comm.all_reduce_sum(t)
            t.add_(base)
            base.copy_(t)
        part = engine.state[1:2].clone()
        self.comm.all_reduce_sum(part)
        engine.state[1:2].copy_(part)


class OverlappedReplicaMerge:
    """ReplicaMerge with the big all-reduce off the critical path (BASELINE config 5 at 8 GPUs: 2 x 10^6 x 256 changes
    = 2.05 GB as fp32 per pass, against a 12 ms pass).  At the end of pass i a rank
      1. takes its own change of the pass, d_i = x - xs (xs = the table when the pass began),
      2. adds what the OTHER ranks changed in pass i-1 (S_{i-1} - d_{i-1}, whose all-reduce ran under pass i),
      3. starts the all-reduce of d_i (async_op: it runs on the communicator's stream under pass i+1).
    Every change is applied exactly once on every rank (its owner applies it while training, the others one pass
    later), so after flush() all ranks hold start + the sum of all changes — the same total as the synchronous merge,
    bit for bit when the sums are exact (tests/test_bine_host.py).  Only the small loss scalar is reduced synchronously:
    the learning-rate rule (src/bine_train.py:495-502) needs the whole list's loss of THIS pass."""

    def __init__(self, engine, comm):
        self.comm = comm
        self.xs = [engine.emb.clone(), engine.ctx.clone()]
        self.pending = None          # [(summed-change buffer, own change, handle)] of the previous pass

    def _fold_pending(self, tables):
        if self.pending is None:
            return
        for t, (buf, own, handle) in zip(tables, self.pending):
            if handle is not None:
                handle.wait()
            buf.sub_(own)            # what the other ranks changed
            t.add_(buf.to(t.dtype))
        self.pending = None

    def __call__(self, engine):
        wire = getattr(self.comm, "wire_dtype_f64", None)
        tables = (engine.emb, engine.ctx)
        own = [(t - xs).to(wire if wire is not None else t.dtype) for t, xs in zip(tables, self.xs)]
        self._fold_pending(tables)
        nxt = []
        for t, xs, d in zip(tables, self.xs, own):
            xs.copy_(t)
            buf = d.clone()
            nxt.append((buf, d, self.comm.all_reduce_async(buf)))
        self.pending = nxt
        part = engine.state[1:2].clone()
        self.comm.all_reduce_sum(part)
        engine.state[1:2].copy_(part)

    def flush(self, engine):
        self._fold_pending((engine.emb, engine.ctx))
        for t, xs in zip((engine.emb, engine.ctx), self.xs):
            xs.copy_(t)
