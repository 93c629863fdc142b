"""Drop-in for the reference's ``src/node2vec.py`` on MI355X.

Same surface — ``Graph(nx_G, is_directed, p, q, popwalk="none")``,
``.preprocess_transition_probs()``, ``.simulate_walks(num_walks, walk_length, nodes=None,
verbose=False)``, ``.simulate_walks_on_the_fly(...)``, ``.node2vec_walk(walk_length,
start_node)``, module-level ``alias_setup`` / ``alias_draw`` — with the tables and the
walks computed by the HIP kernels in ``csrc/`` through the C-ABI of ``include/n2v_hip.h``.

RNG contract (SURVEY.md 8(a) row 6'): the reference draws two ``np.random.rand()`` per step
from numpy's GLOBAL MT19937 state.  In the default ``rng="numpy"`` mode this class takes
exactly the uniforms the reference would take from that same global state (so a harness
that calls ``np.random.seed(s)`` first gets bit-identical walks, and the global state
afterwards is where the reference would have left it) and the kernel consumes them by
(walk, step) index.  ``rng="philox"`` is the throughput mode: uniforms are generated in
the kernel (Philox4x32-10 keyed by ``seed`` and the walk's global index), nothing is read
from numpy; the walk rule is identical.

Only ``popwalk="none"`` is implemented (the "pop" variants of the reference are an
experiment knob outside the hot path).
"""
import numpy as np
import torch

from n2v_hip import csr as _csr
from n2v_hip import mt19937 as _mt
from n2v_hip.engine import WalkEngine


# --------------------------------------------------------------------------- module-level API
def alias_setup(probs):
    """src/node2vec.py:240-269 — one table, built on the GPU by the kernel that builds
    every other table (n2v_alias_setup_tables)."""
    from n2v_hip.engine import alias_setup_device
    return alias_setup_device([float(x) for x in probs])


def alias_draw(J, q):
    """src/node2vec.py:271-281 — host helper kept for API compatibility (two draws from
    numpy's global stream, both always consumed)."""
    K = len(J)
    kk = int(np.floor(np.random.rand() * K))
    if np.random.rand() < q[kk]:
        return kk
    return J[kk]


class WalkCorpus:
    """The list of walks ``simulate_walks`` returns, kept on the device.

    Behaves like the reference's ``list[list[int]]`` (len, indexing, iteration, ==),
    holding node LABELS; ``walks``/``lens`` are the device tensors of DENSE ids that
    ``learn_embeddings`` consumes without a host round trip."""

    def __init__(self, walks, lens, labels):
        self.walks = walks    # int32 [W, L] device, padded with -1
        self.lens = lens      # int32 [W] device
        self.labels = labels  # int64 [N] host (dense id -> label)
        self._host = None

    def _materialise(self):
        if self._host is None:
            w = self.walks.cpu().numpy()
            n = self.lens.cpu().numpy()
            lab = self.labels
            if (n == w.shape[1]).all():
                self._host = lab[w].tolist() if w.size else [[] for _ in range(len(n))]
            else:
                self._host = [lab[row[:k]].tolist() for row, k in zip(w, n)]
        return self._host

    def tolist(self):
        return self._materialise()

    def __len__(self):
        return int(self.walks.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._materialise()[i]
        k = int(self.lens[i].item())
        return self.labels[self.walks[i, :k].cpu().numpy()].tolist()

    def __iter__(self):
        return iter(self._materialise())

    def __eq__(self, other):
        if isinstance(other, WalkCorpus):
            other = other.tolist()
        return self._materialise() == other

    def __add__(self, other):
        return self._materialise() + list(other)

    def __repr__(self):
        return "WalkCorpus(%d walks x %d)" % tuple(self.walks.shape)


def as_corpus(walks, device=None):
    """A WalkCorpus as is; a list of lists of node labels (e.g. walks reloaded from a walk
    file, src/main_link.py:345-349) is densified and moved to the device."""
    if isinstance(walks, WalkCorpus):
        return walks
    if not torch.cuda.is_available():
        raise RuntimeError("n2v_hip: no GPU visible; there is no CPU fallback")
    rows = [np.asarray(list(w), dtype=np.int64) for w in walks]
    L = max([len(r) for r in rows] + [1])
    labels = np.unique(np.concatenate(rows)) if rows else np.zeros(0, dtype=np.int64)
    dense = np.full((len(rows), L), -1, dtype=np.int32)
    lens = np.zeros(len(rows), dtype=np.int32)
    for i, r in enumerate(rows):
        dense[i, :len(r)] = np.searchsorted(labels, r)
        lens[i] = len(r)
    d = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    return WalkCorpus(torch.from_numpy(dense).to(d), torch.from_numpy(lens).to(d), labels)


class _AliasNodes:
    """Dict-like view of the node tables: ``alias_nodes[node] -> (J, q)``."""

    def __init__(self, graph):
        self._g = graph

    def __getitem__(self, node):
        g = self._g
        return g._engine.node_table(int(g._csr.dense_of([node])[0]))

    def __contains__(self, node):
        try:
            self._g._csr.dense_of([node])
            return True
        except (KeyError, TypeError, ValueError):
            return False

    def __len__(self):
        return self._g._csr.n_nodes

    def keys(self):
        return [int(x) for x in self._g._csr.labels[self._g._csr.start_order]]

    def __iter__(self):
        return iter(self.keys())

    def items(self):
        return ((k, self[k]) for k in self.keys())


class _AliasEdges:
    """Dict-like view of the edge tables: ``alias_edges[(src, dst)] -> (J, q)``; keys are the
    directed adjacency entries (both orientations of an undirected edge), as in
    src/node2vec.py:193-199."""

    def __init__(self, graph):
        self._g = graph

    def _entry(self, key):
        g = self._g
        try:
            u, v = key
            du, dv = g._csr.dense_of([u])[0], g._csr.dense_of([v])[0]
        except (KeyError, TypeError, ValueError):
            raise KeyError(key)
        e = g._engine.edge_index(int(du), int(dv))
        if e < 0:
            raise KeyError(key)
        return e

    def __getitem__(self, key):
        return self._g._engine.edge_table(self._entry(key))

    def __contains__(self, key):
        try:
            self._entry(key)
            return True
        except KeyError:
            return False

    def __len__(self):
        return self._g._csr.nnz

    def keys(self):
        c = self._g._csr
        lab = c.labels
        return list(zip(lab[c.src_of()].tolist(), lab[c.col].tolist()))

    def __iter__(self):
        return iter(self.keys())

    def items(self):
        return ((k, self[k]) for k in self.keys())


class Graph():
    def __init__(self, nx_G, is_directed, p, q, popwalk="none", device=None, rng="numpy", seed=0):
        self.G = nx_G
        self.is_directed = is_directed
        self.p = p
        self.q = q
        self.popwalk = popwalk
        self.device = device
        self.rng = rng
        self.seed = seed
        self._csr_cache = None
        self._engine = None

    # construction without networkx for graphs too large for a dict-of-dict container
    @classmethod
    def from_csr(cls, csr_graph, p, q, device=None, rng="philox", seed=0):
        g = cls(None, csr_graph.directed, p, q, device=device, rng=rng, seed=seed)
        g._csr_cache = csr_graph
        return g

    @property
    def _csr(self):
        if self._csr_cache is None:
            if isinstance(self.G, _csr.CsrGraph):
                self._csr_cache = self.G
            else:
                self._csr_cache = _csr.from_networkx(self.G)
        return self._csr_cache

    def _check_popwalk(self):
        if self.popwalk != "none":
            raise NotImplementedError("popwalk=%r: only the 'none' walk is on the MI355X hot path" % (self.popwalk,))

    # src/node2vec.py:176-204
    def preprocess_transition_probs(self, budget_bytes=None):
        """budget_bytes (extension; also the attribute `table_budget_bytes`): keep the edge tables under that many
        bytes — tables that do not fit are rebuilt per step by the walk, as the reference's on-the-fly variant does
        for ALL of them (src/node2vec.py:34-53); the walks are the same."""
        self._check_popwalk()
        if self.p == 0 or self.q == 0:
            raise ZeroDivisionError("float division by zero")
        # the engine (graph on the device) is reused when p, q and the graph are unchanged, and its tables are released
        # BEFORE the new ones are allocated: the 58.5 GB of C3 then come back from the allocator's cache instead of a
        # second hipMalloc beside the old ones (1.5 s on this stack when the driver has to hand out recently freed
        # memory: tools/alloc_probe2.py)
        eng = self._engine
        if eng is None or eng.csr is not self._csr or eng.p != float(self.p) or eng.q != float(self.q):
            self._engine = eng = None
            eng = WalkEngine(self._csr, self.p, self.q, device=self.device)
        if budget_bytes is None:
            budget_bytes = getattr(self, "table_budget_bytes", None)
        eng.preprocess(budget_bytes=budget_bytes)
        self._engine = eng
        self.alias_nodes = _AliasNodes(self)
        self.alias_edges = _AliasEdges(self)
        return

    def get_alias_edge(self, src, dst):
        """src/node2vec.py:133-152: the (J, q) alias table of the step that arrives at `dst` from `src`, built by
        the table kernel for this one pair (no stored tables needed)."""
        self._check_popwalk()
        try:
            du, dv = (int(x) for x in self._csr.dense_of([src, dst]))
        except (KeyError, TypeError, ValueError):
            raise KeyError((src, dst))
        if self.p == 0 or self.q == 0:
            raise ZeroDivisionError("float division by zero")
        eng = self._graph_engine()
        e = eng.edge_index(du, dv)
        if e < 0:
            raise KeyError((src, dst))
        return eng.build_one_edge_table(e)

    def get_alias_edges_cur(self, src, dst):
        """src/node2vec.py:27-32 (popwalk "none")."""
        return self.get_alias_edge(src, dst)

    def get_alias_nodes_cur(self, cur):
        """src/node2vec.py:13-21 (popwalk "none"): the node table of `cur`, built on demand."""
        self._check_popwalk()
        try:
            dc = int(self._csr.dense_of([cur])[0])
        except (KeyError, TypeError, ValueError):
            raise KeyError(cur)
        return self._graph_engine().build_one_node_table(dc)

    def _starts(self, nodes):
        c = self._csr
        if not nodes:
            return c.start_order
        return c.dense_of(list(nodes))

    # src/node2vec.py:81-95
    def simulate_walks(self, num_walks, walk_length, nodes=None, verbose=False):
        '''
        Repeatedly simulate random walks from each node.
        '''
        if self._engine is None or not self._engine.ready:
            raise AttributeError("'Graph' object has no attribute 'alias_nodes'")  # as the reference would
        if verbose:
            for walk_iter in range(num_walks):
                print(str(walk_iter + 1), '/', str(num_walks))
        walks, lens = self._simulate(num_walks, walk_length, self._starts(nodes))
        return WalkCorpus(walks, lens, self._csr.labels)

    # src/node2vec.py:97-111 — identical output to simulate_walks for popwalk == "none".
    # Needs no preprocess_transition_probs(): the (prev, cur) table is rebuilt at every step
    # by the on-the-fly kernel (no sum-of-deg^2 storage).  If the tables already exist they
    # are used instead (same walks, faster) unless `self.force_on_the_fly` is set.
    def simulate_walks_on_the_fly(self, num_walks, walk_length, nodes=None, verbose=False):
        self._check_popwalk()
        if verbose:
            for walk_iter in range(num_walks):
                print(str(walk_iter + 1), '/', str(num_walks))
        otf = self._otf_engine()
        walks, lens = self._simulate(num_walks, walk_length, self._starts(nodes), otf=otf)
        return WalkCorpus(walks, lens, self._csr.labels)

    def _graph_engine(self):
        """The engine with the graph on the device (tables or not)."""
        if self._engine is None:
            self._engine = WalkEngine(self._csr, self.p, self.q, device=self.device)
        return self._engine

    def _otf_engine(self):
        """True if the on-the-fly kernel has to (or is asked to) run; makes sure an engine
        (graph on the device, no tables) exists."""
        if self._engine is not None and self._engine.ready and not getattr(self, "force_on_the_fly", False):
            return False
        if self._engine is None:
            self._engine = WalkEngine(self._csr, self.p, self.q, device=self.device)
        return True

    # src/node2vec.py:55-79
    def node2vec_walk(self, walk_length, start_node):
        if self._engine is None or not self._engine.ready:
            raise AttributeError("'Graph' object has no attribute 'alias_nodes'")
        walks, lens = self._simulate(1, walk_length, self._csr.dense_of([start_node]))
        return WalkCorpus(walks, lens, self._csr.labels)[0]

    def node2vec_walk_on_the_fly(self, walk_length, start_node):
        self._check_popwalk()
        otf = self._otf_engine()
        walks, lens = self._simulate(1, walk_length, self._csr.dense_of([start_node]), otf=otf)
        return WalkCorpus(walks, lens, self._csr.labels)[0]

    # ------------------------------------------------------------------ internals
    def _simulate(self, num_walks, walk_length, starts_host, otf=False):
        eng = self._engine
        self._walk = eng.walk_on_the_fly if otf else eng.walk
        d = eng.device
        L = max(int(walk_length), 1)  # walk = [start] even for walk_length <= 1 (:63-65)
        num_walks = int(num_walks)
        starts = torch.from_numpy(np.ascontiguousarray(starts_host, dtype=np.int32)).to(d)
        n = int(starts.numel())
        if n * num_walks == 0:
            return (torch.empty((0, L), dtype=torch.int32, device=d), torch.empty(0, dtype=torch.int32, device=d))
        if self.rng == "philox":
            return self._walk(starts, num_walks, L, rng="philox", seed=self.seed)
        if self.rng != "numpy":
            raise ValueError("rng must be 'numpy' or 'philox'")
        return self._simulate_numpy_stream(starts, n, num_walks, L)

    def simulate_walks_shard(self, num_walks, walk_length, rank, world):
        """The walks one of `world` GPUs owns: all rounds over the rank's
        contiguous block of start positions (as src/main_link.py:261-264 splits the start nodes).
        Row r*len(block)+i of the result is row r*N + block_begin + i of simulate_walks(): in
        Philox mode because the counter is the global walk index, in numpy mode because every
        rank jumps numpy's global MT19937 stream to the offsets its walks own — and leaves the
        global state where the full, single-process call would have left it."""
        if self._engine is None or not self._engine.ready:
            raise AttributeError("'Graph' object has no attribute 'alias_nodes'")
        eng = self._engine
        d = eng.device
        L = max(int(walk_length), 1)
        num_walks = int(num_walks)
        n = int(eng.start_order.numel())
        per = -(-n // int(world))
        b = min(int(rank) * per, n)
        cnt = min(b + per, n) - b
        if self.rng == "philox":
            walks, lens = eng.walk(eng.start_order, num_walks, L, rng="philox", seed=self.seed, pos_begin=b, pos_count=cnt)
            return WalkCorpus(walks, lens, self._csr.labels)
        if bool(self._csr.directed) and bool((eng.deg == 0).any().item()):
            # reachable sinks: a walk's place in the stream depends on the lengths of ALL earlier walks, whoever
            # owns them, so every rank resolves the whole chain (same work as the single-process call) and keeps
            # its block of rows — the union over ranks is the single-process result and the global state ends
            # where that call leaves it
            full = self.simulate_walks(num_walks, L)
            rows = (torch.arange(num_walks, device=d)[:, None] * n + torch.arange(b, b + cnt, device=d)[None, :]).reshape(-1)
            return WalkCorpus(full.walks[rows].contiguous(), full.lens[rows].contiguous(), self._csr.labels)
        active = (eng.deg[eng.start_order.long()] > 0).to(torch.int64) * (2 * (L - 1))
        prefix = torch.cumsum(active, 0) - active           # uniforms owned by earlier starts of a round
        per_round = int(active.sum().item())
        base = int(prefix[b].item()) if cnt else 0
        seg = int(active[b:b + cnt].sum().item())
        uoff = (prefix[b:b + cnt] - base).contiguous()
        walks = torch.empty((cnt * num_walks, L), dtype=torch.int32, device=d)
        lens = torch.empty(cnt * num_walks, dtype=torch.int32, device=d)
        st0 = np.random.get_state()
        self._walk = eng.walk
        tiled = self._tiled_uniforms_ok(L)
        for it in range(num_walks):
            if cnt == 0:
                break
            np.random.set_state(st0)
            _mt.advance_global_state((per_round * it + base))
            U = self._global_uniforms(max(seg, 2), d, L - 1 if tiled else None)
            eng.walk(eng.start_order, 1, L, rng="uniforms_tiled" if tiled else "uniforms", uniforms=U, walk_uoff=uoff,
                     pos_begin=b, pos_count=cnt, round_begin=it,
                     out=(walks[it * cnt:(it + 1) * cnt], lens[it * cnt:(it + 1) * cnt]))
        np.random.set_state(st0)
        _mt.advance_global_state(per_round * num_walks)
        return WalkCorpus(walks, lens, self._csr.labels)

    def _global_uniforms(self, n, device, tiled_pairs=None, out=None):
        """The next n doubles of numpy's global MT19937 stream as a device tensor: generated on
        the GPU by jump-ahead (default) or by numpy on the host (`host_rng = True`).  tiled_pairs = L-1: in the
        fat walk kernel's tiled layout (n2v_hip/mt19937.py)."""
        if getattr(self, "host_rng", False):
            assert not tiled_pairs and out is None
            return torch.from_numpy(np.random.random_sample(n)).to(device)
        return _mt.global_uniforms_device(n, device, tiled_pairs=tiled_pairs, out=out)

    def _tiled_uniforms_ok(self, L):
        """Tiled uniforms need the fat-table kernel, device-side generation and walks that cannot end early (the
        callers check the last)."""
        eng = self._engine
        return bool(L > 1 and not getattr(self, "host_rng", False) and not getattr(self, "linear_uniforms", False)
                    and getattr(self, "_walk", None) == eng.walk and eng.edge_fat is not None and not eng.partial)

    def _resolve_stream_offsets(self, starts, n, num_walks, L, active):
        """Directed graph with reachable sinks: a walk that ends early consumes fewer uniforms, so the position
        of walk w in numpy's stream depends on the lengths of all walks before it (src/node2vec.py:76-77 stops
        without drawing).  The chain is resolved on the device window by window: the window's first walk has an
        exact offset; the others are walked with offsets guessed from their last known lengths; after the scan
        of the new lengths every walk before the first one whose offset turned out different is final, and the
        next window starts there.  A pass costs one launch over at most 2^18 walks and one scalar read-back;
        a pass finalises 1 / P(a re-walked walk changes its length) walks on average (round 1 re-walked ALL
        walks per pass).  The uniforms are generated as the chain advances (a sliding piece of numpy's stream of
        at least 2^24 doubles), not for all walks at once."""
        eng = self._engine
        d = eng.device
        W = n * num_walks
        walks = torch.empty((W, L), dtype=torch.int32, device=d)
        lens = torch.empty(W, dtype=torch.int32, device=d)
        step = 2 * (L - 1)
        guess = (active.to(torch.int64) * step).repeat(num_walks)      # uniforms each walk is assumed to consume
        starts_all = starts.repeat(num_walks)
        done, exact_off, window = 0, 0, 4096
        self.stream_passes = 0
        U = torch.empty(0, dtype=torch.float64, device=d)     # stream positions [u_base, u_base + len(U))
        u_base = 0
        while done < W:
            hi = min(W, done + window)
            g = guess[done:hi]
            off = (torch.cumsum(g, 0) - g + exact_off).contiguous()
            need_end = exact_off + int(g.sum().item()) + step + 2      # a re-walked walk may run to full length
            if need_end > u_base + U.numel():
                keep = U[exact_off - u_base:] if exact_off - u_base < U.numel() else U[:0]
                have_end = max(u_base + U.numel(), exact_off)
                U = torch.cat([keep, self._global_uniforms(max(need_end - have_end, 1 << 24), d)])
                u_base = exact_off
            w_win, l_win = self._walk(starts_all[done:hi].contiguous(), 1, L, rng="uniforms", uniforms=U,
                                      walk_uoff=(off - u_base).contiguous())
            used = (l_win.to(torch.int64) - 1) * 2
            new_off = torch.cumsum(used, 0) - used + exact_off
            bad = new_off != off
            m = hi - done
            first = int(torch.nonzero(bad)[0].item()) if bool(bad.any().item()) else m
            # walks [0, first) were walked from their true offsets: final
            walks[done:done + first] = w_win[:first]
            lens[done:done + first] = l_win[:first]
            exact_off = int((new_off[first - 1] + used[first - 1]).item())
            guess[done + first:hi] = used[first:]           # best guess for the re-walk: the length just seen
            done += first
            self.stream_passes += 1
            window = max(1024, min(1 << 18, 4 * first if first < m else 2 * window))
        return walks, lens

    def _simulate_numpy_stream(self, starts, n, num_walks, L):
        """Parity mode.  Walk w = it*n + pos owns the uniforms the sequential reference
        loop would have handed it: offset 2 * sum_{w' < w} (len(w') - 1)."""
        eng = self._engine
        d = eng.device
        W = n * num_walks
        step = 2 * (L - 1)
        active = (eng.deg[starts.long()] > 0)
        per_round = active.to(torch.int64) * step
        per = int(per_round.sum().item())                   # uniforms one round consumes when no walk ends early
        if per == 0:
            return self._walk(starts, num_walks, L, rng="uniforms", uniforms=torch.zeros(2, dtype=torch.float64, device=d),
                              walk_uoff=torch.zeros(W, dtype=torch.int64, device=d))
        # a walk can end early only at a node without out-neighbours that is reachable,
        # i.e. never on an undirected graph (every visited node has the edge it came by)
        may_end_early = bool(self._csr.directed) and bool((eng.deg == 0).any().item())
        if not may_end_early:
            # every round consumes the same number of uniforms (start nodes without edges own none), so a walk's
            # offset is its offset inside the round + round * per: no per-walk array, nothing of size W is computed
            # here.  With fat tables the generator writes the uniforms in the walk kernel's tiled layout (64 walks'
            # segments regrouped step-major: one coalesced 1-KiB read per wavefront and step instead of 64 requests
            # 2(L-1) doubles apart).  The whole call is generated at once when it fits a quarter of the free memory
            # (one device-side jump-ahead per chunk is the fixed cost; generation and walk do not overlap on the
            # chip — a walk launch leaves no wave slots for a second stream — so more chunks only cost more jumps).
            uoff_round = None if per == n * step else (torch.cumsum(per_round, 0) - per_round).contiguous()
            walks = torch.empty((W, L), dtype=torch.int32, device=d)
            lens = torch.empty(W, dtype=torch.int32, device=d)
            host = getattr(self, "host_rng", False)
            tiled = self._tiled_uniforms_ok(L)
            rounds_per_chunk = getattr(self, "uniform_chunk_rounds", None)
            if rounds_per_chunk is None:
                free = torch.cuda.mem_get_info(d)[0]
                budget = (1 << 27) if host else max(1 << 27, min(1 << 32, free // 32))   # doubles per chunk (<= 32 GiB)
                rounds_per_chunk = budget // max(per, 1)
            rounds_per_chunk = int(max(1, min(num_walks, rounds_per_chunk)))
            for it in range(0, num_walks, rounds_per_chunk):
                k = min(rounds_per_chunk, num_walks - it)
                U = self._global_uniforms(per * k, d, L - 1 if tiled else None)
                self._walk(starts, k, L, rng="uniforms_tiled" if tiled else "uniforms", uniforms=U, walk_uoff=uoff_round,
                           uoff_round_stride=per, round_begin=it, out=(walks[it * n:(it + k) * n], lens[it * n:(it + k) * n]))
                del U
            return walks, lens
        state = np.random.get_state()
        walks, lens = self._resolve_stream_offsets(starts, n, num_walks, L, active)
        used = int(((lens.to(torch.int64) - 1) * 2).sum().item())
        np.random.set_state(state)
        _mt.advance_global_state(used)  # leave the global stream where the reference would
        return walks, lens
