"""Device-side walk engine: CSR + alias tables resident in HBM as torch tensors, every
arithmetic step done by the HIP kernels behind the C-ABI (include/n2v_hip.h).

torch is used for memory, streams and index plumbing (prefix sums, size-order sort);
the alias arithmetic and the walk are never done in torch and there is no CPU path.
"""
import time

import numpy as np
import torch

from . import _lib
from .csr import CsrGraph

SLOT_BYTES = 16  # sizeof(n2v_alias_slot)
FAT_BYTES = 32   # sizeof(n2v_fat_slot)


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise RuntimeError("n2v_hip: no GPU visible (torch.cuda.is_available() is False); "
                           "this engine has no CPU fallback")
    return torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())


class WalkEngine:
    """Owns the HBM-resident graph and tables for one (graph, p, q)."""

    def __init__(self, csr: CsrGraph, p, q, device=None):
        self.lib = _lib.load()
        self.device = _require_gpu(device)
        self.csr = csr
        self.p = float(p)
        self.q = float(q)
        d = self.device
        self.row_ptr = torch.from_numpy(csr.row_ptr).to(d)
        self.col = torch.from_numpy(csr.col).to(d)
        # all-ones weights ARE the unweighted graph (`G[u][v]['weight'] = 1` of src/main.py:48-49): the kernels then use 1.0
        # without loading it — same bits — and the on-the-fly walk may count instead of summing (n2v_wave_table.h)
        unit = csr.w is not None and bool(np.all(csr.w == 1.0))
        self.w = None if (csr.w is None or unit) else torch.from_numpy(csr.w).to(d)
        self.start_order = torch.from_numpy(csr.start_order).to(d)
        self.deg = (self.row_ptr[1:] - self.row_ptr[:-1])
        hdeg = np.diff(csr.row_ptr)
        self.max_degree = int(hdeg.max()) if csr.n_nodes else 0     # host arithmetic: see preprocess() on why no device
        # reduction / scan / sort may run before the big table allocation
        if csr.nnz == 0:
            self.total_edge_slots = 0
        elif csr.directed:
            self.total_edge_slots = int((np.bincount(csr.col, minlength=csr.n_nodes).astype(np.int64) * hdeg).sum())
        else:
            self.total_edge_slots = int((hdeg.astype(np.int64) ** 2).sum())   # sum over entries of deg(col[e])
        self.node_slots = None
        self.edge_slots = None
        self.edge_off = None
        self.recs = None
        self.node_fat = self.edge_fat = None
        self.first_order = False
        self.partial = False          # tables under a memory budget: only entries with deg(dst) <= stored_degree_cut

    # ------------------------------------------------------------------ tables
    def _stream(self):
        return _lib.stream_ptr(self.device)

    def preprocess(self, first_order_shortcut=True, fat="auto", builder="wave", budget_bytes=None):
        """preprocess_transition_probs (src/node2vec.py:176-204) on device.

        With p == q == 1 every (src,dst) table is bit-identical to dst's node table
        (w/1 == w), so the Σdeg² edge tables are not materialised unless
        ``first_order_shortcut=False``.

        fat: "auto" (fat 32-B slots if they fit, else thin 16-B slots), True, False, or "both" (tests / probes:
        thin and fat side by side).  The edge tables exist ONCE: the wave-per-table kernel writes the chosen
        layout directly; (J, q) of a stored fat table are recovered by `thin_view` for the dict-like views.
        builder: "wave" (n2v_build_edge_tables_wave) or "lane" (round 1's one-lane-per-table kernel, kept as a
        cross-check of the same bits).
        budget_bytes: tables under a memory budget — the middle path between stored tables and the reference's
        rebuild-every-step fallback (src/node2vec.py:34-53, src/settings.py:18).  If the fat edge tables exceed it,
        only the tables of entries (src -> dst) with deg(dst) <= D are stored (D = the largest cut that fits: every
        table is visited equally often per byte, but a rebuilt table costs a fixed latency on top of its slots, so the
        small ones are the ones to keep); the walk (n2v_walk_hybrid) rebuilds the others per step.  Same walks."""
        csr, d = self.csr, self.device
        N, nnz = csr.n_nodes, csr.nnz
        self.timings = {}
        tick = self._phase_timer()
        self.node_slots = self.edge_slots = self.recs = self.node_fat = self.edge_fat = None
        with torch.cuda.device(d):
            self.first_order = bool(first_order_shortcut and self.p == 1.0 and self.q == 1.0)
            total = nnz if self.first_order else self.total_edge_slots
            self.partial, self.stored_degree_cut, stored_entries = False, None, None
            free, _ = torch.cuda.mem_get_info(d)
            free += torch.cuda.memory_reserved(d) - torch.cuda.memory_allocated(d)    # the allocator's cache is ours to reuse
            if (budget_bytes is None and fat == "auto" and not self.first_order
                    and total * SLOT_BYTES > free - (1 << 30)):
                # not even the thin tables fit: keep the tables that do and let the walk rebuild the others per step
                # (the reference's answer is to rebuild ALL of them, src/settings.py:18) instead of giving up
                budget_bytes = max(0, free - (8 << 30))
            if budget_bytes is not None and not self.first_order and total * FAT_BYTES > int(budget_bytes):
                # host arithmetic (no device reduction before the big allocation)
                from .csr import degree_cut_for_budget
                self.stored_degree_cut, total = degree_cut_for_budget(csr, budget_bytes, FAT_BYTES)
                self.partial, fat, builder = True, True, "wave"
            self.total_slots = total
            if fat == "auto":
                fat = (nnz + (0 if self.first_order else total)) * FAT_BYTES < free - (8 << 30)
            want_thin = fat in (False, "both")
            want_fat = fat in (True, "both")
            need = 0 if self.first_order else total * ((SLOT_BYTES if want_thin else 0) + (FAT_BYTES if want_fat else 0))
            if need > free - (1 << 30):
                raise MemoryError("edge alias tables need %.1f GB (sum of deg^2 = %d slots), %.1f GB free"
                                  % (need / 1e9, total, free / 1e9))
            # The big buffers FIRST, before any device scan / sort / reduction of this call: on this stack a fresh
            # multi-GB allocation that follows such a kernel takes seconds (tools/alloc_probe.py: 58 GB in 0.000 s
            # as the first thing, 3.06 s after a torch.argsort of 2e7 keys; round 1's preprocess spent 1.5-2.4 s of
            # its 2.76 s there).  The slot count comes from host arithmetic on the CSR for the same reason.
            t_alloc = time.perf_counter()
            if not self.first_order:
                if want_thin:
                    self.edge_slots = torch.empty((max(total, 1), 2), dtype=torch.int64, device=d)
                if want_fat:
                    self.edge_fat = torch.empty((max(total, 1), 4), dtype=torch.int64, device=d)
            # host time of the table allocation (hipMalloc blocks the host when the driver has to hand out memory another
            # allocation has just released: 65-80 ms per GiB, tools/alloc_probe2.py; ~0 from the allocator's cache or
            # from memory that was never used) — reported separately from the kernels by bench.py
            self.alloc_seconds = time.perf_counter() - t_alloc
            tick("alloc")
            status = torch.zeros(1, dtype=torch.int32, device=d)
            self.node_slots = torch.zeros((max(nnz, 1), 2), dtype=torch.int64, device=d)
            _lib.check(self.lib.n2v_build_node_tables(
                N, _lib.ptr(self.row_ptr), _lib.ptr(self.col), _lib.ptr(self.w), _lib.ptr(self.node_slots),
                _lib.ptr(status), self._stream()))
            tick("node_tables")
            rec_off = None
            if self.first_order:
                self.edge_off = None
            else:
                kdst = self.deg[self.col.long()]
                if self.partial:
                    stored = kdst <= self.stored_degree_cut
                    kdst = kdst * stored
                    stored_entries = torch.nonzero(stored).flatten().to(torch.int32).contiguous()
                self.edge_off = torch.zeros(nnz + 1, dtype=torch.int64, device=d)
                torch.cumsum(kdst, 0, out=self.edge_off[1:])
                assert int(self.edge_off[-1].item()) == total, "host and device slot counts differ"
                rec_off = self.edge_off
                if self.partial:      # a negative offset marks an entry without a stored table (N2V_NO_TABLE)
                    rec_off = torch.where(stored, self.edge_off[:-1], torch.full_like(self.edge_off[:-1], -1)).contiguous()
                    self.stored_mask = stored
            # walk records first: they depend on the table OFFSETS only, and the fat slots embed them
            self.recs = torch.empty((max(nnz, 1), 4), dtype=torch.int32, device=d)
            _lib.check(self.lib.n2v_build_edge_recs(
                N, nnz, _lib.ptr(self.row_ptr), _lib.ptr(self.col), _lib.ptr(rec_off), 0,
                self.max_degree, total, _lib.ptr(self.recs), self._stream()))
            tick("offsets_recs")
            if self.first_order:
                self.edge_slots = self.node_slots
            else:
                # nothing is sorted here: the wave kernel takes its tables from a shared counter instead of a
                # size-ordered list
                src_of = torch.repeat_interleave(torch.arange(N, dtype=torch.int32, device=d), self.deg)
                sym = 0 if csr.directed else 1
                order = None
                if builder == "lane":        # one lane per table: lanes of a wave should get tables of similar size
                    order = torch.argsort(kdst, descending=True).to(torch.int32)
                work = torch.zeros(2, dtype=torch.int64, device=d)
                # per-wave stacks of the tables that do not fit the wave's LDS slots (C3: 0.8 GB for max degree 16 614)
                built_max = self.stored_degree_cut if self.partial else self.max_degree
                n_build = int(stored_entries.numel()) if self.partial else nnz
                sbytes = int(self.lib.n2v_edge_tables_wave_scratch_bytes(built_max)) if builder == "wave" else 0
                scratch = torch.empty(max(sbytes, 64) // 8, dtype=torch.int64, device=d)
                tick("src_of")
                if want_thin:
                    if builder == "lane":
                        _lib.check(self.lib.n2v_build_edge_tables(
                            N, _lib.ptr(self.row_ptr), _lib.ptr(self.col), _lib.ptr(self.w), _lib.ptr(src_of),
                            self.p, self.q, sym, _lib.ptr(self.edge_off), _lib.ptr(order), 0, nnz,
                            _lib.ptr(self.edge_slots), _lib.ptr(status), self._stream()))
                    else:
                        _lib.check(self.lib.n2v_build_edge_tables_wave(
                            N, _lib.ptr(self.row_ptr), _lib.ptr(self.col), _lib.ptr(self.w), _lib.ptr(src_of),
                            self.p, self.q, sym, _lib.ptr(self.edge_off), _lib.ptr(stored_entries), 0, n_build, None,
                            _lib.ptr(self.edge_slots), None, _lib.ptr(status), work[0:].data_ptr(), built_max,
                            _lib.ptr(scratch), sbytes, self._stream()))
                    tick("edge_tables_thin")
                if want_fat:
                    if builder == "lane":
                        assert want_thin, "the lane builder writes thin tables; fat ones are expanded from them"
                        _lib.check(self.lib.n2v_build_fat_slots(
                            nnz, _lib.ptr(self.edge_off), _lib.ptr(self.col), _lib.ptr(self.row_ptr),
                            _lib.ptr(self.edge_slots), _lib.ptr(self.recs), _lib.ptr(self.edge_fat), self._stream()))
                    else:
                        _lib.check(self.lib.n2v_build_edge_tables_wave(
                            N, _lib.ptr(self.row_ptr), _lib.ptr(self.col), _lib.ptr(self.w), _lib.ptr(src_of),
                            self.p, self.q, sym, _lib.ptr(self.edge_off), _lib.ptr(stored_entries), 0, n_build,
                            _lib.ptr(self.recs), None, _lib.ptr(self.edge_fat), _lib.ptr(status), work[1:].data_ptr(),
                            built_max, _lib.ptr(scratch), sbytes, self._stream()))
                    tick("edge_tables_fat")
                del order, src_of, kdst, scratch
            if want_fat and nnz > 0:
                self.node_fat = torch.empty((max(nnz, 1), 4), dtype=torch.int64, device=d)
                _lib.check(self.lib.n2v_build_fat_slots(
                    N, _lib.ptr(self.row_ptr), None, _lib.ptr(self.row_ptr), _lib.ptr(self.node_slots),
                    _lib.ptr(self.recs), _lib.ptr(self.node_fat), self._stream()))
                if self.first_order:
                    self.edge_fat = self.node_fat
            st = int(status.item())
            tick("node_fat")
        if st & _lib.N2V_STATUS_ZERO_NORM:
            self.node_slots = self.edge_slots = self.recs = self.node_fat = self.edge_fat = None
            raise ZeroDivisionError("float division by zero")

    def _phase_timer(self):
        """Wall time per preprocess phase into self.timings when N2V_TIMING is set (syncs the device after each
        phase, so only for diagnosis: tools/preprocess_probe.py)."""
        import os
        import time
        if not os.environ.get("N2V_TIMING"):
            return lambda name: None
        t = [time.perf_counter()]

        def tick(name):
            torch.cuda.synchronize(self.device)
            now = time.perf_counter()
            self.timings[name] = self.timings.get(name, 0.0) + now - t[0]
            t[0] = now
        return tick

    def build_fat(self):
        """Expand stored THIN tables into 32-byte fat slots (probes; preprocess() writes fat tables directly)."""
        if self.edge_slots is None:
            raise RuntimeError("no thin edge tables to expand (preprocess(fat=False) first)")
        csr, d = self.csr, self.device
        N, nnz = csr.n_nodes, csr.nnz
        with torch.cuda.device(d):
            self.node_fat = torch.empty((max(nnz, 1), 4), dtype=torch.int64, device=d)
            _lib.check(self.lib.n2v_build_fat_slots(
                N, _lib.ptr(self.row_ptr), None, _lib.ptr(self.row_ptr), _lib.ptr(self.node_slots),
                _lib.ptr(self.recs), _lib.ptr(self.node_fat), self._stream()))
            if self.first_order:
                self.edge_fat = self.node_fat
            else:
                self.edge_fat = torch.empty((max(self.total_slots, 1), 4), dtype=torch.int64, device=d)
                _lib.check(self.lib.n2v_build_fat_slots(
                    nnz, _lib.ptr(self.edge_off), _lib.ptr(self.col), _lib.ptr(self.row_ptr),
                    _lib.ptr(self.edge_slots), _lib.ptr(self.recs), _lib.ptr(self.edge_fat), self._stream()))

    @property
    def ready(self):
        return self.recs is not None

    # raw views used by the dict-like alias_nodes / alias_edges and by the tests
    @staticmethod
    def slots_q(slots):
        return slots.view(torch.float64)[:, 0]

    @staticmethod
    def slots_J(slots):
        return slots.view(torch.int32)[:, 2]

    def node_table(self, dense):
        b, e = int(self.csr.row_ptr[dense]), int(self.csr.row_ptr[dense + 1])
        s = self.node_slots[b:e]
        return (self.slots_J(s).cpu().numpy().astype(np.int64), self.slots_q(s).cpu().numpy())

    def edge_index(self, du, dv):
        """CSR entry of (du -> dv) or -1."""
        b, e = int(self.csr.row_ptr[du]), int(self.csr.row_ptr[du + 1])
        k = int(np.searchsorted(self.csr.col[b:e], dv))
        if k < e - b and self.csr.col[b + k] == dv:
            return b + k
        return -1

    def edge_table(self, e):
        if self.first_order:
            return self.node_table(int(self.csr.col[e]))
        if self.partial and not bool(self.stored_mask[e].item()):
            return self.build_one_edge_table(e)         # not stored under the memory budget: built on demand
        off = self.edge_off[e:e + 2].cpu().tolist()
        J, q = self.thin_view(torch.arange(off[0], off[1], device=self.device), table_node=int(self.csr.col[e]))
        return (J.cpu().numpy().astype(np.int64), q.cpu().numpy())

    def all_edge_tables(self):
        """(J int64, q float64) numpy arrays over all edge-table slots (tests / small graphs)."""
        J, q = self.thin_view(torch.arange(self.total_slots, device=self.device))
        return J.cpu().numpy(), q.cpu().numpy()

    def thin_view(self, slot_idx, table_node=None):
        """(J int64, q float64) of the given edge-table slots (device index tensor), whichever layout is stored.
        From fat slots J is recovered as the position of the alias record's node in the row the table draws
        from (rows hold distinct ids): `table_node` = that node when all slots belong to one table, else it is
        looked up per slot."""
        slot_idx = slot_idx.to(device=self.device, dtype=torch.int64)
        if self.edge_slots is not None:
            s = self.edge_slots[slot_idx]
            return self.slots_J(s).long(), self.slots_q(s)
        f = self.edge_fat[slot_idx]
        q = f.view(torch.float64)[:, 0]
        alias_dst = f.view(torch.int32)[:, 7].long()
        if table_node is None:
            if self.first_order:
                node = torch.searchsorted(self.row_ptr, slot_idx, right=True) - 1
            else:
                node = self.col[torch.searchsorted(self.edge_off, slot_idx, right=True) - 1].long()
        else:
            node = torch.full_like(slot_idx, int(table_node))
        # position of (node, alias_dst) in the CSR = global search over the keys row * N + col (ascending)
        n = self.csr.n_nodes
        keys = getattr(self, "_csr_keys", None)
        if keys is None:
            src_of = torch.repeat_interleave(torch.arange(n, device=self.device), self.deg)
            keys = self._csr_keys = src_of * n + self.col.long()
        e2 = torch.searchsorted(keys, node * n + alias_dst)
        return e2 - self.row_ptr[node], q

    def build_one_node_table(self, dense):
        """get_alias_nodes_cur (src/node2vec.py:13-21, popwalk "none"): the table of one node, built on demand."""
        b, e = int(self.csr.row_ptr[dense]), int(self.csr.row_ptr[dense + 1])
        d = self.device
        with torch.cuda.device(d):
            slots = torch.zeros((max(e - b, 1), 2), dtype=torch.int64, device=d)
            status = torch.zeros(1, dtype=torch.int32, device=d)
            rp = torch.tensor([0, e - b], dtype=torch.int64, device=d)
            w = None if self.w is None else self.w[b:e].contiguous()
            _lib.check(self.lib.n2v_build_node_tables(1, _lib.ptr(rp), self.col[b:].data_ptr() if e > b else None,
                                                      _lib.ptr(w), _lib.ptr(slots), _lib.ptr(status), self._stream()))
            if int(status.item()) & _lib.N2V_STATUS_ZERO_NORM:
                raise ZeroDivisionError("float division by zero")
        s = slots[: e - b]
        return (self.slots_J(s).cpu().numpy().astype(np.int64), self.slots_q(s).cpu().numpy())

    def build_one_edge_table(self, e):
        """get_alias_edge (src/node2vec.py:133-152) for one CSR entry e = (src -> dst), built on demand by the
        same kernel that fills the stored tables (one-table launch: the kernel reads src_of[e] and edge_off[e]
        only, so one-element arrays are passed with their base shifted by -e)."""
        dst = int(self.csr.col[e])
        K = int(self.csr.row_ptr[dst + 1] - self.csr.row_ptr[dst])
        src = int(np.searchsorted(self.csr.row_ptr, e, side="right") - 1)
        d = self.device
        with torch.cuda.device(d):
            slots = torch.zeros((max(K, 1), 2), dtype=torch.int64, device=d)
            status = torch.zeros(1, dtype=torch.int32, device=d)
            off = torch.zeros(1, dtype=torch.int64, device=d)
            src_of = torch.tensor([src], dtype=torch.int32, device=d)
            order = torch.tensor([e], dtype=torch.int64, device=d).to(torch.int32)
            _lib.check(self.lib.n2v_build_edge_tables(
                self.csr.n_nodes, _lib.ptr(self.row_ptr), _lib.ptr(self.col), _lib.ptr(self.w), src_of.data_ptr() - 4 * e,
                self.p, self.q, 0 if self.csr.directed else 1, off.data_ptr() - 8 * e, _lib.ptr(order), 0, 1,
                _lib.ptr(slots), _lib.ptr(status), self._stream()))
            if int(status.item()) & _lib.N2V_STATUS_ZERO_NORM:
                raise ZeroDivisionError("float division by zero")
        s = slots[:K]
        return (self.slots_J(s).cpu().numpy().astype(np.int64), self.slots_q(s).cpu().numpy())

    # ------------------------------------------------------------------ walks
    def walk(self, starts, num_rounds, walk_length, rng="philox", seed=0, uniforms=None, walk_uoff=None,
             pos_begin=0, pos_count=None, round_begin=0, out=None, layout=None, uoff_round_stride=0):
        """Launch the walk kernel.  ``starts``: int32 device tensor of dense ids (the
        start order).  Returns (walks int32[n_local, L], lens int32[n_local]) on device.
        uoff_round_stride > 0: walk_uoff holds one round's offsets (pos_count entries) and every round consumes
        that many uniforms (include/n2v_hip.h, n2v_walk_fat); expanded here for the thin-table kernel."""
        if not self.ready:
            raise RuntimeError("preprocess() first")
        if self.partial:          # stored tables for part of the entries: the hybrid kernel rebuilds the others per step
            if rng == "uniforms_tiled":
                raise ValueError("the tiled uniform layout is read by the fat-table walk kernel only")
            return self.walk_on_the_fly(starts, num_rounds, walk_length, rng=rng, seed=seed, uniforms=uniforms,
                                        walk_uoff=walk_uoff, pos_begin=pos_begin, pos_count=pos_count,
                                        round_begin=round_begin, out=out, uoff_round_stride=uoff_round_stride, hybrid=True)
        d = self.device
        L = int(walk_length)
        if L < 1:
            raise ValueError("walk_length must be >= 1")
        n_starts = int(starts.numel())
        if pos_count is None:
            pos_count = n_starts - pos_begin
        n_local = pos_count * num_rounds
        with torch.cuda.device(d):
            if out is None:
                walks = torch.empty((n_local, L), dtype=torch.int32, device=d)
                lens = torch.empty(n_local, dtype=torch.int32, device=d)
            else:
                walks, lens = out
                assert walks.shape == (n_local, L) and walks.dtype == torch.int32 and walks.is_contiguous()
            mode = {"uniforms": _lib.RNG_UNIFORMS, "uniforms_tiled": _lib.RNG_UNIFORMS_TILED}.get(rng, _lib.RNG_PHILOX)
            use_fat = self.edge_fat is not None if layout is None else (layout == "fat")
            if mode == _lib.RNG_UNIFORMS_TILED and not use_fat:
                raise ValueError("the tiled uniform layout is read by the fat-table walk kernel only")
            if not use_fat and self.edge_slots is None:
                raise RuntimeError("thin tables were not built (preprocess(fat=False) or fat='both')")
            if use_fat:
                if self.edge_fat is None:
                    raise RuntimeError("fat tables were not built")
                _lib.check(self.lib.n2v_walk_fat(
                    _lib.ptr(self.row_ptr), _lib.ptr(self.node_fat), _lib.ptr(self.edge_fat), _lib.ptr(starts),
                    n_starts, pos_begin, pos_count, round_begin, num_rounds, L, mode, _lib.ptr(uniforms),
                    _lib.ptr(walk_uoff), int(uoff_round_stride) if walk_uoff is not None else 0,
                    int(seed) & (2**64 - 1), _lib.ptr(walks), _lib.ptr(lens), self._stream()))
            else:
                if uoff_round_stride and walk_uoff is not None:
                    walk_uoff = expand_round_offsets(walk_uoff, num_rounds, uoff_round_stride)
                _lib.check(self.lib.n2v_walk(
                    _lib.ptr(self.row_ptr), _lib.ptr(self.node_slots), _lib.ptr(self.recs),
                    _lib.ptr(self.edge_slots), _lib.ptr(starts), n_starts, pos_begin, pos_count, round_begin,
                    num_rounds, L, mode, _lib.ptr(uniforms), _lib.ptr(walk_uoff), int(seed) & (2**64 - 1),
                    _lib.ptr(walks), _lib.ptr(lens), self._stream()))
        return walks, lens

    def walk_on_the_fly(self, starts, num_rounds, walk_length, rng="philox", seed=0, uniforms=None, walk_uoff=None,
                         pos_begin=0, pos_count=None, round_begin=0, out=None, uoff_round_stride=0, hybrid=False):
        """Launch the on-the-fly walk kernel (no stored edge tables; src/node2vec.py:97-111).
        Same arguments and results as WalkEngine.walk.  hybrid: n2v_walk_hybrid — steps through entries whose table is
        stored (preprocess(budget_bytes=...)) read it, the others rebuild theirs."""
        d = self.device
        L = int(walk_length)
        if L < 1:
            raise ValueError("walk_length must be >= 1")
        if self.p == 0 or self.q == 0:
            raise ZeroDivisionError("float division by zero")
        n_starts = int(starts.numel())
        if pos_count is None:
            pos_count = n_starts - pos_begin
        n_local = pos_count * num_rounds
        with torch.cuda.device(d):
            if out is None:
                walks = torch.empty((n_local, L), dtype=torch.int32, device=d)
                lens = torch.empty(n_local, dtype=torch.int32, device=d)
            else:
                walks, lens = out
            scratch = getattr(self, "_otf_scratch", None)
            if self.max_degree > int(self.lib.n2v_walk_otf_lds_slots()):
                # one scratch row per resident wavefront; grown when a later call launches more waves than the
                # call that first allocated it (a single node2vec_walk_on_the_fly must not pin later launches
                # to one workgroup)
                n_waves = min(int(self.lib.n2v_walk_otf_max_waves()), max(4, (n_local + 3) // 4 * 4))
                if scratch is None or scratch.shape[0] < n_waves * self.max_degree:
                    scratch = self._otf_scratch = torch.empty((n_waves * self.max_degree, 2), dtype=torch.int64,
                                                              device=d)
            status = torch.zeros(1, dtype=torch.int32, device=d)
            mode = _lib.RNG_UNIFORMS if rng == "uniforms" else _lib.RNG_PHILOX
            if uoff_round_stride and walk_uoff is not None:
                walk_uoff = expand_round_offsets(walk_uoff, num_rounds, uoff_round_stride)
            tail = (_lib.ptr(starts), n_starts, pos_begin, pos_count, round_begin, num_rounds, L, mode,
                    _lib.ptr(uniforms), _lib.ptr(walk_uoff), int(seed) & (2**64 - 1), _lib.ptr(scratch),
                    0 if scratch is None else int(scratch.shape[0]), _lib.ptr(walks), _lib.ptr(lens), _lib.ptr(status),
                    self._stream())
            head = (_lib.ptr(self.row_ptr), _lib.ptr(self.col), _lib.ptr(self.w), self.p, self.q,
                    0 if self.csr.directed else 1, self.max_degree)
            if hybrid:
                _lib.check(self.lib.n2v_walk_hybrid(*head, _lib.ptr(self.node_fat), _lib.ptr(self.edge_fat),
                                                    _lib.ptr(self.recs), *tail))
            else:
                _lib.check(self.lib.n2v_walk_on_the_fly(*head, *tail))
            if int(status.item()) & _lib.N2V_STATUS_ZERO_NORM:
                raise ZeroDivisionError("float division by zero")
        return walks, lens


def expand_round_offsets(uoff_round, num_rounds, stride):
    """One round's offsets + a per-round stride -> the per-walk offset array of the kernels that take no stride."""
    r = torch.arange(num_rounds, dtype=torch.int64, device=uoff_round.device) * int(stride)
    return (r[:, None] + uoff_round[None, :]).reshape(-1).contiguous()


def alias_setup_device(prob_tables, device=None):
    """alias_setup (src/node2vec.py:240-269) for one table (a flat list of probabilities)
    or a list of tables; returns (J int64, q float64) per table, computed on the GPU."""
    single = len(prob_tables) == 0 or not hasattr(prob_tables[0], "__len__")
    tables = [prob_tables] if single else list(prob_tables)
    d = _require_gpu(device)
    lib = _lib.load()
    sizes = np.array([len(t) for t in tables], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    total = int(off[-1])
    host = np.zeros((max(total, 1), 2), dtype=np.int64)
    if total:
        host.view(np.float64)[:total, 0] = np.concatenate([np.asarray(t, dtype=np.float64) for t in tables if len(t)])
    with torch.cuda.device(d):
        slots = torch.from_numpy(host).to(d)
        off_d = torch.from_numpy(off).to(d)
        _lib.check(lib.n2v_alias_setup_tables(len(tables), _lib.ptr(off_d), _lib.ptr(slots), _lib.stream_ptr(d)))
        out = slots.cpu().numpy()
    q = out.view(np.float64)[:, 0]
    J = out.view(np.int32)[:, 2]
    res = [(J[off[i]:off[i + 1]].astype(np.int64), q[off[i]:off[i + 1]].copy()) for i in range(len(tables))]
    return res[0] if single else res
