"""CPU restatement of the reference's BiNE path — TEST INFRASTRUCTURE ONLY (never imported by the product).

Parity status: **unpinned at the bit level, by construction of the reference.**  Its BiNE files do not
import on Python 3 (`from collections import Iterable` src/bine_graph.py:14; `keys.sort()` on dict views
src/bine_graph_utils.py:53-54; modules `graph`, `lsh`, `data_utils`, `graph_utils` that do not exist under
those names src/bine_train.py:9-10, src/bine_graph_utils.py:4,8; datasketch absent), hold no fixtures, and
draw from an unseeded `random.Random()` default argument (src/bine_graph.py:169,224,334).  Two layers:

 (L) *literal* restatements of the reference text (materialised projection rows, `random`-module draws,
     networkx-1.11 `hits` power iteration, numpy skip_gram / KL_divergence / train loop) — what the reference
     computes, in distribution;
 (P) *Philox* restatements of the device algorithm (same counters, same arithmetic) — what the HIP kernels
     must reproduce bit for bit (walks, pools, sampled occurrences, contexts, negatives) or to fp64 rounding
     (training).
tests/ check (P) == HIP exactly and (P) ~ (L) statistically (chi-square on next-vertex and length
distributions), so the chain reference-text -> (L) -> (P) -> HIP is closed without the reference running.
"""
import math

import numpy as np

from oracle.n2v_oracle import philox4x32_10

LN10 = math.log(10, math.e)  # src/bine_train.py:300
MAX_TRIALS = 1 << 16
POOL_TRIALS = 16


def _u53(a, b):
    return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0


def _philox(seed, c0, c1, c2, c3):
    return philox4x32_10((c0 & 0xFFFFFFFF, c1 & 0xFFFFFFFF, c2 & 0xFFFFFFFF, c3 & 0xFFFFFFFF),
                         (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


# ------------------------------------------------------------------------------- (L) HITS, networkx 1.11
def hits_nx111(row_ptr, col, w, max_iter=100, tol=1.0e-8):
    """networkx 1.11 `hits(G)` (third-party, pinned by requirements.txt:networkx==1.11; restated from its
    published source): h = 1/n; repeat {a = M^T h_last; h = M a; h /= max h; a /= max a;
    err = sum |h - h_last|} until err < tol; raises after max_iter.  Returns the un-normalised a
    (the final division by sum(a) cancels in the per-side min-max scaling of
    src/bine_graph_utils.py:76-86)."""
    n = len(row_ptr) - 1
    rows = np.repeat(np.arange(n), np.diff(row_ptr))
    h = np.full(n, 1.0 / n)
    for it in range(max_iter):
        hlast = h
        a = np.zeros(n)
        np.add.at(a, col, hlast[rows] * w)          # a[nbr] += hlast[n] * w
        h = np.zeros(n)
        np.add.at(h, rows, a[col] * w)              # h[n] += a[nbr] * w
        h = h * (1.0 / h.max())
        a = a * (1.0 / a.max())
        err = np.abs(h - hlast).sum()
        if err < tol:
            return a, it + 1
    raise RuntimeError("HITS: power iteration failed to converge in %d iterations" % max_iter)


def walk_counts(a, lo, hi, maxT, minT):
    """src/bine_graph_utils.py:62-86 (per-side min-max) + src/bine_graph.py:358 (ceil, floor at minT)."""
    seg = a[lo:hi]
    mx, mn = max(0.0, float(seg.max())), min(100000.0, float(seg.min()))
    span = mx - mn
    auth = (seg - mn) / span if span != 0 else np.zeros(hi - lo)
    return np.maximum(np.ceil(maxT * auth).astype(np.int64), minT), auth


# ------------------------------------------------------------------------------- (L) literal walk
def projection_rows(row_ptr, col, v):
    """Row v of the reference's `matrix` (src/bine_graph_utils.py:223-241): the distinct column indices of
    row v of A*A^T, ascending, v itself included when v has any neighbour."""
    out = set()
    for m in col[row_ptr[v]:row_ptr[v + 1]]:
        out.update(int(x) for x in col[row_ptr[m]:row_ptr[m + 1]])
    return sorted(out)


def literal_walk(matrix_rows, start, percentage, rand, max_tokens=None):
    """random_walk_restart_for_large_bipartite_graph (src/bine_graph.py:263-308) with alpha = 0.
    `matrix_rows(v)` -> list as above; len(G[cur]) is that list without cur (self loops removed,
    src/bine_graph.py:94,128-131).  max_tokens is a harness guard only (the reference has none)."""
    path = [start]
    while len(path) < 1 or rand.random() > percentage:
        if max_tokens is not None and len(path) >= max_tokens:
            break
        cur = path[-1]
        neighbors = matrix_rows(cur)
        if len([x for x in neighbors if x != cur]) > 0:
            if rand.random() >= 0:                      # alpha = 0: always true, consumes a draw
                add_node = rand.choice(neighbors)
                while add_node == cur:
                    add_node = rand.choice(neighbors)
                path.append(add_node)
        else:
            break
    return path


# ------------------------------------------------------------------------------- (P) device walk
def two_hop_prefix(row_ptr, col):
    deg = np.diff(row_ptr)
    return np.concatenate([[0], np.cumsum(deg[col])]).astype(np.int64)


def walk_length(row_ptr, cum2, node, gw, percentage, max_len, seed):
    rb, re = int(row_ptr[node]), int(row_ptr[node + 1])
    paths = int(cum2[re] - cum2[rb])
    length = 1
    if paths - (re - rb) > 0:
        t = 0
        while length < max_len:
            r = _philox(seed, gw, gw >> 32, t, 0)
            if _u53(r[0], r[1]) > percentage:
                length += 1
            else:
                break
            t += 1
    return length


def _intersect(col, a_lo, a_n, b_lo, b_n):
    return len(set(col[a_lo:a_lo + a_n].tolist()) & set(col[b_lo:b_lo + b_n].tolist()))


def device_walk(row_ptr, col, cum2, node, gw, length, seed, stats=None):
    cur = int(node)
    out = [cur]
    for t in range(length - 1):
        rb, re = int(row_ptr[cur]), int(row_ptr[cur + 1])
        base = int(cum2[rb])
        paths = int(cum2[re]) - base
        nxt = cur
        for trial in range(MAX_TRIALS):
            r = _philox(seed, gw, gw >> 32, t, 1 + trial)
            pick = min(int(math.floor(_u53(r[0], r[1]) * float(paths))), paths - 1)
            e = rb + int(np.searchsorted(cum2[rb + 1:re + 1] - base, pick, side="right"))
            mid = int(col[e])
            w = int(col[int(row_ptr[mid]) + pick - (int(cum2[e]) - base)])
            if stats is not None:
                stats["trials"] = stats.get("trials", 0) + 1
            if w == cur:
                continue
            nxt = w
            wb = int(row_ptr[w])
            before = set(col[rb:e].tolist())              # cur's neighbours below mid
            if not (before & set(col[wb:int(row_ptr[w + 1])].tolist())):
                break                                     # mid is the first common neighbour: keep the path
        cur = nxt
        out.append(cur)
    return out


def neg_pool(row_ptr, col, side_lo, side_hi, v, pool_size, max_jaccard, seed):
    vb, vn = int(row_ptr[v]), int(row_ptr[v + 1] - row_ptr[v])
    side_n = side_hi - side_lo
    out = []
    for s in range(pool_size):
        c = v
        for trial in range(POOL_TRIALS + 1):
            r = _philox(seed, v, s, trial, 0)
            k = min(int(math.floor(_u53(r[0], r[1]) * float(side_n))), side_n - 1)
            c = side_lo + k
            if c == v:
                continue
            if trial == POOL_TRIALS:
                break
            cb, cn = int(row_ptr[c]), int(row_ptr[c + 1] - row_ptr[c])
            mult = _intersect(col, vb, vn, cb, cn)
            if not (float(mult) > max_jaccard * float(vn + cn - mult)):
                break
        if c == v:
            c = v + 1 if v + 1 < side_hi else side_lo
        out.append(c)
    return out


def floyd_sample(n, m, words):
    """Floyd's m distinct values of [0, n): value k = floor(words[k] * (j+1) / 2^32), j = n-m+k; j if taken."""
    out = []
    for k in range(m):
        j = n - m + k
        t = (words[k] * (j + 1)) >> 32
        if t in out:
            t = j
        out.append(t)
    return out


def sample_occurrences(c, n_occ, iteration, seed_occ):
    m = min(n_occ, 10)
    words = []
    for blk in range(3):
        words.extend(_philox(seed_occ, c, iteration, blk, 0))
    return floyd_sample(n_occ, m, words)


def occurrence_context(o, c, tokens, tok_walk, walk_off, pool_row, ws, ns, seed_neg):
    """Contexts and negatives of the occurrence at token position o of vertex c
    (src/bine_graph_utils.py:169-187: window within the walk, tokens equal to the centre skipped; negatives
    = distinct pool slots, dropped when empty, inside the window or repeated)."""
    wk = int(tok_walk[o])
    w0, w1 = int(walk_off[wk]), int(walk_off[wk + 1])
    s, e = max(w0, o - ws), min(w1, o + ws + 1)
    window = [int(x) for x in tokens[s:e]]
    contexts = [z for z in window if z != c]
    words = list(_philox(seed_neg, o, o >> 32, 0, 0)) + list(_philox(seed_neg, o, o >> 32, 1, 0))
    m2 = min(ns, len(pool_row))
    negs = []
    for slot in floyd_sample(len(pool_row), m2, words):
        cand = int(pool_row[slot])
        if cand < 0 or cand in window or cand in negs or cand == c:   # -1: empty slot of a short pool
            continue
        negs.append(cand)
    return contexts, negs


# ------------------------------------------------------------------------------- (L) training arithmetic
def skip_gram(center, context, negs, emb, ctx, lam, pa):
    """src/bine_train.py:243-274 on dense tables (emb = 'embedding_vectors', ctx = 'context_vectors')."""
    loss = 0.0
    I_z = {center: 1}
    for node in negs:
        I_z[node] = 0
    V = np.array(emb[context])
    update = np.zeros_like(V)
    for u in I_z.keys():
        Theta = np.array(ctx[u])
        X = float(max(V.dot(Theta), 0))
        sigmod = 1.0 / (1 + (math.exp(-X * 1.0)))
        update += pa * lam * (I_z[u] - sigmod) * Theta
        ctx[u] += pa * lam * (I_z[u] - sigmod) * V
        try:
            loss += pa * (I_z[u] * math.log(sigmod) + (1 - I_z[u]) * math.log(1 - sigmod))
        except ValueError:
            pass
    return update, loss


def kl_divergence(e_ij, u, v, emb, lam, gamma):
    """src/bine_train.py:277-309."""
    U = np.array(emb[u])
    V = np.array(emb[v])
    X = float(max(U.dot(V), 0))
    sigmod = 1.0 / (1 + (math.exp(-X * 1.0)))
    update_u = gamma * lam * ((e_ij * (1 - sigmod)) * 1.0 / LN10) * V
    update_v = gamma * lam * ((e_ij * (1 - sigmod)) * 1.0 / LN10) * U
    loss = gamma * e_ij * math.log(sigmod)
    return update_u, update_v, loss


def train(edge_u, edge_v, edge_w, emb, ctx, occ_ptr, occ_pos, tokens, tok_walk, walk_off, pool, ws, ns, alpha, beta,
          gamma, lam, max_iter, seed_occ, seed_neg, epsilon=1e-3, first=None):
    """The loop of src/bine_train.py:452-504 over the rating list, with the device's sampling rule for the
    occurrences / negatives (the reference uses the global `random`).  emb, ctx: float64 [N, d], updated in
    place.  `first` (optional, uint8 per rating: bit 0 user, bit 1 item) replaces the visited dictionaries when
    only a sample of the rating list is passed.  Returns (lam, per-iteration losses)."""
    last_loss = 0.0
    losses = []
    for it in range(max_iter):
        loss = 0.0
        seen = set()
        for e in range(len(edge_u)):
            u, v, w = int(edge_u[e]), int(edge_v[e]), float(edge_w[e])
            for side, (c, pa) in enumerate(((u, alpha), (v, beta))):
                if first is not None:
                    if not (int(first[e]) >> side) & 1:
                        continue
                elif c in seen:
                    continue
                seen.add(c)
                ob = int(occ_ptr[c])
                n_occ = int(occ_ptr[c + 1]) - ob
                for idx in sample_occurrences(c, n_occ, it, seed_occ):
                    o = int(occ_pos[ob + idx])
                    contexts, negs = occurrence_context(o, c, tokens, tok_walk, walk_off, pool[c], ws, ns, seed_neg)
                    for z in contexts:
                        tmp_z, tmp_loss = skip_gram(c, z, negs, emb, ctx, lam, pa)
                        emb[z] += tmp_z
                        loss += tmp_loss
            update_u, update_v, tmp_loss = kl_divergence(w, u, v, emb, lam, gamma)
            loss += tmp_loss
            emb[u] += update_u
            emb[v] += update_v
        delta_loss = abs(loss - last_loss)
        if last_loss > loss:
            lam *= 1.05
        else:
            lam *= 0.95
        last_loss = loss
        losses.append(loss)
        if delta_loss < epsilon:
            break
    return lam, losses


def init_rows(n, dim, seed):
    """(P) init: U[0,1)^dim rows scaled to unit l2 norm; element c of row r, table t from Philox
    (r lo, r hi, c >> 1, t)."""
    out = np.zeros((2, n, dim))
    for t in range(2):
        for r in range(n):
            for c in range(dim):
                w = _philox(seed, r, r >> 32, c >> 1, t)
                out[t, r, c] = _u53(w[2], w[3]) if c & 1 else _u53(w[0], w[1])
            out[t, r] /= math.sqrt(float((out[t, r] ** 2).sum()))
    return out[0], out[1]
