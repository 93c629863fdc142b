"""CPU restatement of the reference's LSH negative pools — TEST INFRASTRUCTURE ONLY (never imported by the product).

What it follows: src/bine_lsh.py:7-51 (`construct_lsh`, `call_get_negs_by_lsh`), called from
src/bine_graph_utils.py:145-148 (`get_negs`), on top of the third-party `datasketch` package, which is pinned
(requirements.txt:11 `datasketch==1.2.5`) but ABSENT here (not importable, not in the wheelhouse, no network) —
so `MinHash` and `MinHashLSHForest` are restated from the published 1.2.5 source (datasketch/minhash.py,
datasketch/lshforest.py), plain Python/numpy, dictionaries and sorted lists exactly as that source keeps them:

  MinHash(num_perm=128, seed=1): permutations (a_j, b_j) = (RandomState(1).randint(1, M, uint64),
      RandomState.randint(0, M, uint64)) drawn alternately, M = 2^61 - 1; update(b): hv = first four bytes of
      sha1(b), little endian; hashvalues_j = min(hashvalues_j, ((a_j*hv + b_j) mod 2^64) mod M  &  (2^32 - 1))
      (the uint64 product wraps, as numpy's does); empty set = 2^32 - 1 everywhere.
  MinHashLSHForest(num_perm=128, l=8): k = 16 hash values per tree, key of a tree = big-endian bytes of its 16 values
      (`hs.byteswap().data`), buckets in insertion order, `index()` sorts the distinct keys of each tree;
      query(m, k): r = 16 … 1, trees 0 … 7, all keys whose first r values equal m's, in sorted-key then insertion
      order, into a set until it holds k; the result is that set.

Parity status: **unpinned** — no fixture in the reference holds a MinHash value, a forest query or a pool, and
datasketch cannot be run here.  The restatement is pinned only against what can be checked independently:
hashlib's SHA-1, numpy's legacy RandomState stream (stable by numpy's own compatibility promise) and the
collision-probability property P[h_j(A) = h_j(B)] = Jaccard(A, B) (tests/test_bine_lsh_host.py).

Two layers, as in oracle/bine_oracle.py:
 (L) literal: `MinHash`, `LSHForest`, `negs_by_lsh_literal` — the reference text with `random.sample` left to the
     caller's `random.Random`;
 (P) device algorithm: `signatures`, `forest_query_all`, `leaders`, `exclusions`, `sample_pool` — arrays and
     Philox draws exactly as csrc/n2v_lsh.hip computes them; tests check (P) == HIP bit for bit and the
     deterministic part of (P) (signatures, query sets, clusters, exclusion sets) == (L) exactly.
"""
import hashlib
import struct
from collections import defaultdict

import numpy as np

from oracle.n2v_oracle import philox4x32_10

MERSENNE = (1 << 61) - 1
MAX_HASH = (1 << 32) - 1
NUM_PERM = 128
TREES = 8
DEPTH = NUM_PERM // TREES  # 16


def sha1_hash32(b):
    """datasketch 1.2.5 minhash.py `update`: struct.unpack('<I', sha1(b).digest()[:4])[0]."""
    return struct.unpack("<I", hashlib.sha1(b).digest()[:4])[0]


def permutations(num_perm=NUM_PERM, seed=1):
    """datasketch 1.2.5 minhash.py `__init__`: the (a, b) parameters, shape (2, num_perm), uint64."""
    gen = np.random.RandomState(seed)
    return np.array([(gen.randint(1, MERSENNE, dtype=np.uint64), gen.randint(0, MERSENNE, dtype=np.uint64))
                     for _ in range(num_perm)], dtype=np.uint64).T


# ------------------------------------------------------------------------------------------------ (L) literal
class MinHash:
    _perm = None

    def __init__(self, num_perm=NUM_PERM):
        if MinHash._perm is None or MinHash._perm.shape[1] != num_perm:
            MinHash._perm = permutations(num_perm)
        self.permutations = MinHash._perm
        self.hashvalues = np.ones(num_perm, dtype=np.uint64) * np.uint64(MAX_HASH)

    def update(self, b):
        hv = np.uint64(sha1_hash32(b))
        a, bb = self.permutations
        with np.errstate(over="ignore"):
            phv = np.bitwise_and((a * hv + bb) % np.uint64(MERSENNE), np.uint64(MAX_HASH))
        self.hashvalues = np.minimum(phv, self.hashvalues)


class LSHForest:
    def __init__(self, num_perm=NUM_PERM, l=TREES):
        self.l = l
        self.k = int(num_perm / l)
        self.hashtables = [defaultdict(list) for _ in range(l)]
        self.hashranges = [(i * self.k, (i + 1) * self.k) for i in range(l)]
        self.keys = {}
        self.sorted_hashtables = [[] for _ in range(l)]

    @staticmethod
    def _H(hs):
        return bytes(hs.byteswap().data)

    def add(self, key, minhash):
        if key in self.keys:
            raise ValueError("The given key has already been added")
        self.keys[key] = [self._H(minhash.hashvalues[s:e]) for s, e in self.hashranges]
        for H, table in zip(self.keys[key], self.hashtables):
            table[H].append(key)

    def index(self):
        for i, table in enumerate(self.hashtables):
            self.sorted_hashtables[i] = sorted(table.keys())

    def _query(self, minhash, r):
        hps = [self._H(minhash.hashvalues[s:s + r]) for s, _ in self.hashranges]
        size = len(hps[0])
        for ht, hp, table in zip(self.sorted_hashtables, hps, self.hashtables):
            lo, hi = 0, len(ht)
            while lo < hi:                       # first x with ht[x][:size] >= hp
                mid = (lo + hi) // 2
                if ht[mid][:size] >= hp:
                    hi = mid
                else:
                    lo = mid + 1
            j = lo
            while j < len(ht) and ht[j][:size] == hp:
                for key in table[ht[j]]:
                    yield key
                j += 1

    def query(self, minhash, k):
        results = set()
        r = self.k
        while r > 0:
            for key in self._query(minhash, r):
                results.add(key)
                if len(results) >= k:
                    return list(results)
            r -= 1
        return list(results)


def negs_by_lsh_literal(keys, neighbour_labels, k=200, sample_num=200, rng=None):
    """call_get_negs_by_lsh (src/bine_lsh.py:27-51) for one side.  keys: list of vertex labels in dictionary order;
    neighbour_labels[i]: the labels (str) of keys[i]'s neighbours (`for d in values[i]` walks a dict of
    {neighbour: rating}).  Returns (negs_dict, info) where info carries the deterministic intermediates:
    sim[i] (set of keys), leader_of[i] (index of the key whose turn produced i's pool), excluded[i] (set, leaders only)."""
    forest, ms = LSHForest(), []
    for i, key in enumerate(keys):
        m = MinHash()
        for d in neighbour_labels[i]:
            m.update(d.encode("utf8"))
        ms.append(m)
        forest.add(key, m)
    forest.index()
    index_of = {key: i for i, key in enumerate(keys)}
    visited, negs, sim, leader_of, excluded = set(), {}, {}, {}, {}
    for i in range(len(keys)):
        if i in visited:
            continue
        visited.add(i)
        record = [i]
        sim_list = set(forest.query(ms[i], k))
        sim[i] = sim_list
        gone = set(sim_list)
        for j in sim_list:
            ind = index_of[j]
            if ind not in visited:
                visited.add(ind)
                record.append(ind)
            child = set(forest.query(ms[ind], k))
            sim[ind] = child
            gone |= child
        excluded[i] = gone
        rest = [key for key in keys if key not in gone]
        pool = rng.sample(rest, min(sample_num, len(rest))) if rng is not None else rest
        for j in record:
            negs[keys[j]] = pool
            leader_of[j] = i
    return negs, dict(sim=sim, leader_of=leader_of, excluded=excluded, signatures=[m.hashvalues for m in ms])


# ------------------------------------------------------------------------------------- (P) device algorithm
def signatures(row_ptr, col, hv, v_lo, v_hi, perm=None):
    """uint32[v_hi - v_lo][128]: MinHash of every vertex's neighbour set, hv[c] = sha1_hash32 of vertex c's label."""
    perm = permutations() if perm is None else perm
    a, b = perm
    out = np.full((v_hi - v_lo, NUM_PERM), MAX_HASH, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for v in range(v_lo, v_hi):
            for e in range(row_ptr[v], row_ptr[v + 1]):
                phv = ((a * np.uint64(hv[col[e]]) + b) % np.uint64(MERSENNE)) & np.uint64(MAX_HASH)
                out[v - v_lo] = np.minimum(out[v - v_lo], phv)
    return out.astype(np.uint32)


def forest_layout(sig):
    """Per tree: order (sorted position -> local vertex, ties by vertex = insertion order) and lcp (number of equal
    leading values of sorted entries i-1 and i; lcp[0] = 0)."""
    n = sig.shape[0]
    order = np.empty((TREES, n), dtype=np.int64)
    lcp = np.zeros((TREES, n), dtype=np.int32)
    for t in range(TREES):
        s = sig[:, t * DEPTH:(t + 1) * DEPTH]
        o = np.lexsort(tuple(s[:, c] for c in range(DEPTH - 1, -1, -1)))  # stable, first column most significant
        order[t] = o
        eq = s[o][1:] == s[o][:-1]
        lcp[t, 1:] = np.cumprod(eq, axis=1).sum(axis=1)
    return order, lcp


def forest_query_all(sig, k=200):
    """sim[i]: the keys (local vertex ids) of query(ms[i], k) in the order the forest yields NEW keys — the set is
    what the reference uses, the order is what the kernel writes."""
    n = sig.shape[0]
    order, lcp = forest_layout(sig)
    pos = np.empty_like(order)
    for t in range(TREES):
        pos[t, order[t]] = np.arange(n)
    sims = []
    for v in range(n):
        found, seen = [], set()
        lo = [int(pos[t, v]) for t in range(TREES)]
        hi = [x + 1 for x in lo]
        first = True
        for r in range(DEPTH, 0, -1):
            for t in range(TREES):
                nlo, nhi = lo[t], hi[t]
                while nlo > 0 and lcp[t, nlo] >= r:
                    nlo -= 1
                while nhi < n and lcp[t, nhi] >= r:
                    nhi += 1
                wings = (range(nlo, nhi),) if first else (range(nlo, lo[t]), range(hi[t], nhi))
                lo[t], hi[t] = nlo, nhi
                for rng_ in wings:
                    for i in rng_:
                        key = int(order[t, i])
                        if key not in seen:
                            seen.add(key)
                            found.append(key)
                            if len(found) >= k:
                                break
                    if len(found) >= k:
                        break
                if len(found) >= k:
                    break
            first = False
            if len(found) >= k:
                break
        sims.append(found)
    return sims


def leaders(sims):
    """owner[i] = the vertex whose turn of the `for i in range(len(keys))` loop produced i's pool
    (src/bine_lsh.py:32-36,41-45): i itself when it is unvisited at its turn."""
    n = len(sims)
    owner = np.full(n, -1, dtype=np.int64)
    for i in range(n):
        if owner[i] >= 0:
            continue
        owner[i] = i
        for j in sims[i]:
            if owner[j] < 0:
                owner[j] = i
    return owner


def exclusions(sims, i):
    """sim(i) | U_{j in sim(i)} sim(j)  (src/bine_lsh.py:39-47)."""
    gone = set(sims[i])
    for j in sims[i]:
        gone |= set(sims[j])
    return gone


def _u53(a, b):
    return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0


def sample_pool(gone, n_side, pool_size, seed, leader):
    """`random.sample(total_list, min(sample_num, len(total_list)))` as the kernel draws it: rounds of 64 candidates
    floor(u * n_side), u from Philox(seed; leader, round, lane); a candidate is taken when it is not excluded, not
    already taken and no lower lane of the same round proposes it; lanes in order fill the pool.  When fewer than
    pool_size vertices are left, all of them in ascending order, then -1."""
    rest = n_side - len(gone)
    out = np.full(pool_size, -1, dtype=np.int32)
    if rest <= pool_size:
        keep = [c for c in range(n_side) if c not in gone]
        out[:len(keep)] = keep
        return out
    taken, cnt, rnd = set(), 0, 0
    while cnt < pool_size:
        for lane in range(64):
            r = philox4x32_10((leader & 0xFFFFFFFF, rnd & 0xFFFFFFFF, lane, 0),
                              (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
            c = int(np.floor(_u53(r[0], r[1]) * float(n_side)))
            c = min(c, n_side - 1)
            if c in gone or c in taken:
                continue
            taken.add(c)
            out[cnt] = c
            cnt += 1
            if cnt >= pool_size:
                break
        rnd += 1
    return out
