"""ORACLE — test infrastructure, not product code.

Python side of the SGNS comparator and the link-prediction scoring flow.

PARITY UNPINNED for the SGNS half: gensim 3.2.0 (requirements.txt:17 of the reference) is
not in the reference tree and not installable offline, and the reference has no test
vector at that boundary.  ``vocab_tables`` restates gensim's scale_vocab / make_cum_table
with Python floats, word by word, in gensim's own vocabulary order (descending count);
the scoring functions restate src/main_link.py:173-204,519-563 of the reference.
"""
import math
import random

import numpy as np


def vocab_tables(counts, sample=1e-3, power=0.75):
    """counts: sequence of corpus counts per word id.  Returns (sample_int, cum_table) as
    uint32 arrays indexed by word id (ids with count 0 are outside the vocabulary)."""
    counts = [int(c) for c in counts]
    n = len(counts)
    retain_total = sum(counts)
    if not sample:
        threshold_count = retain_total
    elif sample < 1.0:
        threshold_count = sample * retain_total
    else:
        threshold_count = int(sample * (3 + math.sqrt(5)) / 2)
    sample_int = np.zeros(n, dtype=np.uint32)
    for w, v in enumerate(counts):
        if v == 0:
            sample_int[w] = 2**32 - 1
            continue
        word_probability = (math.sqrt(v / threshold_count) + 1) * (threshold_count / v)
        if word_probability >= 1.0:
            word_probability = 1.0
        sample_int[w] = min(int(round(word_probability * 2**32)), 2**32 - 1)
    # make_cum_table (domain 2^31 - 1); word ids keep their order (the distribution does not
    # depend on the order the words are laid out in)
    domain = 2**31 - 1
    train_words_pow = 0.0
    for v in counts:
        train_words_pow += v**power
    cum = np.zeros(n, dtype=np.uint32)
    cumulative = 0.0
    for w, v in enumerate(counts):
        cumulative += v**power
        cum[w] = int(round(cumulative / train_words_pow * domain))
    last_nz = max(w for w, v in enumerate(counts) if v > 0)
    assert cum[last_nz] == domain
    cum[last_nz:] = domain
    return (sample_int if sample else None), cum


# ---------------------------------------------------------------------------- link prediction
def split_edges(edges, test_ratio=0.5, seed=123):
    """src/main_link.py:525-526 (settings.py:1-2): sklearn train_test_split on the edge array."""
    from sklearn.model_selection import train_test_split
    tr, te = train_test_split(np.asarray(edges), test_size=test_ratio, random_state=seed)
    return tr, te


def build_neg_samples(nodes, true_edges, seed=0):
    """src/main_link.py:191-204: as many non-edges as there are edges, sampled uniformly
    over node pairs.  (The reference draws from an unseeded `random`; a seed is taken here
    so CPU and GPU runs score the same pairs, and membership is tested on the normalised
    pair — the reference tests (min,max) against un-normalised tuples.)"""
    rnd = random.Random(seed)
    true_set = set((min(int(a), int(b)), max(int(a), int(b))) for a, b in true_edges)
    nodes = list(nodes)
    false_samples = set()
    while len(false_samples) < len(true_set):
        a, b = rnd.sample(nodes, 2)
        e = (min(a, b), max(a, b))
        if e in true_set or e in false_samples:
            continue
        false_samples.add(e)
    return sorted(false_samples)


def cosine_scores(vectors_by_node, pairs):
    out = np.empty(len(pairs))
    for i, (a, b) in enumerate(pairs):
        x, y = vectors_by_node[int(a)], vectors_by_node[int(b)]
        out[i] = float(np.dot(x / np.linalg.norm(x), y / np.linalg.norm(y)))
    return out


def roc_score(vectors_by_node, edges_pos, edges_neg):
    """src/main_link.py:173-189 with link_method == "cos"."""
    from sklearn.metrics import average_precision_score, roc_auc_score
    pp = cosine_scores(vectors_by_node, edges_pos)
    pn = cosine_scores(vectors_by_node, edges_neg)
    preds = np.hstack([pp, pn])
    labels = np.hstack([np.ones(len(pp)), np.zeros(len(pn))])
    return roc_auc_score(labels, preds), average_precision_score(labels, preds)
