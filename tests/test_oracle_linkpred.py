"""CPU checks of the top-k link-prediction restatement (oracle/linkpred_oracle.py, src/main_link.py:62-170):
the vectorised form used for the 2 000-node GPU test equals the literal, loop-by-loop form on a small graph,
for the separated (user x item) and the unseparated layouts and several `segment` values; precision_at_k and
calculate_pop on hand-made cases.  PARITY UNPINNED (no fixture in the reference)."""
import numpy as np
import pytest

from oracle import linkpred_oracle as lo


def _case(seed, unseparated):
    rs = np.random.RandomState(seed)
    n_u, n_i, d = 23, 17, 12
    users = [int(x) for x in rs.permutation(500)[:n_u]]
    items = [int("9999999%d" % x) for x in rs.permutation(300)[:n_i]]
    nodes = users + items
    emb = {str(x): rs.normal(size=d).astype(np.float32) for x in nodes}
    if unseparated:
        pairs = [(nodes[a], nodes[b]) for a, b in rs.randint(0, len(nodes), size=(120, 2)) if a != b]
    else:
        pairs = [(users[a], items[b]) for a, b in zip(rs.randint(0, n_u, 120), rs.randint(0, n_i, 120))]
    train, test = pairs[:60], pairs[60:]
    g = {x: set() for x in nodes}
    for a, b in train:
        g[a].add(b)
        g[b].add(a)
    return emb, g, train, test


@pytest.mark.parametrize("unseparated", [False, True])
@pytest.mark.parametrize("segment", [1, 3, 10])
def test_vectorised_equals_literal(unseparated, segment):
    emb, g, train, test = _case(7, unseparated)
    ks = [1, 10, 50, 100]
    res, fin = lo.link_prediction(unseparated, segment, g, emb, train, test, ks)
    vres, vfin = lo.link_prediction_vectorised(unseparated, g, emb, train, test, ks)
    for k in ks:
        assert [p for p, _, _ in res[k]] == [p for p, _, _ in vres[k]], k
        np.testing.assert_allclose([s for _, s, _ in res[k]], [s for _, s, _ in vres[k]], atol=1e-6)
        assert [q for _, _, q in res[k]] == [q for _, _, q in vres[k]]
        assert fin[k] == vfin[k]
    train_set = {(str(a), str(b)) for a, b in train}
    assert not any(p in train_set for p, _, _ in res[100])


def test_precision_at_k_counts_both_orientations():
    test = [("1", "2"), ("3", "4")]
    assert lo.precision_at_k([("1", "2"), ("4", "3"), ("5", "6"), ("2", "3")], test) == 0.5


def test_calculate_pop_python2_arithmetic():
    g = {1: [0] * 3, 2: [0] * 4, 99999995: [0] * 7}
    assert lo.calculate_pop(True, g, [("1", "2")]) == [3]            # int((3 + 4) / 2) under Python 2
    assert lo.calculate_pop(False, g, [("1", "99999995"), ("99999995", "2")]) == [7, 7]
