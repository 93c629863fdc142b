"""ORACLE — test infrastructure, not product code.

Restatement of the reference's top-k link prediction (src/main_link.py:62-170): precision_at_k, links_score,
make_links_and_score, calculate_pop, link_prediction — function by function, in plain Python with the
reference's Python-2 arithmetic (`/` on ints floors: `batch_nodes`, `int((a+b)/2)`, `avg_pop`).
PARITY UNPINNED: main_link.py cannot be imported here (pathos, gensim absent) and holds no fixture for these
functions; the text was restated by reading it.  `emb` is a dict str(node) -> float32 vector; `g` a dict
node(int) -> set/list of neighbours (what `len(g[node])` needs, :112-119)."""
import numpy as np

KS = [1, 10, 50, 100, 500, 1000]   # default `ks` of link_prediction (:123)


def similarity(emb, a, b):
    """gensim KeyedVectors.similarity (link_method "cos", :44-49): float32 dot of the unit vectors."""
    x, y = np.asarray(emb[a], dtype=np.float32), np.asarray(emb[b], dtype=np.float32)
    return float(np.dot(x / np.linalg.norm(x), y / np.linalg.norm(y)))


def precision_at_k(pred_k, test_edges):                                   # :62-67
    count = 0.0
    test = set(test_edges)
    for pred in pred_k:
        if pred in test or (pred[1], pred[0]) in test:
            count += 1
    return count / len(pred_k)


def links_score(emb, edges_to_eval, ks):                                  # :92-108
    pred_score = [similarity(emb, e[0], e[1]) for e in edges_to_eval]
    score_index = np.argsort(pred_score)[::-1]
    return {k: [(edges_to_eval[x], pred_score[x]) for x in score_index[:k]] for k in ks}


def make_links_and_score(emb, nodes, start_point, end_point, train_edges, ks):   # :69-90
    if len(nodes) == 1:
        nodes = nodes[0]
        edges_to_eval = set((nodes[i], nodes[j]) for i in range(start_point, end_point) for j in range(i + 1, len(nodes)))
    else:
        user_nodes, item_nodes = nodes
        edges_to_eval = set((user_nodes[i], item) for i in range(start_point, end_point) for item in item_nodes)
    edges_to_eval -= set(train_edges)
    return links_score(emb, sorted(edges_to_eval), ks)     # sorted(): a deterministic stand-in for set order


def calculate_pop(unseparated, g, chosen_k_links):                        # :110-120
    pop_list = []
    for link in chosen_k_links:
        node_a, node_b = int(link[0]), int(link[1])
        if unseparated:
            pop_list.append(int((len(g[node_a]) + len(g[node_b])) // 2))
        else:
            if str(node_a).startswith('9999999'):
                pop_list.append(len(g[node_a]))
            elif str(node_b).startswith('9999999'):
                pop_list.append(len(g[node_b]))
    assert len(chosen_k_links) == len(pop_list)
    return pop_list


def link_prediction(unseparated, segment, g, emb, train_edges, test_edges, ks=KS):   # :123-170
    nodes = [str(x) for x in sorted(int(x) for x in emb.keys())]
    if not unseparated:
        item_nodes = [x for x in nodes if x.startswith('9999999')]
        user_nodes = [x for x in nodes if not x.startswith('9999999')]
        nodes = (user_nodes, item_nodes)
    else:
        nodes = (nodes,)
    batch_nodes = len(nodes[0]) // segment
    test_edges = [(str(x[0]), str(x[1])) for x in test_edges]
    train_edges = [(str(x[0]), str(x[1])) for x in train_edges]
    results = {k: [] for k in ks}
    for i in range(segment):
        end = batch_nodes * (i + 1) if i != segment - 1 else len(nodes[0])
        partial = make_links_and_score(emb, nodes, batch_nodes * i, end, train_edges, ks)
        for k in ks:
            results[k].extend(partial[k])
    final_results = {}
    for k in ks:
        results[k] = list(set(results[k]))
        temp = sorted(results[k], key=lambda tup: (-tup[1], tup[0]))
        chosen_k_links = [x[0] for x in temp[:k]]
        pops = calculate_pop(unseparated, g, chosen_k_links)
        avg_pop = sum(pops) // len(pops)
        results[k] = [x + (pops[i],) for i, x in enumerate(temp[:k])]
        final_results[k] = (precision_at_k(chosen_k_links, test_edges), avg_pop)
    return results, final_results


def link_prediction_vectorised(unseparated, g, emb, train_edges, test_edges, ks=KS):
    """The same result through one float32 matrix product (for graphs where the literal loops above take too
    long); checked against link_prediction() on a small graph in tests/test_oracle_linkpred.py."""
    names = [str(x) for x in sorted(int(x) for x in emb.keys())]
    if unseparated:
        rows, cols = names, names
    else:
        cols = [x for x in names if x.startswith('9999999')]
        rows = [x for x in names if not x.startswith('9999999')]
    unit = lambda n: np.stack([np.asarray(emb[x], np.float32) / np.linalg.norm(np.asarray(emb[x], np.float32)) for x in n])
    S = unit(rows) @ unit(cols).T
    if unseparated:
        S[np.tril_indices(len(rows))] = -np.inf
    ri, ci = {x: i for i, x in enumerate(rows)}, {x: i for i, x in enumerate(cols)}
    for a, b in train_edges:
        a, b = str(a), str(b)
        if a in ri and b in ci and (not unseparated or ri[a] < ci[b]):
            S[ri[a], ci[b]] = -np.inf
    kmax = min(max(ks), int(np.isfinite(S).sum()))
    flat = np.argsort(-S, axis=None, kind="stable")[:kmax]
    r, c = np.unravel_index(flat, S.shape)
    top = [((rows[i], cols[j]), float(S[i, j])) for i, j in zip(r, c)]
    test = [(str(x[0]), str(x[1])) for x in test_edges]
    results, final_results = {}, {}
    for k in ks:
        chosen = [x[0] for x in top[:k]]
        pops = calculate_pop(unseparated, g, chosen)
        results[k] = [x + (pops[i],) for i, x in enumerate(top[:k])]
        final_results[k] = (precision_at_k(chosen, test), sum(pops) // len(pops))
    return results, final_results
