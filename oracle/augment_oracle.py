"""ORACLE — test infrastructure, not product code.

Restatement of the reference's similarity-driven edge selection (src/main_link.py:358-453) in
plain Python, one function per reference function.  PARITY UNPINNED: main_link.py cannot be
imported here (it imports pathos and gensim, both absent) and the reference holds no fixture for
these functions; the text was restated by reading it."""
import numpy as np


def similarity(emb, a, b):
    """gensim KeyedVectors.similarity: dot of the unit vectors (float32), src/main_link.py:359-360."""
    x, y = np.asarray(emb[a], dtype=np.float32), np.asarray(emb[b], dtype=np.float32)
    return float(np.dot(x / np.linalg.norm(x), y / np.linalg.norm(y)))


def js(p, q):                                                       # :351-356
    from scipy.stats import entropy
    p_norm = p / p.sum()
    q_norm = q / q.sum()
    m = (p_norm + q_norm) / 2
    return (entropy(p_norm, m) + entropy(q_norm, m)) / 2


def get_similarity(emb, user1, user2, sim_method="cos"):           # :358-365
    if sim_method == "cos":
        return similarity(emb, user1, user2)
    if sim_method == "pearson":
        from scipy.stats import pearsonr
        return pearsonr(np.asarray(emb[user1]), np.asarray(emb[user2]))[0]
    if sim_method == "jsd":
        return js(np.asarray(emb[user1]), np.asarray(emb[user2]))
    raise ValueError(sim_method)


SIM_METHOD = "cos"   # the reference threads sim_method through every call; a module switch keeps the restatement short


def _sim_list(emb, user_nodes, i, user):
    lst = [get_similarity(emb, user, user2, SIM_METHOD) for user2 in user_nodes]   # :385 / :403 / :420 / :437 / :453
    lst[i] = 0
    return lst


def get_add_edge_by_ratio(user_nodes, ratio, emb):                # :379-394
    add_edge_num = int(len(user_nodes) * ratio)
    add_edge = []
    for i, user in enumerate(user_nodes):
        similarities = sorted(zip(user_nodes, _sim_list(emb, user_nodes, i, user)), key=lambda tup: -tup[1])
        add_edge.extend([(user, x[0], 1) for x in similarities[:add_edge_num]])
    return add_edge


def get_add_edge_by_step(user_nodes, thre, emb):                   # :396-409
    add_edge = []
    for i, user in enumerate(user_nodes):
        similarities = [x for x in zip(user_nodes, _sim_list(emb, user_nodes, i, user)) if x[1] > thre]
        add_edge.extend([(user, x[0], 1) for x in similarities])
    return add_edge


def get_add_edge_by_relu(user_nodes, thre, emb):                   # :411-424
    add_edge = []
    for i, user in enumerate(user_nodes):
        similarities = [x for x in zip(user_nodes, _sim_list(emb, user_nodes, i, user)) if x[1] > thre]
        add_edge.extend([(user, x[0], x[1]) for x in similarities])
    return add_edge


def get_add_edge_linear(user_nodes, emb):                          # :442-453
    add_edge = []
    for i, user in enumerate(user_nodes):
        add_edge.extend([(user, x[0], x[1]) for x in zip(user_nodes, _sim_list(emb, user_nodes, i, user))])
    return add_edge


def add_user_edge(user_nodes, emb, mode, ratio, thre, sim_method="cos"):   # :455-475
    global SIM_METHOD
    SIM_METHOD = sim_method
    if mode == "ratio":
        return get_add_edge_by_ratio(user_nodes, ratio, emb)
    if mode == "step":
        return get_add_edge_by_step(user_nodes, thre, emb)
    if mode == "relu":
        return get_add_edge_by_relu(user_nodes, thre, emb)
    if mode == "relu-ratio":
        return get_add_edge_by_relu(user_nodes, ratio, emb)           # :469 passes the ratio as threshold
    if mode == "linear":
        return get_add_edge_linear(user_nodes, emb)
    raise Exception("user-edges-mode value fault: " + str(mode))
