"""Drop-in for the reference's src/bine_graph_utils.py `GraphUtils` on MI355X.

Same method names and call order as src/bine_train.py:433-447,600-612 uses them; the state the reference
keeps in Python dicts lives in HBM (n2v_hip/bine.py) and is exposed through the same attribute names where
callers read them (`node_u`, `node_v`, `edge_list`, `edge_dict_u`, `authority_u/v`, `walks_u/v`).
There is no CPU path: without the HIP library / a GPU the device steps raise.
"""
import os

from n2v_hip import bine


class GraphUtils(object):
    def __init__(self, model_path, device=None, seed=0):
        self.model_path = model_path
        self.device = device
        self.seed = seed
        self.graph = None      # n2v_hip.bine.BipartiteGraph (host)
        self.engine = None     # n2v_hip.bine.BineEngine (device)
        self.authority_u, self.authority_v = {}, {}

    # src/bine_graph_utils.py:32-58
    def construct_training_graph(self, filename=None):
        if filename is None:
            filename = os.path.join(self.model_path, "ratings_train.dat")
        self.graph = bine.BipartiteGraph.read(filename)
        self.engine = bine.BineEngine(self.graph, device=self.device, seed=self.seed)

    def construct_from_arrays(self, users, items, ratings):
        """Same as construct_training_graph for ratings already in memory (synthetic configurations)."""
        self.graph = bine.BipartiteGraph(users, items, ratings)
        self.engine = bine.BineEngine(self.graph, device=self.device, seed=self.seed)

    @property
    def node_u(self):
        return self.graph.user_labels.tolist()

    @property
    def node_v(self):
        return self.graph.item_labels.tolist()

    @property
    def edge_list(self):
        g = self.graph
        return [(g.user_labels[u], g.item_labels[v - g.n_u], w)
                for u, v, w in zip(g.edge_u.tolist(), g.edge_v.tolist(), g.edge_w.tolist())]

    @property
    def edge_dict_u(self):
        out = {}
        for u, v, w in self.edge_list:
            out.setdefault(u, {})[v] = w
        return out

    # src/bine_graph_utils.py:60-86
    def calculate_centrality(self):
        self.engine.calculate_centrality()

    def _fill_authority(self):
        g, a = self.graph, self.engine.auth_scaled.cpu().numpy()
        self.authority_u = dict(zip(g.user_labels.tolist(), a[: g.n_u].tolist()))
        self.authority_v = dict(zip(g.item_labels.tolist(), a[g.n_u:].tolist()))

    # src/bine_graph_utils.py:112-131 (the --large 1 path; --large 0 walks the same projections from files,
    # src/bine_graph_utils.py:88-110, and yields the same distribution of walks)
    def homogeneous_graph_random_walks_for_large_bipartite_graph(self, datafile=None, percentage=0.15, maxT=32, minT=1):
        self.engine.generate_walks(percentage=percentage, maxT=maxT, minT=minT)
        self._fill_authority()

    homogeneous_graph_random_walks = homogeneous_graph_random_walks_for_large_bipartite_graph

    @property
    def walks_u(self):
        return self.engine.walks_as_lists("u")

    @property
    def walks_v(self):
        return self.engine.walks_as_lists("v")

    # src/bine_graph_utils.py:145-148
    def get_negs(self, pool_size=200):
        self.engine.build_negative_pools(pool_size=pool_size)
        return self.engine.pool[: self.graph.n_u], self.engine.pool[self.graph.n_u:]

    # src/bine_graph_utils.py:150-191: only the occurrence index is materialised; windows and negatives are
    # formed inside the training kernel from it
    def get_context_and_negatives(self, *unused_args):
        self.engine.build_occurrences()
        return self.engine.occ_ptr, self.engine.occ_pos
