'''
Drop-in for the reference's ``src/main.py`` (node2vec pipeline) on MI355X.

Same flags and the same three calls — ``read_graph()``, ``node2vec.Graph(...)``
-> ``preprocess_transition_probs()`` -> ``simulate_walks()``, ``learn_embeddings(walks)`` —
with the walk and the skip-gram training done by the HIP kernels of this package instead of
pure Python + gensim (src/main.py:66-101).  ``learn_embeddings`` reads the module-global
``args`` exactly as the reference does (src/main.py:87).
'''
import argparse
import os

import torch

import node2vec
from n2v_hip import csr as _csr
from n2v_hip import sgns as _sgns

args = None


# (flag, type, default, help) of src/main.py:18-64 — same names, same defaults
_FLAGS = [
    ("--input", str, "graph/karate.edgelist", "input edgelist"),
    ("--output", str, "emb/karate.emb", "where the word2vec-text embeddings go"),
    ("--dimensions", int, 128, "embedding width d"),
    ("--walk-length", int, 80, "nodes per walk"),
    ("--num-walks", int, 10, "walks per start node"),
    ("--window-size", int, 10, "skip-gram window"),
    ("--iter", int, 1, "SGD epochs"),
    ("--workers", int, 8, "accepted for compatibility; the GPU replaces gensim's worker threads"),
    ("--p", float, 1, "return parameter"),
    ("--q", float, 1, "in-out parameter"),
]
_SWITCHES = [("weighted", "unweighted"), ("directed", "undirected")]   # default: the second of each pair


def parse_args(argv=None):
    """The reference's command line (src/main.py:18-64) plus --rng / --seed."""
    ap = argparse.ArgumentParser(description="node2vec on MI355X")
    for flag, typ, default, doc in _FLAGS:
        ap.add_argument(flag, type=typ, default=default, nargs="?" if typ is str else None, help=doc)
    for on, off in _SWITCHES:
        ap.add_argument("--" + on, dest=on, action="store_true")
        ap.add_argument("--" + off, dest=off, action="store_false")
        ap.set_defaults(**{on: False})
    ap.add_argument("--rng", default="numpy", choices=["numpy", "philox"],
                    help="numpy: consume numpy's global MT19937 stream like the reference; philox: in-kernel RNG")
    ap.add_argument("--seed", type=int, default=1, help="seed of the philox walk RNG and of the SGNS trainer")
    ap.add_argument("--merge", default="tsum", choices=["tsum", "hot"],
                    help="more than one GPU: how the ranks' replicas are merged (n2v_hip/merge.py: tiered pure sums, or "
                         "weighted sums — faster, but 0.005-0.006 AUC below the sequential comparator at 131k nodes: refused "
                         "above 32 768 nodes without --allow-out-of-band)")
    ap.add_argument("--allow-out-of-band", action="store_true",
                    help="permit modes measured OUTSIDE the +-0.002 link-prediction AUC band (merge=hot on large graphs, "
                         "lossy update modes on short corpora, shared negatives)")
    return ap.parse_args(argv)


def read_graph():
    '''
    Reads the input network (src/main.py:66-80) straight into the sorted-CSR container; node
    order, duplicate lines and the to_undirected() weight rule follow networkx.
    '''
    return _csr.read_edgelist(args.input, weighted=args.weighted, directed=args.directed)


def learn_embeddings(walks, **overrides):
    '''
    Learn embeddings by optimizing the Skipgram objective using SGD (src/main.py:82-90):
    Word2Vec(walks, size=args.dimensions, window=args.window_size, min_count=0, sg=1,
    workers=args.workers, iter=args.iter) with gensim's defaults for everything else.
    `walks` is what simulate_walks returned (a WalkCorpus: stays on the device) or any list
    of lists of node ids.
    '''
    a = args
    dim = overrides.get("dimensions", getattr(a, "dimensions", 128))
    window = overrides.get("window_size", getattr(a, "window_size", 10))
    epochs = overrides.get("iter", getattr(a, "iter", 1))
    seed = overrides.get("seed", getattr(a, "seed", 1))
    corpus = node2vec.as_corpus(walks)
    model = _sgns.SgnsModel(len(corpus.labels), dim=dim, window=window, negative=overrides.get("negative", 5),
                            alpha=overrides.get("alpha", 0.025), min_alpha=overrides.get("min_alpha", 1e-4),
                            sample=overrides.get("sample", 1e-3), seed=seed, device=corpus.walks.device,
                            update_mode=overrides.get("update_mode", "auto"),
                            share_negatives=overrides.get("share_negatives", False),
                            allow_out_of_band=overrides.get("allow_out_of_band", getattr(a, "allow_out_of_band", False)))
    ctx = overrides.get("ctx")
    if ctx is None or ctx.world == 1:
        model.build_vocab(corpus.walks)
        _sgns.train(model, corpus.walks, corpus.lens, epochs=epochs)
    else:
        # one process per GPU: `walks` is this rank's shard (Graph.simulate_walks_shard); word counts
        # and the learning-rate schedule are global, the replicas are merged over RCCL
        from n2v_hip import dist as _dist
        model.build_vocab(counts=_dist.global_counts(corpus.walks, len(corpus.labels), ctx))
        assert model.device == ctx.device, "replica on %s but this rank owns %s" % (model.device, ctx.device)
        n_local = int(corpus.walks.shape[0])
        tot = torch.tensor([n_local], dtype=torch.int64, device=corpus.walks.device)
        ctx.comm.all_reduce_sum(tot)
        b, _ = _sgns.shard_bounds(overrides["n_starts"], ctx.world, ctx.rank)
        _dist.train_sharded(model, corpus.walks, corpus.lens, ctx, n_walks_global=int(tot.item()),
                            shard_offset=b * overrides["num_walks"], epochs=epochs, merge=overrides.get("merge", "tsum"))
    wv = _sgns.KeyedVectors(corpus.labels, model.counts, model.vectors().cpu().numpy())
    return _sgns.Word2VecResult(wv, model, model.pairs_trained())


def main(args_):
    '''
    Pipeline for representational learning for all nodes in a graph (src/main.py:92-101).
    '''
    global args
    args = args_
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx = None
    if world > 1:
        # one rank per GPU: bind this process to ITS device before anything is allocated, so graph, tables,
        # walks and both embedding tables live on cuda:LOCAL_RANK (not all on cuda:0)
        from n2v_hip import dist as _dist
        ctx = _dist.RankContext()
    nx_G = read_graph()
    G = node2vec.Graph(nx_G, args.directed, args.p, args.q, rng=getattr(args, "rng", "numpy"),
                       seed=getattr(args, "seed", 1), device=None if ctx is None else ctx.device)
    G.preprocess_transition_probs()
    if ctx is None:
        walks = G.simulate_walks(args.num_walks, args.walk_length)
        return learn_embeddings(walks)
    # launched by torch.distributed.run: walks shard by start vertex, every rank ends with the same merged
    # embedding (BASELINE config C4)
    assert G._engine.device == ctx.device, (G._engine.device, ctx.device)
    walks = G.simulate_walks_shard(args.num_walks, args.walk_length, ctx.rank, ctx.world)
    emb = learn_embeddings(walks, ctx=ctx, n_starts=G._csr.n_nodes, num_walks=args.num_walks,
                           merge=getattr(args, "merge", "tsum"))
    ctx.barrier()
    return emb


def save_embeddings(emb, path):
    """src/main.py:88 writes args.output unconditionally; the directory is created if it is missing (the
    reference would raise IOError after the whole run)."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    emb.wv.save_word2vec_format(path)


if __name__ == "__main__":
    args = parse_args()
    emb = main(args)
    if int(os.environ.get("RANK", "0")) == 0 and args.output:
        save_embeddings(emb, args.output)
