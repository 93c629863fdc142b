"""File formats either side of the hot path.

* walk text: one walk per line, node ids separated by one space — what src/main_link.py:544-546
  writes and gensim's LineSentence reads back (src/main_link.py:345-349, ``-walk-path``);
* word2vec text embeddings: header ``N d`` then ``id v1 ... vd`` (gensim's
  save_word2vec_format, src/main.py:88), read back the way src/utils.py:417-426 does.
"""
import numpy as np


def save_walks(walks, path, chunk=200000):
    """`walks`: a WalkCorpus (device) or any sequence of sequences of node ids."""
    w = getattr(walks, "walks", None)
    if w is None:
        with open(path, "w") as f:
            for walk in walks:
                f.write(" ".join(map(str, walk)) + "\n")
        return
    labels = walks.labels
    n = int(w.shape[0])
    with open(path, "w") as f:
        for b in range(0, n, chunk):
            rows = w[b:b + chunk].cpu().numpy()
            lens = walks.lens[b:b + chunk].cpu().numpy()
            lab = labels[np.maximum(rows, 0)]
            if (lens == rows.shape[1]).all():
                np.savetxt(f, lab, fmt="%d", delimiter=" ")
            else:
                for r, k in zip(lab, lens):
                    f.write(" ".join(map(str, r[:k].tolist())) + "\n")


def load_walks(path):
    """List of lists of int node ids (gensim LineSentence splits on whitespace)."""
    out = []
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if tok:
                out.append([int(t) for t in tok])
    return out


def load_word2vec_format(path):
    """-> (words: list[str], vectors: float32 [N, d]) from the text format."""
    with open(path, "r") as f:
        lines = f.read().splitlines()
    n, d = (int(x) for x in lines[0].split())
    words, vecs = [], np.empty((n, d), dtype=np.float32)
    for i, line in enumerate(lines[1:n + 1]):
        tok = line.split()
        words.append(tok[0])
        vecs[i] = [float(x) for x in tok[1:d + 1]]
    return words, vecs


def emb_file_to_dict(path, skip_prefix=None):
    """src/utils.py:417-426 (emb_file_to_user_dict): {id string: list of floats}, optionally
    skipping ids with a prefix (the reference skips item nodes, prefix '9999999')."""
    words, vecs = load_word2vec_format(path)
    return {w: [float(x) for x in v] for w, v in zip(words, vecs) if not (skip_prefix and w.startswith(skip_prefix))}
