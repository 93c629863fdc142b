"""CPU tests of the walk / embedding text formats."""
import numpy as np

from n2v_hip import io as n2v_io
from n2v_hip.sgns import KeyedVectors


def test_walk_text_round_trip(tmp_path):
    walks = [[1, 32, 22, 1], [5], [99999990001, 3, 3]]
    p = tmp_path / "walks.txt"
    n2v_io.save_walks(walks, str(p))
    assert p.read_text() == "1 32 22 1\n5\n99999990001 3 3\n"   # src/main_link.py:545-546 format
    assert n2v_io.load_walks(str(p)) == walks


def test_walk_text_from_corpus_like(tmp_path):
    import torch

    class Corpus:  # what node2vec.WalkCorpus exposes (CPU tensors stand in for device ones)
        walks = torch.tensor([[0, 2, 1], [1, -1, -1], [2, 0, -1]], dtype=torch.int32)
        lens = torch.tensor([3, 1, 2], dtype=torch.int32)
        labels = np.array([10, 20, 30], dtype=np.int64)
    p = tmp_path / "w.txt"
    n2v_io.save_walks(Corpus(), str(p))
    assert n2v_io.load_walks(str(p)) == [[10, 30, 20], [20], [30, 10]]
    Corpus.walks = torch.tensor([[0, 2], [1, 1]], dtype=torch.int32)
    Corpus.lens = torch.tensor([2, 2], dtype=torch.int32)
    n2v_io.save_walks(Corpus(), str(p))
    assert p.read_text() == "10 30\n20 20\n"


def test_word2vec_text_round_trip(tmp_path):
    labels = np.array([7, 99999991, 12], dtype=np.int64)
    counts = np.array([5, 9, 0], dtype=np.int64)           # id 12 never appeared: not in the vocabulary
    vecs = np.arange(12, dtype=np.float32).reshape(3, 4) / 8
    kv = KeyedVectors(labels, counts, vecs)
    assert kv.index2word == ["99999991", "7"]               # descending count, like gensim's sorted vocab
    p = tmp_path / "e.emb"
    kv.save_word2vec_format(str(p))
    words, got = n2v_io.load_word2vec_format(str(p))
    assert words == ["99999991", "7"] and np.allclose(got, vecs[[1, 0]], atol=1e-6)
    d = n2v_io.emb_file_to_dict(str(p), skip_prefix="9999999")   # src/utils.py:417-426 skips item ids
    assert list(d) == ["7"] and np.allclose(d["7"], vecs[0], atol=1e-6)
    assert abs(kv.similarity("7", "99999991") - float(np.dot(vecs[0], vecs[1]) / np.linalg.norm(vecs[0]) / np.linalg.norm(vecs[1]))) < 1e-6
