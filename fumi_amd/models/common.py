"""Text encoders (host-side mirror of fumi/models/common.py).

``WordEmbedding`` (common.py:8-41) keeps its constructor, attributes (``embed``, ``embedding_dim``, ``padding_token``)
and ``state_dict`` key (``embed.weight``); its forward is the HIP embedding-bag kernel (csrc/glove.hip).
The bi-LSTM encoders ``RNN`` / ``RnnHid`` (common.py:44-161) are outside the hot path named by BASELINE.json
 and raise.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import engine as _engine


_WORD_MODELS = {}


class ArrayKeyedVectors:
    """Minimal KeyedVectors-like table over an in-memory array (``words[i]`` <-> ``vectors[i]``)."""

    def __init__(self, words, vectors):
        self.vector_size = int(vectors.shape[1])
        self.key_to_index = {w: i for i, w in enumerate(words)}
        self._v = vectors

    def __getitem__(self, w):
        return self._v[self.key_to_index[w]]


def register_word_vectors(text_encoder_type, keyed_vectors):
    """Supply the pretrained table for 'glove' / 'w2v' when gensim's downloader is unavailable (no network)."""
    _WORD_MODELS[text_encoder_type] = keyed_vectors


def get_embedding_weights(dictionary, text_encoder_type, word_model=None):
    """common.py:164-196: rows of the pretrained table for the dictionary's tokens; OOV rows ~ U(-1,1); PAD row = 0.
    The reference downloads the table with gensim (network); here the KeyedVectors-like object must be supplied
    (``word_model``: ``vector_size``, ``key_to_index``, ``__getitem__``) or gensim must be installed."""
    if word_model is None:
        word_model = _WORD_MODELS.get(text_encoder_type)
    if word_model is None:
        try:
            import gensim.downloader as api
        except Exception as e:   # ModuleNotFoundError in this image
            raise RuntimeError("gensim is not installed and no table was registered: call "
                               "fumi_amd.models.common.register_word_vectors(name, keyed_vectors) first") from e
        word_model = api.load({"glove": "glove-wiki-gigaword-300", "w2v": "word2vec-google-news-300"}[text_encoder_type])
    dim = word_model.vector_size
    weights = 2 * np.random.rand(len(dictionary), dim) - 1
    for word, token in dictionary.items():
        if word == "PAD":
            weights[token, :] = 0.0
        elif word in word_model.key_to_index:
            weights[token, :] = word_model[word]
    return weights


class WordEmbedding(nn.Module):
    def __init__(self, text_encoder_type, pooling_strat, dictionary, word_model=None, weights=None):
        super().__init__()
        self.pooling_strat = pooling_strat
        self.dictionary = dictionary
        self.padding_token = self.dictionary["PAD"]
        self.text_encoder_type = text_encoder_type
        if weights is None:
            weights = get_embedding_weights(dictionary, text_encoder_type, word_model)
        self.embed = nn.Embedding.from_pretrained(torch.as_tensor(np.asarray(weights), dtype=torch.float32))
        self.embedding_dim = int(self.embed.weight.shape[-1])

    def forward(self, x):
        """x: int64 tokens (b, N*K, L) -> (b, N*K, E): masked mean (sum / #non-PAD) or max over all positions."""
        if self.pooling_strat not in ("mean", "max"):
            raise NameError(f"{self.pooling_strat} pooling strat not defined")
        return _engine.get_engine().glove_bag(x.contiguous(), self.embed.weight, self.padding_token, self.pooling_strat)


class RNN(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("bi-LSTM text encoders (fumi/models/common.py:44-161) are outside the MI355X hot path")


RnnHid = RNN
