"""Text encoders (host-side mirror of fumi/models/common.py).

``WordEmbedding`` (common.py:8-41) keeps its constructor, attributes (``embed``, ``embedding_dim``, ``padding_token``)
and ``state_dict`` key (``embed.weight``); its forward is the HIP embedding-bag kernel (csrc/glove.hip).
The bi-LSTM encoders ``RNN`` / ``RnnHid`` (common.py:44-161) keep their constructors, attributes and ``state_dict`` keys
(``embed.weight``, ``rnn.weight_ih_l0`` ...); their forward is the engine's bidirectional-LSTM op (csrc/textenc.hip).  Like
every text encoder they are frozen unless ``--fine_tune`` (fumi.py:65-67); under ``--fine_tune`` the LSTM is trained through
``forward_train`` / ``backward`` (the engine's tape-keeping forward and back-propagation through time), which FUMI.evaluate
drives with the text adjoint its meta-step returns.
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


class _BiLstmEncoder(nn.Module):
    """Shared body of RNN / RnnHid (common.py:44-75,110-138): word embedding + single-layer bidirectional LSTM."""
    use_cell = False

    def __init__(self, embedding_type, pooling_strat, dictionary, rnn_hid_dim, word_model=None):
        super().__init__()
        self.pooling_strat = pooling_strat
        self.dictionary = dictionary
        self.embedding_type = embedding_type
        self.rnn_hid_dim = rnn_hid_dim // 2      # assuming bidirectional
        self.padding_token = self.dictionary["PAD"]
        if embedding_type == "rand":
            self.embed = nn.Embedding(len(self.dictionary), rnn_hid_dim)
            self.text_emb_size = rnn_hid_dim
        else:
            weights = get_embedding_weights(dictionary, embedding_type, word_model)
            self.text_emb_size = weights.shape[-1]
            self.embed = nn.Embedding.from_pretrained(torch.as_tensor(np.asarray(weights), dtype=torch.float32))
        self.rnn = nn.LSTM(input_size=self.text_emb_size, hidden_size=self.rnn_hid_dim, num_layers=1, bidirectional=True,
                           batch_first=True)

    def lstm_weights(self):
        names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
        return [getattr(self.rnn, n + suffix) for suffix in ("", "_reverse") for n in names]

    def forward(self, x):
        """x: int64 tokens (b, N*K, L) -> (b, N*K, rnn_hid_dim): the forward direction's state at the last real token and the
        backward direction's at token 0 (common.py:98-107 / :154-161), i.e. each direction's final state."""
        if self.trainable() and torch.is_grad_enabled() and self.training:
            raise NotImplementedError("a trainable bi-LSTM text encoder (--fine_tune with RNN / RNNhid) is trained through "
                                      "forward_train / backward (FUMI.evaluate does); this forward is the frozen encoder's")
        return _engine.get_engine().lstm_bidir(x.contiguous(), self.embed.weight.detach(),
                                               [w.detach().contiguous() for w in self.lstm_weights()], self.padding_token,
                                               self.use_cell)

    def trainable(self):
        """--fine_tune leaves the LSTM's parameters trainable (fumi.py:65-67); the pretrained word table stays frozen
        (nn.Embedding.from_pretrained, common.py:60-63)."""
        return any(p.requires_grad for p in self.rnn.parameters())

    def forward_train(self, x):
        """(out, tape): forward that keeps what `backward` needs (the engine's tape, csrc/textenc.hip)."""
        if self.embed.weight.requires_grad:
            raise NotImplementedError("embedding_type='rand' with a trainable word table is not supported by the engine "
                                      "(the models build RNN / RnnHid on the frozen pretrained table)")
        return _engine.get_engine().lstm_bidir_train(x.contiguous(), self.embed.weight.detach(),
                                                     [w.detach().contiguous() for w in self.lstm_weights()], self.padding_token,
                                                     self.use_cell)

    def backward(self, x, tape, d_out):
        """Back-propagation through time: leaves d_out's pull-back in .grad of the eight LSTM tensors (what loss.backward() does
        for the reference's trainable encoder).  Returns the gradient tensors."""
        ws = self.lstm_weights()
        gs = _engine.get_engine().lstm_bidir_bwd(x.contiguous(), self.embed.weight.detach(), [w.detach().contiguous() for w in ws],
                                                 self.padding_token, self.use_cell, tape, d_out.contiguous())
        for w, g in zip(ws, gs):
            w.grad = g
        return gs


class RNN(_BiLstmEncoder):
    """common.py:44-107: the LSTM's OUTPUT states (= final hidden state h_n of each direction)."""
    use_cell = False


class RnnHid(_BiLstmEncoder):
    """common.py:110-161: the LSTM's final CELL states c_n of each direction."""
    use_cell = True
