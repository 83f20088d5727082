"""Synthetic episodic loader with the reference loader's batch contract (fumi/dataset/data.py:571-581 + torchmeta
collate):

    batch = {'train': ([idx i64 [B,S], text f32 [B,S,Dt] | i64 [B,S,L], im f32 [B,S,D]], targets i64 [B,S]),
             'test' : same with Qn = N * num_shots_test rows}

The iNat-Anim files (Zenodo JSON + HDF5 embeddings, BERT/GloVe downloads) cannot be fetched here, so
``--dataset synthetic`` draws a learnable task instead: class prototypes mu_c ~ N(0, I_D), images x = mu_c + 2 eps,
class text = P mu_c + 0.5 eps with a fixed random P [Dt, D] (the text is identical for all samples of a class, like
data.py:543-549).  Deterministic in (seed, split, batch index); every rank regenerates the SAME meta-batch."""
import numpy as np
import torch


class SyntheticEpisodes:
    def __init__(self, n_classes, D, Dt, N, K, Q, batch_size, seed, split, tokens=None, length=None, image_shape=None):
        """image_shape=(C, H, W): the samples are raw images (--im_encoder conv4) -- the class prototype is a D = C*H*W pattern
        and a sample is prototype + noise, reshaped to [C, H, W]."""
        rs = np.random.RandomState(seed * 7 + {"train": 0, "val": 1, "test": 2}[split])
        self.image_shape = image_shape
        if image_shape is not None:
            D = int(np.prod(image_shape))
        self.mu = rs.standard_normal((n_classes, D)).astype(np.float32)
        proj = np.random.RandomState(seed + 99).standard_normal((Dt if tokens is None else 16, D)).astype(np.float32) / np.sqrt(D)
        self.N, self.K, self.Q, self.B, self.D = N, K, Q, batch_size, D
        self.tokens = tokens
        if tokens is None:
            self.text = (self.mu @ proj.T + 0.5 * rs.standard_normal((n_classes, Dt))).astype(np.float32)
        else:
            V, L, pad = tokens
            self.text = np.full((n_classes, L), pad, dtype=np.int64)
            for c in range(n_classes):
                ln = rs.randint(4, L + 1)
                self.text[c, :ln] = rs.randint(1, V, size=ln)
        self.seed, self.split, self.length = seed, split, length

    def batch(self, index):
        rs = np.random.RandomState((self.seed * 1000003 + index * 31 + len(self.split)) % (2 ** 31))
        B, N, K, Q = self.B, self.N, self.K, self.Q
        S, Qn = N * K, N * Q
        cls = np.stack([rs.choice(len(self.mu), N, replace=False) for _ in range(B)])          # [B,N]
        y_s = np.stack([rs.permutation(np.repeat(np.arange(N), K)) for _ in range(B)])
        y_q = np.stack([rs.permutation(np.repeat(np.arange(N), Q)) for _ in range(B)])
        cs, cq = np.take_along_axis(cls, y_s, 1), np.take_along_axis(cls, y_q, 1)
        x_s = self.mu[cs] + 2.0 * rs.standard_normal((B, S, self.D)).astype(np.float32)
        x_q = self.mu[cq] + 2.0 * rs.standard_normal((B, Qn, self.D)).astype(np.float32)
        if self.image_shape is not None:
            x_s, x_q = x_s.reshape(B, S, *self.image_shape), x_q.reshape(B, Qn, *self.image_shape)
        t = torch.from_numpy
        return {'train': ([t(cs.astype(np.int64)), t(self.text[cs]), t(x_s.astype(np.float32))], t(y_s.astype(np.int64))),
                'test': ([t(cq.astype(np.int64)), t(self.text[cq]), t(x_q.astype(np.float32))], t(y_q.astype(np.int64)))}

    def __iter__(self):
        i = 0
        while self.length is None or i < self.length:
            yield self.batch(i)
            i += 1


def get_synthetic(args):
    """(train_loader, val_loader, test_loader, dictionary) like fumi/dataset/data.py:25-86."""
    tokens, dictionary = None, None
    if args.text_encoder in ("glove", "w2v"):
        V, L = args.synthetic_vocab, args.synthetic_seq_len
        tokens = (V, L, 0)
        dictionary = {"PAD": 0}
        dictionary.update({f"tok{i}": i for i in range(1, V)})
    shape = (args.image_channels, args.image_size, args.image_size) if getattr(args, "im_encoder", "") in ("conv4", "resnet12") else None
    mk = lambda split, q: SyntheticEpisodes(args.synthetic_classes, args.im_emb_dim, args.text_emb_dim, args.num_ways,
                                            args.num_shots, q, args.batch_size, args.seed, split, tokens, image_shape=shape)
    q_eval = int(100 / args.num_ways)                      # data.py:163-166,180-183
    return mk("train", args.num_shots_test), mk("val", q_eval), mk("test", q_eval), dictionary


def get_synthetic_resident(args, images_per_class=48):
    """``--dataset synthetic-resident``: the same learnable task family as ``synthetic``, but as a FIXED table of image
    embeddings per split (n_classes x images_per_class rows) that lives in HBM and is sampled by the GPU-resident episode
    sampler (fumi_amd/dataset/gpu_sampler.py) -- the shape of the real pipeline (precomputed embeddings + class descriptions)
    without the files.  Needs a GPU."""
    from .gpu_sampler import GpuEpisodeSampler
    tokens, dictionary = None, None
    if args.text_encoder in ("glove", "w2v"):
        V, L = args.synthetic_vocab, args.synthetic_seq_len
        tokens = (V, L, 0)
        dictionary = {"PAD": 0}
        dictionary.update({f"tok{i}": i for i in range(1, V)})
    q_eval = int(100 / args.num_ways)
    per = max(images_per_class, args.num_shots + max(args.num_shots_test, q_eval))

    def mk(split, q):
        base = SyntheticEpisodes(args.synthetic_classes, args.im_emb_dim, args.text_emb_dim, args.num_ways, args.num_shots, q,
                                 args.batch_size, args.seed, split, tokens)
        rs = np.random.RandomState(args.seed * 13 + len(split))
        coi = np.repeat(np.arange(args.synthetic_classes), per)
        images = base.mu[coi] + 2.0 * rs.standard_normal((len(coi), args.im_emb_dim)).astype(np.float32)
        return GpuEpisodeSampler(torch.from_numpy(images.astype(np.float32)).to(args.device), coi, torch.from_numpy(base.text),
                                 args.num_ways, args.num_shots, q, args.batch_size, seed=args.seed + len(split))
    return mk("train", args.num_shots_test), mk("val", q_eval), mk("test", q_eval), dictionary


class SyntheticSupervised:
    """Supervised (image embedding, class text embedding, class id) mini-batches with the item contract of the reference's
    SupervisedInatAnim + DataLoader (fumi/dataset/data.py:54-70,231-291): batch = [images [bs, D], text [bs, Dt], ids [bs]],
    shuffled every epoch.  Same learnable task family as SyntheticEpisodes."""

    def __init__(self, n_classes, per_class, D, Dt, batch_size, seed, split):
        base = SyntheticEpisodes(n_classes, D, Dt, 1, 1, 1, 1, seed, split)
        rs = np.random.RandomState(seed * 17 + len(split))
        self.ids = np.repeat(np.arange(n_classes), per_class)
        self.images = torch.from_numpy((base.mu[self.ids] + 2.0 * rs.standard_normal((len(self.ids), D))).astype(np.float32))
        self.text = torch.from_numpy(base.text)
        self.bs, self.rs = batch_size, rs

    def __iter__(self):
        perm = self.rs.permutation(len(self.ids))
        for i in range(0, len(perm), self.bs):
            idx = perm[i:i + self.bs]
            yield [self.images[idx], self.text[self.ids[idx]], torch.from_numpy(self.ids[idx])]


def get_synthetic_supervised(args, per_class=6):
    mk = lambda split: SyntheticSupervised(args.synthetic_classes, per_class, args.im_emb_dim, args.text_emb_dim, args.batch_size,
                                           args.seed, split)
    return mk("train"), mk("val"), mk("test"), {}
