"""iNat-Anim on-disk formats -> HBM-resident episode tables.

What the reference reads (fumi/dataset/data.py) and what this module does with it:

* ``<data_dir>/iNat-Anim/inat_anim.json`` (data.py:373-375): ``categories[i].{name, common_name, description}``,
  ``images[i].id``, ``annotations[image_id].category_id``.  Parsed once per process.
* class split (data.py:316-318,377-388): ``np.random.seed(0)``; shuffle ``arange(N)``; 60 / 20 / 20 % slices.  Every split is
  built after re-seeding, so all three see the same permutation.  ``class_split`` draws it from a private
  ``RandomState(0)`` (same MT19937 stream as the legacy global seeding, without touching global state).  The reference's
  ``np.sort(self.categories)`` (data.py:390) discards its result: classes stay in shuffled order, and so do they here.
* image lists (data.py:392-410): images in file order whose category is in the split; the embedding row of an image is its
  id (data.py:545: ``self.image_embeddings[indices]`` with image ids).
* ``image_embeddings_{resnet-152,resnet-34}.hdf5['images']`` (data.py:420-430): read with h5py when it is installed;
  otherwise from ``image_embeddings_<model>.npy`` next to it (``tools/convert_embeddings.py`` writes it on a machine that
  has h5py).  Only the rows of the split are kept and they go to HBM once.
* text (data.py:412-414,497-512): per class the chosen fields joined by a blank.  The reference passes the fields as a
  ``set`` of enum members, whose iteration order follows the per-process string hash; here the order is fixed to
  description, label, common_name.  Stop words (data.py:432-439) are removed from the whitespace-split raw text,
  case-sensitively, from nltk's English list when nltk has it, else from ``<root>/stopwords_english.txt`` (one per line).
* STANDARD tokenisation for glove / w2v (data.py:449-467): every description lower-cased and padded with ``<PAD>`` words to
  the longest (length counted in tokens), the dictionary built over the descriptions of ALL categories (no stop-word
  removal there) in category order plus the PAD document, ids from ``gensim.corpora.Dictionary``.  gensim is not installed
  here: ``tokenize`` and ``Dictionary`` are restated from the published behaviour of gensim 3.8 ([memory]: alphabetic runs
  ``(((?![\\d])\\w)+)``; new ids per document in sorted token order) -- "parity unpinned" against the real package, and used
  only if gensim itself is missing.
* BERT precompute (data.py:441-448,470-495 -- row f3): ``BertTokenizer`` with ``padding=True, truncation=True``, batches of
  64 through ``BertModel`` and the mean over ALL positions of ``last_hidden_state`` -- padding included, no mask in the
  mean, as the reference does.  Weights come from ``$FUMI_BERT_PATH`` (a local directory) or the Hugging Face cache; there
  is no network here, so absent weights raise.

The result per split is a ``GpuEpisodeSampler`` (fumi_amd/dataset/gpu_sampler.py) over the split's rows: the torchmeta
loader of the reference re-reads whole classes from HDF5 on the training thread for every task (data.py:533-549).
Sampling differs from torchmeta's by construction (counter-based hash instead of Python's ``random``): DESIGN.md section 8."""
import json
import os
import re

import numpy as np
import torch

TEXT_FIELDS = (("description", "description"), ("label", "name"), ("common_name", "common_name"))    # data.py:96-100,499-503
_ALPHABETIC = re.compile(r"(((?![\d])\w)+)", re.UNICODE)


def load_annotations(json_path):
    with open(json_path) as f:
        return json.load(f)


def class_split(n_categories, split):
    """Category ids of a meta-split in the reference's (shuffled) order."""
    cats = np.arange(n_categories)
    np.random.RandomState(0).shuffle(cats)
    a, b = int(0.6 * n_categories), int(0.8 * n_categories)
    if split == "train":
        return cats[:a]
    if split == "val":
        return cats[a:b]
    if split == "test":
        return cats[b:]
    raise ValueError("One of meta_train, meta_val, meta_test must be true")


def split_images(annotations, categories):
    """(image_ids, class_of_image): images in file order whose category is in ``categories``; the class of an image is the
    position of its category in ``categories`` (the index torchmeta's ClassDataset hands to __getitem__)."""
    pos = {int(c): i for i, c in enumerate(categories)}
    ann = annotations["annotations"]
    ids, cls = [], []
    for im in annotations["images"]:
        c = ann[im["id"]]["category_id"]
        if c in pos:
            ids.append(im["id"])
            cls.append(pos[c])
    return np.asarray(ids, dtype=np.int64), np.asarray(cls, dtype=np.int64)


def descriptions(annotations, categories, text_type):
    unknown = [t for t in text_type if t not in dict(TEXT_FIELDS)]
    if unknown:
        raise NameError("Invalid text type used")
    keys = [k for name, k in TEXT_FIELDS if name in text_type]
    return [" ".join(annotations["categories"][int(i)][k] for k in keys) for i in categories]


def english_stop_words(root):
    try:
        from nltk.corpus import stopwords
        return stopwords.words("english")
    except Exception:
        path = os.path.join(root, "stopwords_english.txt")
        if not os.path.exists(path):
            raise FileNotFoundError("--remove_stop_words needs nltk's English stop-word list; nltk (or its corpus) is not "
                                    f"available and {path} (one word per line) does not exist")
        with open(path) as f:
            return [w.strip() for w in f if w.strip()]


def strip_stop_words(texts, stop_words):
    stop = set(stop_words)
    return [" ".join(w for w in s.split() if w not in stop) for s in texts]


def tokenize(text):
    try:
        from gensim.utils import tokenize as g_tokenize
        return list(g_tokenize(text))
    except ImportError:
        return [m.group() for m in _ALPHABETIC.finditer(text)]


def build_dictionary(documents):
    """token2id of gensim.corpora.Dictionary(documents): unseen tokens of a document get the next ids in sorted order."""
    try:
        from gensim import corpora
        return dict(corpora.Dictionary(documents).token2id)
    except ImportError:
        token2id = {}
        for doc in documents:
            for w in sorted(set(doc) - token2id.keys()):
                token2id[w] = len(token2id)
        return token2id


def standard_tokens(split_texts, all_texts):
    """(tokens int64 [C, L], token2id) for the glove / w2v encoders."""
    lengths = [len(tokenize(d)) for d in split_texts]
    L = max(lengths)
    padded = [d.lower() + " " + " ".join("<PAD>" for _ in range(L - n)) for d, n in zip(split_texts, lengths)]
    token2id = build_dictionary([tokenize(d.lower()) for d in all_texts] + [tokenize("<PAD>")])
    return torch.tensor([[token2id[z] for z in tokenize(d)] for d in padded], dtype=torch.int64), token2id


def bert_embeddings(texts, device=None, model_path=None, batch_size=64):
    """[C, hidden] fp32 on the CPU: mean over all token positions (padding included) of BERT's last hidden state."""
    from transformers import BertModel, BertTokenizer
    path = model_path or os.environ.get("FUMI_BERT_PATH", "bert-base-uncased")
    try:
        tokenizer = BertTokenizer.from_pretrained(path)
        model = BertModel.from_pretrained(path)
    except Exception as e:
        raise FileNotFoundError(f"BERT weights / vocabulary not found at '{path}' (no network here): point FUMI_BERT_PATH at a "
                                "local copy of bert-base-uncased") from e
    model.eval()
    if device is not None:
        model.to(device)
    tok = tokenizer(list(texts), return_token_type_ids=False, return_tensors="pt", padding=True, truncation=True)
    ids, mask = tok["input_ids"], tok["attention_mask"]
    out = torch.zeros(len(texts), model.config.hidden_size)
    for s in range(0, len(texts), batch_size):
        e = min(len(texts), s + batch_size)
        with torch.no_grad():
            d, m = (ids[s:e].to(device), mask[s:e].to(device)) if device is not None else (ids[s:e], mask[s:e])
            out[s:e] = torch.mean(model(input_ids=d, attention_mask=m, output_attentions=False).last_hidden_state, dim=1).cpu()
    return out


def open_embeddings(root, image_embedding_model):
    """Array-like [n_images, D] over the embedding file (rows are read lazily)."""
    stem = os.path.join(root, f"image_embeddings_{image_embedding_model}")
    if os.path.exists(stem + ".hdf5"):
        try:
            import h5py
            return h5py.File(stem + ".hdf5", "r")["images"]
        except ImportError:
            if not os.path.exists(stem + ".npy"):
                raise FileNotFoundError(f"{stem}.hdf5 needs h5py, which is not installed; convert it once with "
                                        f"tools/convert_embeddings.py to {stem}.npy")
    if os.path.exists(stem + ".npy"):
        return np.load(stem + ".npy", mmap_mode="r")
    raise FileNotFoundError(f"no image embeddings at {stem}.hdf5 / {stem}.npy")


def read_rows(emb, ids):
    """Rows ``ids`` (any order, repeats allowed) of an h5py dataset or array, as fp32."""
    uniq, inv = np.unique(ids, return_inverse=True)              # h5py wants increasing, distinct indices
    return np.asarray(emb[uniq], dtype=np.float32)[inv]


class InatAnimSplit:
    """The host-side description of one meta-split: categories, image lists, text rows (everything but the GPU tables)."""

    def __init__(self, root, json_path, split, text_encoder, text_type, remove_stop_words=False, annotations=None,
                 device=None, bert_path=None):
        if root not in json_path:
            json_path = os.path.join(root, json_path)
        ann = annotations if annotations is not None else load_annotations(json_path)
        n = len(ann["categories"])
        self.categories = class_split(n, split)
        self.image_ids, self.class_of_image = split_images(ann, self.categories)
        texts = descriptions(ann, self.categories, text_type)
        if remove_stop_words:
            texts = strip_stop_words(texts, english_stop_words(root))
        self.texts = texts
        self.dictionary = None
        if text_encoder == "BERT":
            self.class_text = bert_embeddings(texts, device, bert_path)
        else:
            self.class_text, self.dictionary = standard_tokens(texts, descriptions(ann, np.arange(n), text_type))


def get_inat_anim(args):
    """``--dataset inat-anim`` (data.py:125-186): (train, val, test, dictionary) with the loaders replaced by GPU samplers."""
    from .gpu_sampler import GpuEpisodeSampler
    root = args.data_dir + "/iNat-Anim"
    json_path = root + "/inat_anim.json"
    if not os.path.exists(json_path):
        raise FileNotFoundError(f"{json_path} not found: --dataset inat-anim needs the iNat-Anim files under {root} "
                                "(or use --dataset synthetic / synthetic-resident, same batch layout)")
    if args.device.type != "cuda":
        raise RuntimeError("--dataset inat-anim keeps the embedding table in HBM and samples on the GPU: no GPU visible")
    ann = load_annotations(json_path)
    emb = open_embeddings(root, args.image_embedding_model)
    q_eval = int(100 / args.num_ways)                                                   # data.py:163-166,180-183
    out, train_dictionary = [], None
    for split, q in (("train", args.num_shots_test), ("val", q_eval), ("test", q_eval)):
        sp = InatAnimSplit(root, json_path, split, args.text_encoder, args.text_type, args.remove_stop_words, ann, args.device)
        if split == "train":
            train_dictionary = sp.dictionary
        images = torch.from_numpy(read_rows(emb, sp.image_ids)).to(args.device)
        out.append(GpuEpisodeSampler(images, sp.class_of_image, sp.class_text, args.num_ways, args.num_shots, q,
                                     args.batch_size, seed=args.seed + len(split), row_ids=sp.image_ids, torchmeta_tasks=True))
    return out[0], out[1], out[2], ({} if args.text_encoder == "BERT" else train_dictionary)


class SupervisedSplit:
    """``--dataset supervised-inat-anim`` (data.py:54-70,231-291): items (image embedding, mean-pooled BERT embedding of the
    class description, category id), shuffled mini-batches like DataLoader(shuffle=True)."""

    def __init__(self, images, class_of_image, class_text, category_ids, batch_size, seed):
        self.images, self.coi, self.text = images, np.asarray(class_of_image), class_text
        self.cat = np.asarray(category_ids)
        self.bs, self.rs = batch_size, np.random.RandomState(seed)

    def __len__(self):
        return len(self.coi)

    def __iter__(self):
        perm = self.rs.permutation(len(self.coi))
        for i in range(0, len(perm), self.bs):
            idx = perm[i:i + self.bs]
            cls = self.coi[idx]
            yield [self.images[torch.as_tensor(idx)], self.text[torch.as_tensor(cls)], torch.from_numpy(self.cat[cls])]


def get_supervised_inat_anim(args):
    """(train, val, test, {}) loaders of the CLIP baseline (data.py:54-70): BERT text only, like the reference."""
    if args.text_encoder != "BERT":
        raise NotImplementedError()                                                     # data.py:62-63
    root = args.data_dir + "/iNat-Anim"
    json_path = root + "/inat_anim.json"
    if not os.path.exists(json_path):
        raise FileNotFoundError(f"{json_path} not found: --dataset supervised-inat-anim needs the iNat-Anim files under {root}")
    ann = load_annotations(json_path)
    emb = open_embeddings(root, args.image_embedding_model)
    out = []
    for split in ("train", "val", "test"):
        sp = InatAnimSplit(root, json_path, split, "BERT", args.text_type, args.remove_stop_words, ann, args.device)
        images = torch.from_numpy(read_rows(emb, sp.image_ids))
        out.append(SupervisedSplit(images, sp.class_of_image, sp.class_text.cpu(), np.arange(sp.class_text.shape[0]), args.batch_size,
                                   args.seed + len(split)))
    return out[0], out[1], out[2], {}
