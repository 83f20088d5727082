"""iNat-Anim on-disk formats: split, image lists, text rows, embedding rows."""
import json
import os

import numpy as np
import pytest
import torch

from fumi_amd.dataset import inat_anim as IA

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORDS = ("the bird has a long Beak and 2 red wings it lives near water of rivers Small mammal with brown fur eats "
         "insects at night large cat spotted coat hunts alone über-fast frog").split()


def make_files(tmp, n_cat=20, per=14, D=16, seed=3):
    rs = np.random.RandomState(seed)
    root = os.path.join(tmp, "iNat-Anim")
    os.makedirs(root, exist_ok=True)
    cats = [dict(name=f"Genus{i} species{i}", common_name=f"common {WORDS[i % len(WORDS)]}",
                 description=" ".join(rs.choice(WORDS, size=rs.randint(3, 12)))) for i in range(n_cat)]
    cat_of = rs.permutation(np.repeat(np.arange(n_cat), per))
    images = [dict(id=i, file_name=f"{i}.jpg") for i in range(len(cat_of))]
    anns = [dict(category_id=int(c)) for c in cat_of]
    with open(os.path.join(root, "inat_anim.json"), "w") as f:
        json.dump(dict(categories=cats, images=images, annotations=anns), f)
    emb = rs.standard_normal((len(cat_of), D)).astype(np.float32)
    np.save(os.path.join(root, "image_embeddings_resnet-34.npy"), emb)
    return root, cats, cat_of, emb


def test_class_split_is_the_seed0_shuffle():
    st = np.random.get_state()
    try:
        np.random.seed(0)                                     # data.py:317 + :378-379
        ref = np.arange(25)
        np.random.shuffle(ref)
    finally:
        np.random.set_state(st)
    assert np.array_equal(IA.class_split(25, "train"), ref[:15])
    assert np.array_equal(IA.class_split(25, "val"), ref[15:20])
    assert np.array_equal(IA.class_split(25, "test"), ref[20:])
    with pytest.raises(ValueError):
        IA.class_split(25, "all")


def test_split_images_and_text(tmp_path):
    root, cats, cat_of, emb = make_files(str(tmp_path))
    ann = IA.load_annotations(os.path.join(root, "inat_anim.json"))
    seen = []
    for split in ("train", "val", "test"):
        sp = IA.InatAnimSplit(root, "inat_anim.json", split, "glove", ["description"], annotations=ann)
        seen += list(sp.categories)
        # images in file order, class = position of the category in the split's (shuffled) class list
        want = [i for i in range(len(cat_of)) if cat_of[i] in set(sp.categories)]
        assert list(sp.image_ids) == want
        assert all(sp.categories[c] == cat_of[i] for i, c in zip(sp.image_ids, sp.class_of_image))
        # token rows: one per class, equal length, PAD-filled, and they decode back to the description's tokens
        inv = {v: k for k, v in sp.dictionary.items()}
        assert sp.class_text.dtype == torch.int64 and sp.class_text.shape[0] == len(sp.categories)
        for row, c in zip(sp.class_text.tolist(), sp.categories):
            words = [inv[t] for t in row]
            toks = IA.tokenize(cats[c]["description"].lower())
            assert words[:len(toks)] == toks and all(w == "PAD" for w in words[len(toks):])
        assert max(len(IA.tokenize(cats[c]["description"])) for c in sp.categories) == sp.class_text.shape[1]
    assert sorted(seen) == list(range(20))
    # the dictionary spans the descriptions of ALL categories and is the same for every split
    d_train = IA.InatAnimSplit(root, "inat_anim.json", "train", "glove", ["description"], annotations=ann).dictionary
    d_test = IA.InatAnimSplit(root, "inat_anim.json", "test", "glove", ["description"], annotations=ann).dictionary
    assert d_train == d_test and "PAD" in d_train
    assert sorted(d_train.values()) == list(range(len(d_train)))


def test_dictionary_ids_follow_document_order_then_sorted_tokens():
    t2i = IA.build_dictionary([["pear", "apple", "pear"], ["zebra", "apple", "bee"], ["PAD"]])
    assert t2i == {"apple": 0, "pear": 1, "bee": 2, "zebra": 3, "PAD": 4}
    assert IA.tokenize("A 12 red-winged bird2, über <PAD>") == ["A", "red", "winged", "bird", "über", "PAD"]


def test_text_fields_and_stop_words(tmp_path):
    root, cats, _, _ = make_files(str(tmp_path))
    ann = IA.load_annotations(os.path.join(root, "inat_anim.json"))
    both = IA.descriptions(ann, [3, 1], ["common_name", "description"])
    assert both == [cats[3]["description"] + " " + cats[3]["common_name"], cats[1]["description"] + " " + cats[1]["common_name"]]
    assert IA.descriptions(ann, [2], ["label"]) == [cats[2]["name"]]
    with pytest.raises(NameError):
        IA.descriptions(ann, [0], ["caption"])
    assert IA.strip_stop_words(["the bird has The beak"], ["the", "has"]) == ["bird The beak"]      # case-sensitive
    with open(os.path.join(root, "stopwords_english.txt"), "w") as f:
        f.write("the\nhas\na\n")
    try:
        import nltk  # noqa: F401
    except ImportError:
        assert IA.english_stop_words(root) == ["the", "has", "a"]
        sp = IA.InatAnimSplit(root, "inat_anim.json", "val", "glove", ["description"], remove_stop_words=True, annotations=ann)
        assert not any(w in ("the", "has", "a") for t in sp.texts for w in t.split())


def test_embedding_rows(tmp_path):
    root, _, _, emb = make_files(str(tmp_path))
    e = IA.open_embeddings(root, "resnet-34")
    ids = np.array([7, 3, 3, 250, 0])
    assert np.array_equal(IA.read_rows(e, ids), emb[ids])
    with pytest.raises(FileNotFoundError):
        IA.open_embeddings(root, "resnet-152")


def test_bert_rows_are_the_unmasked_mean(tmp_path):
    """data.py:489-492: mean over every position of the padded batch, padding included (tiny random BERT, no download)."""
    transformers = pytest.importorskip("transformers")
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted({w.lower() for w in WORDS})
    path = str(tmp_path / "bert")
    os.makedirs(path)
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        f.write("\n".join(vocab) + "\n")
    cfg = transformers.BertConfig(vocab_size=len(vocab), hidden_size=32, num_hidden_layers=1, num_attention_heads=2,
                                  intermediate_size=64, max_position_embeddings=64)
    torch.manual_seed(0)
    model = transformers.BertModel(cfg)
    model.save_pretrained(path)
    transformers.BertTokenizer(os.path.join(path, "vocab.txt")).save_pretrained(path)
    texts = ["the bird has a long beak", "small mammal", "large cat spotted coat hunts alone at night near water"]
    out = IA.bert_embeddings(texts, None, path, batch_size=2)
    assert out.shape == (3, 32)
    tok = transformers.BertTokenizer.from_pretrained(path)
    enc = tok(texts, return_token_type_ids=False, return_tensors="pt", padding=True, truncation=True)
    model.eval()
    with torch.no_grad():
        h = model(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"]).last_hidden_state
    assert torch.allclose(out, h.mean(1), atol=1e-5)
    L = int(enc["attention_mask"][1].sum())
    assert not torch.allclose(out[1], h[1, :L].mean(0), atol=1e-4)        # a masked mean would differ: padding is counted


@pytest.mark.gpu
def test_cli_on_inat_anim_files(tmp_path, monkeypatch):
    """python -m fumi_amd.main --dataset inat-anim over files in the reference's formats: trains and tests through the
    GPU sampler; batches carry the dataset's image ids."""
    from types import SimpleNamespace
    root, cats, cat_of, emb = make_files(str(tmp_path), n_cat=30, per=40, D=512)
    args = SimpleNamespace(data_dir=str(tmp_path), image_embedding_model="resnet-34", num_ways=5, num_shots=2, num_shots_test=3,
                           text_encoder="glove", text_type=["description"], remove_stop_words=False, batch_size=4, seed=1,
                           device=torch.device("cuda", 0))
    train, val, test, dictionary = IA.get_inat_anim(args)
    b = train.batch(0)
    (ids, text, x), y = b["train"]
    assert ids.shape == (4, 10) and x.shape == (4, 10, 512) and text.dtype == torch.int64
    assert torch.equal(x.cpu(), torch.from_numpy(emb[ids.cpu().numpy()]))             # the rows of the reported image ids
    split_cats = IA.class_split(30, "train")
    assert all(int(cat_of[i]) in set(split_cats) for i in ids.cpu().numpy().ravel())
    for bi in range(4):                                                             # one category per class slot
        for n in range(5):
            assert len({int(cat_of[i]) for i in ids[bi, 2 * n:2 * n + 2].cpu().numpy()}) == 1
    assert val.Q == 20 and test.Q == 20 and "PAD" in dictionary
    from fumi_amd import main as cli
    from fumi_amd.models import common
    monkeypatch.chdir(tmp_path)
    common.register_word_vectors("glove", common.ArrayKeyedVectors(
        sorted(dictionary), np.random.RandomState(1).standard_normal((len(dictionary), 300)).astype(np.float32)))
    argv = ["--model", "fumi", "--dataset", "inat-anim", "--data_dir", str(tmp_path), "--image_embedding_model", "resnet-34",
            "--im_emb_dim", "512", "--im_hid_dim", "64", "32", "--text_encoder", "glove", "--text_emb_dim", "300",
            "--num_shots", "2", "--num_shots_test", "3", "--batch_size", "4", "--epochs", "6", "--eval_freq", "3",
            "--num_ep_test", "8", "--num_train_adapt_steps", "1", "--num_test_adapt_steps", "2", "--dropout", "0",
            "--log_dir", str(tmp_path / "logs"), "--wandb_offline"]
    res = cli.main(cli.parse_args(argv))
    assert np.isfinite(res["test_loss"]) and 0.0 <= res["test_acc"] <= 1.0
    with pytest.raises(FileNotFoundError):
        cli.main(cli.parse_args(argv[:5] + [str(tmp_path / "nowhere")] + argv[6:]))
