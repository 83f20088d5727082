"""CPU restatement (oracle) of the reference's episodic hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product (``fumi_amd/``) never does and fails loudly without its HIP library.

What it restates (reference = /root/reference, s-a-malik/fumi):
  * FuMI meta-step          fumi/models/fumi.py:115-196 (+ :198-212 get_hyper_params, :214-218 im_forward)
  * MAML meta-step          fumi/models/maml.py:134-193 (PureImageNetwork :15-33)
  * AM3 step                fumi/models/am3.py:90-126,128-212 + fumi/utils/utils.py:302-402
  * WordEmbedding pooling   fumi/models/common.py:23-41
  * bi-LSTM text encoders   fumi/models/common.py:44-161 (RNN: output states, RnnHid: cell states)
  * CLIP baseline           fumi/models/clip.py:11-41 (forward), :96-108 (symmetric cross-entropy step)
  * torchmeta 1.7.0 functional-linear / gradient_update_parameters contract (requirements.txt:10; source
    absent from /root/reference -> restated from its published behaviour)

It is eager PyTorch on CPU, one Python loop per episode, ``autograd.grad(create_graph=True)`` per
inner step -- op-for-op the structure of the reference, written out-of-place (the reference's
in-place ``hyper_params -=`` at fumi.py:168 raises on torch>=2; SURVEY.md section 8c proves the
out-of-place form equals the reference run under ``allow_mutation_on_saved_tensors``).

PINNING: ``tests/test_oracle_golden.py`` checks every function here against golden vectors that
``oracle/refharness/gen_golden.py`` produced by importing and running the reference's own files in
the build container (fixtures under ``tests/golden/``).  The torchmeta boundary itself is "parity
unpinned" (the reference holds no tests at it).

All functions are dtype-generic (run them in float64 for a high-precision oracle).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def class_text_select(text_enc, targets, n_way):
    """fumi.py:207-210 -- row of the FIRST support sample of each class.  text_enc [S,Dt], targets [S]."""
    rows = []
    for i in range(n_way):
        idx = (targets == i).nonzero(as_tuple=True)[0]
        if idx.numel() == 0:
            raise IndexError(f"class {i} has no support sample (reference raises here too)")
        rows.append(text_enc[idx[0]])
    return torch.stack(rows, 0)


def hyper_net(c, phi, tanh_head):
    """fumi.py:70-86,104-113 -- Linear(Dt,Ht) . ReLU . Linear(Ht,H+1) [. Tanh].  phi = (A0,a0,A1,a1)."""
    A0, a0, A1, a1 = phi
    u = torch.relu(F.linear(c, A0, a0))
    h = F.linear(u, A1, a1)
    return torch.tanh(h) if tanh_head else h


def im_net(x, theta, masks=None):
    """fumi.py:89-100 / torchmeta MetaSequential(MetaLinear, ReLU[, Dropout])* with external params.
    theta = [W0,b0,W1,b1,...]; every layer is followed by ReLU and, in train mode with dropout_rate > 0, by nn.Dropout:
    ``masks[i]`` is that layer's mask already scaled by 1/(1-p) (the reference draws it from torch's RNG; parity tests
    feed the engine's counter-based masks instead)."""
    for i in range(0, len(theta), 2):
        x = torch.relu(F.linear(x, theta[i], theta[i + 1]))
        if masks is not None:
            x = x * masks[i // 2]
    return x


def im_forward(x, theta, h, masks=None):
    """fumi.py:214-218 -- features @ h[:, :-1].T + h[:, -1]  (probe-verified equal to the matmul/squeeze form)."""
    feat = im_net(x, theta, masks)
    return feat @ h[:, :-1].t() + h[:, -1]


def word_embedding_pool(tokens, table, pad_id, mode="mean"):
    """common.py:23-41 -- frozen embedding gather then masked mean (sum / #non-PAD) or max over ALL positions."""
    emb = table[tokens]                              # [..., L, E]
    if mode == "mean":
        lens = (tokens != pad_id).sum(-1, keepdim=True)
        return emb.sum(-2) / lens
    elif mode == "max":
        return emb.max(-2)[0]
    raise NameError(f"{mode} pooling strat not defined")


# ----------------------------------------------------------------------------------------------
# FuMI meta-step
# ----------------------------------------------------------------------------------------------
QUERY_CALL = 1 << 20


def fumi_episode(theta, phi, text_s, x_s, y_s, x_q, n_way, T, alpha, tanh_head, first_order=False, drop=None):
    """One episode up to the query logits (graph kept).  fumi.py:156-178.
    drop(call, layer, rows, width) -> scaled dropout mask of that forward call (call = inner step, QUERY_CALL = query)."""
    c = class_text_select(text_s, y_s, n_way)
    h = hyper_net(c, phi, tanh_head)
    th = list(theta)
    L = len(theta) // 2
    mk = lambda call, x: None if drop is None else [drop(call, i, x.shape[0], theta[2 * i].shape[0]) for i in range(L)]
    for t in range(T):
        logit = im_forward(x_s, th, h, mk(t, x_s))
        inner = F.cross_entropy(logit, y_s)
        grads = torch.autograd.grad(inner, [h] + th, create_graph=not first_order)
        h = h - alpha * grads[0]
        th = [p - alpha * g for p, g in zip(th, grads[1:])]
    return im_forward(x_q, th, h, mk(QUERY_CALL, x_q))


def fumi_meta_step(theta, phi, text_s, x_s, y_s, x_q, y_q, n_way, T, alpha, tanh_head,
                   need_grad=True, first_order=False, dropout=None, extra=None):
    """fumi.py:115-196.  theta/phi: lists of leaf tensors (requires_grad set by the caller when need_grad).
    text_s [B,S,Dt] (already text-encoded), x_s [B,S,D], y_s [B,S], x_q [B,Qn,D], y_q [B,Qn].
    Returns dict(logits [B,Qn,N], preds [B,Qn] int64, loss_b [B], acc_b [B], loss, acc,
                 g_theta, g_phi = gradients of the MEAN loss (what ``outer_loss.backward()`` leaves in .grad)).
    extra: further tensors text_s depends on (a trainable text encoder's weights under --fine_tune, fumi.py:65-67, or text_s
    itself): their gradients of the same loss come back as g_extra."""
    B = x_s.shape[0]
    logits, loss_b = [], []
    for b in range(B):
        drop = None if dropout is None else (lambda call, layer, rows, width, b=b: dropout(b, call, layer, rows, width))
        lq = fumi_episode(theta, phi, text_s[b], x_s[b], y_s[b], x_q[b], n_way, T, alpha, tanh_head,
                          first_order, drop)
        logits.append(lq)
        loss_b.append(F.cross_entropy(lq, y_q[b]))
    loss = torch.stack(loss_b).sum() / B
    out = _pack(logits, loss_b, y_q)
    out["loss"] = loss.detach()
    if need_grad:
        leaves = list(theta) + list(phi) + list(extra or [])
        g = torch.autograd.grad(loss, leaves, allow_unused=True)
        g = [torch.zeros_like(p) if gi is None else gi for gi, p in zip(g, leaves)]
        out["g_theta"], out["g_phi"] = g[:len(theta)], g[len(theta):len(theta) + len(phi)]
        out["g_extra"] = g[len(theta) + len(phi):]
    return out


def _pack(logits, loss_b, y_q):
    logits = torch.stack([l.detach() for l in logits])
    preds = logits.max(dim=-1)[1]                                   # fumi.py:180 / :329-331 (first max)
    acc_b = preds.eq(y_q).float().mean(-1)
    return dict(logits=logits, preds=preds, loss_b=torch.stack([l.detach() for l in loss_b]),
                acc_b=acc_b, acc=acc_b.mean())


# ----------------------------------------------------------------------------------------------
# MAML meta-step (maml.py:134-193): same inner loop, head = learned lin_final inside the parameter list
# ----------------------------------------------------------------------------------------------
def maml_forward(x, params):
    """PureImageNetwork.forward (maml.py:15-33): (Linear, ReLU)* then lin_final (no ReLU)."""
    for i in range(0, len(params) - 2, 2):
        x = torch.relu(F.linear(x, params[i], params[i + 1]))
    return F.linear(x, params[-2], params[-1])


def maml_meta_step(params, x_s, y_s, x_q, y_q, T, alpha, first_order=False, need_grad=True):
    B = x_s.shape[0]
    logits, loss_b = [], []
    for b in range(B):
        p = list(params)
        for _ in range(T):
            inner = F.cross_entropy(maml_forward(x_s[b], p), y_s[b])
            grads = torch.autograd.grad(inner, p, create_graph=not first_order)
            p = [w - alpha * g for w, g in zip(p, grads)]
        lq = maml_forward(x_q[b], p)
        logits.append(lq)
        loss_b.append(F.cross_entropy(lq, y_q[b]))
    loss = torch.stack(loss_b).sum() / B
    out = _pack(logits, loss_b, y_q)
    out["loss"] = loss.detach()
    if need_grad:
        out["g_params"] = list(torch.autograd.grad(loss, list(params)))
    return out


# ----------------------------------------------------------------------------------------------
# AM3 (am3.py:90-212, utils.py:302-402)
# ----------------------------------------------------------------------------------------------
def get_num_samples(targets, num_classes, dtype):
    """utils.py:379-387"""
    ones = torch.ones_like(targets, dtype=dtype)
    out = ones.new_zeros((targets.size(0), num_classes))
    out.scatter_add_(1, targets, ones)
    return out


def get_prototypes(im_emb, text_emb, lamdas, targets, num_classes):
    """utils.py:331-376 -- per-class means (count clamped >= 1) then lamda*im + (1-lamda)*text."""
    B, P = im_emb.size(0), im_emb.size(-1)
    n = get_num_samples(targets, num_classes, im_emb.dtype).unsqueeze(-1)
    n = torch.max(n, torch.ones_like(n))
    idx = targets.unsqueeze(-1).expand_as(im_emb)
    im_p = im_emb.new_zeros((B, num_classes, P)).scatter_add(1, idx, im_emb) / n
    tx_p = text_emb.new_zeros((B, num_classes, P)).scatter_add(1, idx, text_emb) / n
    lam = lamdas.new_zeros((B, num_classes, 1)).scatter_add(1, targets.unsqueeze(-1), lamdas) / n
    return lam * im_p + (1 - lam) * tx_p


def sq_distances(prototypes, emb):
    """[B,N,Qn] squared euclidean distances (utils.py:400-401)."""
    return ((prototypes.unsqueeze(2) - emb.unsqueeze(1)) ** 2).sum(-1)


def prototypical_loss(prototypes, emb, targets):
    """utils.py:390-402 -- CE over the class dim (dim 1) of -dist, mean over B*Qn."""
    return F.cross_entropy(-sq_distances(prototypes, emb), targets)


def am3_step(w, text_s, x_s, y_s, x_q, y_q, n_way, lamda_fixed=None, need_grad=True, masks=None, extra=None):
    """am3.py:128-212 with dropout 0.  w = dict(Wi,bi, G0,g0,G1,g1, H0,h0,H1,h1) (image_encoder, g, h).
    text_s [B,S,Dt] is the per-sample text encoding (identical within a class in the dataset, not required)."""
    im_s = F.linear(x_s, w["Wi"], w["bi"])
    im_q = F.linear(x_q, w["Wi"], w["bi"])
    # masks = (mask_g [B*S,Ht], mask_h [B*S,Ht]) scaled by 1/(1-p): the train-mode nn.Dropout of g and h (am3.py:82,88)
    t1 = torch.relu(F.linear(text_s, w["G0"], w["g0"]))
    if masks is not None:
        t1 = t1 * masks[0].view_as(t1)
    tx = F.linear(t1, w["G1"], w["g1"])
    l1 = torch.relu(F.linear(tx, w["H0"], w["h0"]))
    if masks is not None:
        l1 = l1 * masks[1].view_as(l1)
    lam = torch.sigmoid(F.linear(l1, w["H1"], w["h1"]))
    if lamda_fixed == 0:
        lam = torch.zeros_like(lam)
    elif lamda_fixed == 1:
        lam = torch.ones_like(lam)
    proto = get_prototypes(im_s, tx, lam, y_s, n_way)
    d = sq_distances(proto, im_q)                                   # [B,N,Qn]
    loss = F.cross_entropy(-d, y_q)
    preds = d.detach().transpose(1, 2).min(dim=-1)[1]               # utils.py:315-317 (first min)
    out = dict(loss=loss.detach(), preds=preds, lamda_s=lam.detach().squeeze(-1),
               avg_lamda=lam.detach().mean(), dist=d.detach(),
               acc=preds.eq(y_q).float().mean())
    if need_grad:
        names = ["Wi", "bi", "G0", "g0", "G1", "g1", "H0", "h0", "H1", "h1"]
        leaves = [w[k] for k in names] + list(extra or [])          # extra: tensors text_s depends on (a trainable text encoder)
        g = torch.autograd.grad(loss, leaves, allow_unused=True)
        out["grads"] = OrderedDict((k, torch.zeros_like(w[k]) if gi is None else gi) for k, gi in zip(names, g))
        out["g_extra"] = [torch.zeros_like(p) if gi is None else gi for gi, p in zip(g[len(names):], leaves[len(names):])]
    return out


# ----------------------------------------------------------------------------------------------
# bi-LSTM text encoders (common.py:44-161): frozen unless --fine_tune
# ----------------------------------------------------------------------------------------------
def lstm_encode(tokens, table, lstm_w, pad_id, use_cell):
    """RNN.forward (use_cell=False, common.py:76-107) / RnnHid.forward (use_cell=True, :139-161) on tokens [B, S, L]:
    embedding gather, packed bidirectional single-layer LSTM, then the forward direction's state at the last real token and
    the backward direction's state at token 0 -- i.e. each direction's FINAL hidden state h_n (RNN) or cell state c_n (RnnHid).
    lstm_w = [W_ih, W_hh, b_ih, b_hh, W_ih_reverse, W_hh_reverse, b_ih_reverse, b_hh_reverse] (nn.LSTM's own layout: gate order
    i, f, g, o).  Rows with no real token get zeros (pack_padded_sequence would raise; the loader never produces them)."""
    B, S, L = tokens.shape
    flat = tokens.reshape(-1, L)
    lens = (flat != pad_id).sum(-1)
    x = table[flat]                                                  # [R, L, E]
    H = lstm_w[1].shape[1]
    outs = []
    for d in range(2):
        W_ih, W_hh, b_ih, b_hh = lstm_w[4 * d:4 * d + 4]
        h = x.new_zeros(flat.shape[0], H)
        c = x.new_zeros(flat.shape[0], H)
        steps = range(L) if d == 0 else range(L - 1, -1, -1)
        for t in steps:
            g = F.linear(x[:, t], W_ih, b_ih) + F.linear(h, W_hh, b_hh)
            i, f, gg, o = g.chunk(4, -1)
            c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h2 = torch.sigmoid(o) * torch.tanh(c2)
            live = (t < lens).unsqueeze(-1)
            c, h = torch.where(live, c2, c), torch.where(live, h2, h)
        outs.append(c if use_cell else h)
    return torch.cat(outs, -1).view(B, S, 2 * H)


# ----------------------------------------------------------------------------------------------
# CLIP baseline (clip.py)
# ----------------------------------------------------------------------------------------------
CLIP_KEYS = ["text_fc.weight", "text_fc.bias", "text_fc2.weight", "text_fc2.bias",
             "image_fc.weight", "image_fc.bias", "image_fc2.weight", "image_fc2.bias"]


def clip_forward(w, text, image):
    """clip.py:27-41 -- two Linear.ReLU.Linear towers, cosine similarity of every (text row, image row) pair: [nt, ni]."""
    tl = F.linear(torch.relu(F.linear(text, w[0], w[1])), w[2], w[3])
    il = F.linear(torch.relu(F.linear(image, w[4], w[5])), w[6], w[7])
    return (tl @ il.t()) / tl.norm(dim=1)[:, None] / il.norm(dim=1)[None, :]


def clip_step(w, text, image, need_grad=True):
    """clip.py:96-108 -- symmetric cross-entropy of the similarity matrix against the diagonal (no temperature)."""
    sim = clip_forward(w, text, image)
    labels = torch.arange(sim.shape[0])
    loss = (F.cross_entropy(sim, labels) + F.cross_entropy(sim.t(), labels)) / 2.
    out = dict(sim=sim.detach(), loss=loss.detach())
    if need_grad:
        out["grads"] = list(torch.autograd.grad(loss, list(w)))
    return out


# ----------------------------------------------------------------------------------------------
# hypernet head initialiser live path (hypernet_init.py:137-167 -> :88-117 -> :23-25 -> :12-19)
# ----------------------------------------------------------------------------------------------
def hypernet_bias_init_(weight, bias, gain=2.0 ** 0.5):
    """--hypernet_bias_init: head weight <- 0; head bias <- N(0,1) direction scaled to norm ``gain``
    (normc over the single [1, H+1] row; gain = calculate_gain('relu'))."""
    with torch.no_grad():
        weight.zero_()
        row = bias.view(1, -1)
        row.normal_(0, 1)
        row *= gain / torch.sqrt(row.pow(2).sum(1, keepdim=True))
    return weight, bias
