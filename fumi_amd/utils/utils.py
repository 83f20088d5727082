"""Flag system, model / optimizer factories and checkpoint IO -- the host-side mirror of fumi/utils/utils.py.

Same flag names, defaults and types as the reference parser (fumi/utils/utils.py:19-229); same factory
behaviour (:232-299) and checkpoint dictionary (:406-441).  Additive flags of this engine are listed last.
The AM3 head math of the reference file (get_prototypes / prototypical_loss / get_preds, :302-402) runs inside the
fused HIP AM3 step; the functions of the same names below are thin device wrappers around it.
"""
import argparse
import os
import shutil

import numpy as np
import torch

from .wandb_compat import wandb

# (flag, kwargs) in the reference's order; defaults/types/help are the contract
_FLAGS = [
    ("--wandb_entity", dict(type=str, default="multimodal-image-cls", help="W&B entity")),
    ("--wandb_project", dict(type=str, default="fumi", help="W&B project")),
    ("--dataset", dict(type=str, default="inat-anim", help="Dataset to use (inat-anim, supervised-inat-anim, synthetic, synthetic-resident)")),
    ("--data_dir", dict(type=str, default="./data", help="Directory to use for data")),
    ("--checkpoint", dict(type=str, default=None, help="Path to pretrained model (a best.pth.tar file, or a W&B run id when wandb is installed)")),
    ("--log_dir", dict(type=str, default="./results", help="Directory to use for results")),
    ("--remove_stop_words", dict(action="store_true", help="Whether to remove stop words")),
    ("--colab", dict(action="store_true", help="Whether the script is running on Google Colab")),
    # optimizer
    ("--epochs", dict(type=int, default=50000, help="Number of meta-learning batches to train for")),
    ("--optim", dict(type=str, default="adam", help="Optimiser")),
    ("--lr", dict(type=float, default=3e-5, help="Learning rate")),
    ("--momentum", dict(type=float, default=0.9, help="Momentum for SGD")),
    ("--batch_size", dict(type=int, default=4, help="Number of tasks in mini-batch")),
    ("--weight_decay", dict(type=float, default=5e-4, help="L2 regulariser")),
    ("--num_warmup_steps", dict(type=float, default=10, help="Warm up lr scheduler")),
    # dataloader
    ("--num_shots", dict(type=int, default=5, help="Number of examples per class (k-shot)")),
    ("--num_ways", dict(type=int, default=5, help="Number of classes per task (N-way)")),
    ("--num_shots_test", dict(type=int, default=32, help="Number of examples per class in query set")),
    ("--augment", dict(action="store_true", help="Augment data with image transformations")),
    ("--num_workers", dict(type=int, default=0, help="Number of workers for dataloader")),
    ("--image_embedding_model", dict(type=str, default="resnet-152", help="resnet-152 embedding (2048 dimensions) or resnet-34 (512 dimensions)")),
    # model
    ("--model", dict(type=str, default="fumi", help="Model to be trained")),
    ("--prototype_dim", dict(type=int, default=64, help="Dimension of latent space")),
    ("--im_encoder", dict(type=str, default="precomputed", help="Type of vision feature extractor (resnet, precomputed; conv4 / resnet12 = Conv4 / bf16 ResNet-12 on raw images, this engine's extensions at the im_net seam)")),
    ("--im_emb_dim", dict(type=int, default=2048, help="Dimension of image embedding (if precomputed)")),
    ("--im_hid_dim", dict(type=int, nargs="+", default=[256, 64], help="Hidden dimension of image model")),
    ("--text_encoder", dict(type=str, choices=["glove", "w2v", "RNN", "RNNhid", "BERT", "rand"], default="BERT",
                            help="Type of text embedding (glove, w2v, RNN, RNNhid, BERT, rand)")),
    ("--pooling_strat", dict(type=str, default="mean", help="Pooling strategy if using word embeddings (mean, max)")),
    ("--fine_tune", dict(action="store_true", help="Whether to fine tune text encoder")),
    ("--text_type", dict(type=str, nargs="+", default=["description"], help="What to use for text embedding (label, description or common_name)")),
    ("--text_emb_dim", dict(type=int, default=768, help="Dimension of text embedding (if precomputed)")),
    ("--text_hid_dim", dict(type=int, default=256, help="Hidden dimension for NN mapping to prototypes and lamda")),
    ("--dropout", dict(type=float, default=0.25, help="Dropout rate")),
    ("--step_size", dict(type=float, default=0.01, help="MAML step size")),
    ("--first_order", dict(action="store_true", help="Whether to use first-order MAML")),
    ("--num_train_adapt_steps", dict(type=int, default=5, help="Number of MAML inner train loop adaptation steps")),
    ("--num_test_adapt_steps", dict(type=int, default=100, help="Number of MAML inner test loop adaptation steps")),
    ("--init_all_layers", dict(action="store_true", help="Whether to initialise all (vs. last) layer weights in FUMI")),
    ("--norm_hypernet", dict(action="store_true", help="Whether to normalize output of the FUMI hypernetwork (tanh)")),
    ("--hypernet_bias_init", dict(action="store_true", help="Whether to initialise hypernet bias for policy")),
    ("--lamda_fixed", dict(default=None, type=int, help="Lambda fixed for am3. Lambda = 0 is text only, Lambda = 1 is image only")),
    ("--clip_latent_dim", dict(type=int, default=512, help="Dimension of CLIP latent space")),
    # run
    ("--seed", dict(type=int, default=123, help="patience for early stopping")),
    ("--patience", dict(type=int, default=10000, help="Early stopping patience")),
    ("--eval_freq", dict(type=int, default=2500, help="Number of batches between validation/checkpointing")),
    ("--wandb_experiment", dict(type=str, default="debug", help="Name for experiment (for wandb group)")),
    ("--evaluate", dict(action="store_true", help="skip training")),
    ("--num_ep_test", dict(type=int, default=1000, help="Number of few-shot episodes to compute test accuracy")),
    ("--disable_cuda", dict(action="store_true", help="don't use GPU")),
    ("--wandb_offline", dict(action="store_true", help="don't save to wandb")),
]
_ENGINE_FLAGS = [     # additive, not in the reference
    ("--synthetic_classes", dict(type=int, default=64, help="[synthetic dataset] number of classes per split")),
    ("--synthetic_vocab", dict(type=int, default=2000, help="[synthetic dataset] vocabulary size for token text")),
    ("--synthetic_seq_len", dict(type=int, default=32, help="[synthetic dataset] token sequence length")),
    ("--image_size", dict(type=int, default=84, help="[--im_encoder conv4] height = width of the input images")),
    ("--image_channels", dict(type=int, default=3, help="[--im_encoder conv4] input channels (1-3)")),
]


def parser():
    p = argparse.ArgumentParser(description="Multimodal image classification")
    for flag, kw in _FLAGS + _ENGINE_FLAGS:
        p.add_argument(flag, **kw)
    return p


def init_model(args, dictionary, watch=True):
    """Model factory (utils.py:232-274).  Unknown names fall through to AM3 exactly like the reference."""
    from ..models import am3, clip, fumi, maml
    if args.model == "maml":
        conv = (dict(im_encoder=args.im_encoder, image_size=args.image_size, image_channels=args.image_channels)
                if args.im_encoder in ("conv4", "resnet12") else {})
        model = maml.PureImageNetwork(im_embed_dim=args.im_emb_dim, n_way=args.num_ways, hidden_dims=args.im_hid_dim, **conv)
    elif args.model == "fumi":
        model = fumi.FUMI(n_way=args.num_ways, im_emb_dim=args.im_emb_dim, im_hid_dim=args.im_hid_dim,
                          text_encoder=args.text_encoder, text_emb_dim=args.text_emb_dim,
                          text_hid_dim=args.text_hid_dim, dropout_rate=args.dropout, dictionary=dictionary,
                          pooling_strat=args.pooling_strat, init_all_layers=args.init_all_layers,
                          norm_hypernet=args.norm_hypernet, fine_tune=args.fine_tune, init_bias=args.hypernet_bias_init,
                          **(dict(im_encoder=args.im_encoder, image_size=args.image_size, image_channels=args.image_channels)
                             if args.im_encoder in ("conv4", "resnet12") else {}))
    elif args.model == "clip":
        model = clip.CLIP(text_input_dim=args.text_emb_dim, image_input_dim=args.im_emb_dim, latent_dim=args.clip_latent_dim)
    else:
        model = am3.AM3(im_encoder=args.im_encoder, im_emb_dim=args.im_emb_dim, text_encoder=args.text_encoder,
                        text_emb_dim=args.text_emb_dim, text_hid_dim=args.text_hid_dim,
                        prototype_dim=args.prototype_dim, dropout=args.dropout, fine_tune=args.fine_tune,
                        dictionary=dictionary, pooling_strat=args.pooling_strat, lamda_fixed=args.lamda_fixed,
                        **(dict(image_size=args.image_size, image_channels=args.image_channels) if args.im_encoder == "conv4" else {}))
    if watch:
        wandb.watch(model, log="all")
    model.to(args.device)
    return model


def _linear_warmup_schedule(opt, num_warmup_steps, num_training_steps):
    """transformers.get_linear_schedule_with_warmup (utils.py:11,293-294): linear ramp then linear decay to 0."""
    def lr_lambda(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)


def init_optim(args, model):
    """Optimizer factory (utils.py:277-299); may return an (optimizer, scheduler) tuple."""
    if args.optim == "adam":
        from ..optim import Adam            # torch.optim.Adam whose step() is one fused HIP launch on the GPU
        return Adam(params=model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    if args.optim == "SGD":
        return torch.optim.SGD(params=model.parameters(), lr=args.lr, weight_decay=args.weight_decay,
                               momentum=args.momentum)
    if args.optim == "adamw":
        # transformers.AdamW(lr) of the pinned 4.5.1: decoupled weight decay, default weight_decay 0.0
        return torch.optim.AdamW(params=model.parameters(), lr=args.lr, weight_decay=0.0)
    if args.optim == "adamw_lin_schedule":
        opt = torch.optim.AdamW(params=model.parameters(), lr=args.lr, weight_decay=0.0)
        return opt, _linear_warmup_schedule(opt, args.num_warmup_steps, args.epochs)
    raise NotImplementedError()


# ---- checkpoints (utils.py:406-441): same dictionary keys, same ckpt/best file names ---------------------------------
def save_checkpoint(checkpoint_dict, is_best):
    """fumi/utils/utils.py:406-419.  Episode-sharded runs keep replicated parameters, so rank 0 alone writes; the others wait
    (the best checkpoint is reloaded by every rank at the end of training)."""
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not sharded or dist.get_rank() == 0:
        ckpt = os.path.join(wandb.run.dir, "ckpt.pth.tar")
        best = os.path.join(wandb.run.dir, "best.pth.tar")
        torch.save(checkpoint_dict, ckpt)
        wandb.save(ckpt)
        if is_best:
            shutil.copyfile(ckpt, best)
            wandb.save(best)
    if sharded:
        dist.barrier()


def load_checkpoint(model, optimizer, device, checkpoint_file):
    checkpoint = torch.load(checkpoint_file, map_location=device, weights_only=False)
    model.load_state_dict(checkpoint["state_dict"])
    optimizer.load_state_dict(checkpoint["optimizer"])
    print(f"Loaded {checkpoint_file}, trained to epoch {checkpoint['batch_idx']} "
          f"with best loss (acc for CLIP) {checkpoint['best_loss']}")
    return model, optimizer


# ---- classification metrics of AM3 (utils.py:319-326) on host integers ------------------------------------------------
def macro_metrics(flat_targets, flat_preds):
    """accuracy + macro precision/recall/F1 (what sklearn's accuracy_score / precision_recall_fscore_support
    (average='macro', zero_division -> 0) return); labels = union of targets and predictions."""
    t = np.asarray(flat_targets).reshape(-1)
    p = np.asarray(flat_preds).reshape(-1)
    acc = float((t == p).mean()) if t.size else 0.0
    labels = np.union1d(t, p)
    prec, rec, f1 = [], [], []
    for c in labels:
        tp = float(((p == c) & (t == c)).sum())
        pp, tt = float((p == c).sum()), float((t == c).sum())
        pr = tp / pp if pp > 0 else 0.0
        rc = tp / tt if tt > 0 else 0.0
        prec.append(pr)
        rec.append(rc)
        f1.append(2 * pr * rc / (pr + rc) if (pr + rc) > 0 else 0.0)
    return acc, float(np.mean(f1)), float(np.mean(prec)), float(np.mean(rec))
