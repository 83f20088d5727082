"""CLIP baseline -- host-side mirror of fumi/models/clip.py (supervised, not episodic: SURVEY.md section 8, row f4).

Same class name, constructor keywords, ``state_dict`` keys (``text_fc``, ``text_fc2``, ``image_fc``, ``image_fc2``) and the
free functions ``evaluate`` / ``training_run`` with the reference's signatures and control flow (clip.py:44-141).  The
arithmetic -- both towers, the cosine-similarity matrix, the symmetric cross-entropy and its backward -- is one call of the
engine (csrc/textenc.hip: fumi_hip_clip_step)."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import engine as _engine
from ..utils import utils as utils
from ..utils.wandb_compat import wandb


class CLIP(nn.Module):
    def __init__(self, text_input_dim, image_input_dim, latent_dim):
        super().__init__()
        self.text_input_dim = text_input_dim
        self.image_input_dim = image_input_dim
        self.latent_dim = latent_dim
        self.text_fc = nn.Linear(text_input_dim, latent_dim)
        self.text_af = nn.ReLU()
        self.text_fc2 = nn.Linear(latent_dim, latent_dim)
        self.image_fc = nn.Linear(image_input_dim, latent_dim)
        self.image_af = nn.ReLU()
        self.image_fc2 = nn.Linear(latent_dim, latent_dim)

    def _w(self):
        return [self.text_fc.weight, self.text_fc.bias, self.text_fc2.weight, self.text_fc2.bias,
                self.image_fc.weight, self.image_fc.bias, self.image_fc2.weight, self.image_fc2.bias]

    def forward(self, text, image):
        """[len(text), len(image)] cosine similarities (clip.py:27-41)."""
        out = _engine.get_engine().clip_step(text.contiguous().float(), image.contiguous().float(), [p.detach() for p in self._w()],
                                             need_loss=False, need_grad=False)
        return out["sim"]

    def loss_and_grads(self, text, image):
        """Symmetric cross-entropy of the similarity matrix against the diagonal (clip.py:101-105); leaves its gradient in
        every parameter's .grad (what loss.backward() does in the reference).  Returns the loss as a 0-d device tensor."""
        w = self._w()
        out = _engine.get_engine().clip_step(text.contiguous().float(), image.contiguous().float(), [p.detach() for p in w],
                                             need_loss=True, need_grad=True)
        for p, g in zip(w, out["grads"]):
            p.grad = g
        return out["loss"].reshape(())


def evaluate(args, model, data):
    """Zero-shot accuracy (clip.py:44-77): every n_ways-th text row against the n_ways images that start at it; a hit when
    its own image (position 0) has the largest similarity."""
    device = args.device
    correct, total, n_ways = 0, 0, args.num_ways
    model.eval()
    for i, batch in enumerate(data):
        batch_text, batch_image = batch[1].to(device), batch[0].to(device)
        batch_size = batch_text.shape[0]
        shot_i, rows = 0, []
        while shot_i + n_ways < batch_size:
            rows.append(shot_i)
            shot_i += n_ways
        if not rows:
            continue
        # the reference calls the model once per group; the similarities of a text row do not depend on the other rows, so
        # one call on the whole batch gives the same numbers -- read the block of each group from it
        sim = model(batch_text, batch_image)
        for r in rows:
            correct += int(int(sim[r, r:r + n_ways].argmax()) == 0)
            total += 1
    return correct / total


def training_run(args, model, optimizer, train_loader, val_loader, n_epochs):
    """clip.py:80-141: epochs over the supervised loader, repeated classes of a batch discarded, symmetric CE, validation accuracy
    per epoch, checkpoint + early stopping, best checkpoint reloaded at the end."""
    device = args.device
    best_acc = evaluate(args, model, val_loader)
    best_epoch = 0
    print('init val_acc', best_acc)
    for epoch in range(n_epochs):
        model.train()
        model.zero_grad()
        for bid, batch in enumerate(train_loader):
            batch_text, batch_image, batch_ids = batch[1].to(device), batch[0].to(device), batch[2]
            _, unique_idxs = np.unique(np.asarray(batch_ids), return_index=True)       # discard repeated classes (clip.py:92-96)
            unique_idxs = torch.as_tensor(unique_idxs, device=device)
            batch_text, batch_image = batch_text[unique_idxs], batch_image[unique_idxs]
            optimizer.zero_grad()
            model.loss_and_grads(batch_text, batch_image)
            optimizer.step()
        val_acc = evaluate(args, model, val_loader)
        print('epoch', epoch, 'val_acc', val_acc)
        wandb.log({'val/acc': val_acc}, step=epoch)
        is_best = val_acc > best_acc
        if is_best:
            best_acc, best_epoch = val_acc, epoch
        utils.save_checkpoint({"batch_idx": epoch, "state_dict": model.state_dict(), "best_loss": best_acc,
                               "optimizer": optimizer.state_dict(), "args": vars(args)}, is_best)
        if args.patience > 0 and epoch - best_epoch > args.patience:
            break
    best_file = os.path.join(wandb.run.dir, "best.pth.tar")
    if os.path.exists(best_file):            # (the reference fails here when no epoch improved on the initial accuracy)
        model, _ = utils.load_checkpoint(model, optimizer, args.device, best_file)
    return model
