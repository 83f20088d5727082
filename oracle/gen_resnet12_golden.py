"""Generates tests/golden/resnet12_20way.npz: ONE pair of full-size configs[4] episodes (20-way 5-shot, 3 x 84 x 84 images, ResNet-12
channels 64/160/320/640, one inner step, second-order FuMI meta-gradient) through oracle/resnet12_manual.py, in its bf16-rounded form
(what the engine computes) and in plain float32 -- minutes of host time, so the GPU test reads the stored outputs instead of
re-running the sweep.  Inputs are regenerated from the seeds below by the test; only outputs are stored.

    python -m oracle.gen_resnet12_golden

TEST INFRASTRUCTURE ONLY; the oracle is this repository's own restatement ("parity unpinned": the reference has no ResNet-12)."""
import os
import sys
import time

import numpy as np
import torch

from . import conv4_ref as CR
from . import resnet12_manual as M
from . import resnet12_ref as RR

SEED, B, N, K, Q, H, DT, HT, T, ALPHA = 20, 2, 20, 5, 1, 84, 16, 8, 1, 0.01
SAMPLES = 64


def case():
    ep = CR.make_image_episodes(SEED, B, N, K, Q, 3, H, H, DT)
    theta = RR.make_params(SEED, 3, RR.CHANNELS, torch.float32)
    rs = np.random.RandomState(SEED)
    F_ = RR.CHANNELS[-1]
    phi = [torch.from_numpy((rs.standard_normal(s) * sc).astype(np.float32))
           for s, sc in (((HT, DT), 0.3), ((HT,), 0.1), ((F_ + 1, HT), 0.05), ((F_ + 1,), 0.02))]
    return ep, theta, phi


def digest(t):
    """[l2 norm, sum, SAMPLES values at fixed strided positions]"""
    f = t.reshape(-1).double()
    idx = torch.linspace(0, f.numel() - 1, SAMPLES).long()
    return np.concatenate([[float(f.norm()), float(f.sum())], f[idx].numpy()])


def main():
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ep, theta, phi = case()
    out = {}
    for name, rnd in (("bf16", M.bf16_round), ("f32", M._id)):
        t0 = time.time()
        r = M.fumi_meta_step(theta, phi, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, ALPHA, False, rnd=rnd)
        print(f"{name}: {time.time() - t0:.0f} s, loss {float(r['loss']):.5f}", flush=True)
        out[f"{name}.logits"] = r["logits"].numpy()
        out[f"{name}.loss_b"] = r["loss_b"].numpy()
        for i, g in enumerate(r["g_theta"]):
            out[f"{name}.g_theta.{i}"] = digest(g)
        for i, g in enumerate(r["g_phi"]):
            out[f"{name}.g_phi.{i}"] = g.numpy()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resnet12_20way.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
