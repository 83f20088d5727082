"""CPU check of the TEST INFRASTRUCTURE tests/rn12_stages.py (no GPU, no HIP library): a mock of ``fumi_hip_rn12_probe`` serves the
stored maps of oracle/resnet12_manual.py's own bf16-rounded sweep in the engine's layouts (bf16 padded channels-last maps, the
parameter slab of fumi_amd/csrc/rn12.hip::net_init, coefficient tables).  The stage checker must accept that tape completely --
which pins its wiring of every stage to the sweep that tests/test_resnet12_manual.py proves equal to autograd -- and must reject
a tape with one corrupted map, one wrong-signed gradient or one swapped buffer."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conv4_ref as CR
from oracle import resnet12_manual as M
from oracle import resnet12_ref as RR

import rn12_stages as ST


def _cl(t):
    """[B, M, C, H, W] float64 -> flat bf16 padded channels-last"""
    return F.pad(t, (1, 1, 1, 1)).permute(0, 1, 3, 4, 2).contiguous().reshape(-1).to(torch.bfloat16)


class MockHip:
    """hip.rn12_probe / resnet12_set_option backed by the oracle's trace of every episode."""

    class FumiHipError(RuntimeError):
        pass

    def __init__(self, ep, theta, head0, channels, T, alpha, corrupt=None):
        self.ep, self.theta, self.head0, self.channels, self.T, self.alpha = ep, [t.double() for t in theta], head0, channels, T, alpha
        self.stop, self.corrupt = 0, corrupt
        self.B = ep["x_s"].shape[0]
        self.lay = ST.Layout(channels, ep["x_s"].shape[2], ep["x_s"].shape[3])
        self.tr = None

    def resnet12_set_option(self, key, value):
        if key == 1:
            self.stop = value

    def run(self):
        self.tr = []
        zs = []
        for b in range(self.B):
            tr = dict(hvp_stop=self.stop)
            zq, _, bth, bh = M.episode_grads(self.theta, self.head0[b], self.ep["x_s"][b].double(), self.ep["y_s"][b],
                                             self.ep["x_q"][b].double(), self.ep["y_q"][b], self.T, self.alpha, rnd=M.bf16_round, trace=tr)
            tr["bar"] = (bth, bh)
            self.tr.append(tr)
            zs.append(zq)
        return dict(logits=torch.stack(zs).float())

    def _vec(self, tensors):
        v = torch.zeros(self.lay.PSZ, dtype=torch.float64)
        flat = torch.cat([t.reshape(-1) for t in tensors])
        v[:flat.numel()] = flat
        return v

    def _coef(self, tp_bn, u):
        c = torch.zeros(ST.RCF_N, u.shape[1], dtype=torch.float64)
        r, g = tp_bn["r"].reshape(-1), tp_bn["g"].reshape(-1)
        c[ST.RCF["MU"]], c[ST.RCF["R"]], c[ST.RCF["A"]] = u.mean((0, 2, 3)), r, g * r
        return c

    def rn12_probe(self, ws, dev, pas, kind, l=0, idx=0):
        T, B = self.T, self.B
        per = lambda fn: [fn(self.tr[b]) for b in range(B)]
        if pas == -1:
            if kind == 0:
                return torch.stack(per(lambda tr: self._vec(tr["steps"][idx][0] if idx < T else tr["theta_T"]))).float().reshape(-1)
            if kind == 1:
                return torch.stack(per(lambda tr: tr["steps"][idx][1] if idx < T else tr["head_T"])).float().reshape(-1)
            if kind == 2:
                g = torch.stack(per(lambda tr: self._vec(tr["steps"][idx][2])))
                if self.corrupt == "sign" and idx == 0:
                    o = self.lay.off[(0, 1)][0]
                    g[:, o:o + 8] *= -1.0
                return g.float().reshape(-1)
            if kind == 3:
                return torch.stack(per(lambda tr: tr["steps"][idx][3])).float().reshape(-1)
            if kind == 4:
                return torch.stack(per(lambda tr: self._vec(tr["bar"][0]))).float().reshape(-1)
            if kind == 5:
                return torch.stack(per(lambda tr: tr["bar"][1])).float().reshape(-1)
            if kind == 6:
                return torch.stack(per(lambda tr: self._vec(tr["hv"][-1][0]))).float().reshape(-1)
            if kind == 7:
                return torch.stack(per(lambda tr: tr["hv"][-1][1])).float().reshape(-1)
            if kind == 8:
                return torch.stack(per(lambda tr: self._vec(tr["V"][0]))).float().reshape(-1)
            if kind == 9:
                return torch.stack(per(lambda tr: tr["V"][1])).float().reshape(-1)
            x = self.ep["x_q" if kind == 11 else "x_s"].double()
            x16 = torch.zeros(x.shape[0], x.shape[1], 16, x.shape[3], x.shape[4], dtype=torch.float64)
            x16[:, :, :x.shape[2]] = M.bf16_round(x)
            return _cl(x16)
        tangent = pas == T + 1
        tape_of = lambda tr: tr["tapes"][self.stop] if tangent else (tr["query"] if pas == T else tr["tapes"][pas])
        blk = lambda tr: tape_of(tr)["blocks"][l]
        if kind <= 5:
            names = (("ud", "ad", "outd", "dud", "dad", "dod") if tangent else ("u", "a", "out", "du", "da", "do"))[kind]
            get = (lambda tr: blk(tr)[names]) if kind in (2, 5) else (lambda tr: blk(tr)[names][idx])
            t = torch.stack(per(get))
            if self.corrupt == "map" and not tangent and pas == 0 and kind == 1 and l == 1 and idx == 0:
                t = t.clone(); t[0, 0, 3, 1, 1] += 0.25 * t.abs().max()
            if self.corrupt == "swap" and tangent and kind == 3 and l == 0 and idx == 0:
                t = torch.stack(per(lambda tr: blk(tr)["dud"][3]))
            return _cl(t)
        if kind == 6:
            return torch.stack(per(lambda tr: self._coef(blk(tr)["bn"][idx], blk(tr)["u"][idx]))).float().reshape(-1)
        key = {7: "fd" if tangent else "f", 8: "dfd", 10: "p", 11: "dzd" if tangent else "dz"}.get(kind)
        if kind == 8 and not tangent:
            return torch.stack(per(lambda tr: tape_of(tr)["dz"] @ tape_of(tr)["h"][:, :-1])).float().reshape(-1)
        if kind == 9:
            return torch.stack(per(lambda tr: tape_of(tr)["f"] @ tape_of(tr)["h"][:, :-1].t() + tape_of(tr)["h"][:, -1])).float().reshape(-1)
        return torch.stack(per(lambda tr: tape_of(tr)[key])).float().reshape(-1)


class _Ws:
    def read_status(self):
        return 0


def _setup(corrupt=None):
    channels, H, B, N, T, alpha = (32, 64), 8, 2, 3, 2, 0.05
    ep = CR.make_image_episodes(41, B, N, 2, 2, 3, H, H, 6)
    theta = RR.make_params(41, 3, channels, torch.float32)
    rs = np.random.RandomState(41)
    head0 = torch.from_numpy(rs.standard_normal((N, channels[-1] + 1)) * 0.1)[None].expand(B, -1, -1)
    mock = MockHip(ep, theta, head0, channels, T, alpha, corrupt)
    chk, lay, final = ST.check_step(mock, _Ws(), None, mock.run, ep, theta, head0, channels, T, alpha, hvp_steps=[1, 0])
    return chk


def test_stage_checker_accepts_the_sweeps_own_tape():
    chk = _setup()
    assert len(chk.rows) > 600
    assert not chk.bad, "\n".join(chk.bad[:20])
    # float64 on both sides: the wiring is exact, up to single bf16 ulps where the mock's fp32 parameter vectors cross a rounding boundary
    assert max(r[1] for r in chk.rows) <= 2.0 ** -8 and sum(r[1] > 1e-6 for r in chk.rows) <= 8


@pytest.mark.parametrize("corrupt", ["map", "sign", "swap"])
def test_stage_checker_rejects_a_corrupted_tape(corrupt):
    chk = _setup(corrupt)
    assert chk.bad, corrupt
