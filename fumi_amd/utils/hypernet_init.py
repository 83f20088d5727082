"""Initialiser of the hypernetwork head used by ``--hypernet_bias_init``.

Only the live path of the reference file is provided (fumi/utils/hypernet_init.py:137-167 -> :88-117 -> :23-25
-> :12-19, called from fumi/models/fumi.py:81-84 with ('relu','normc', adjust_weights=False, adjust_bias=True)):
the head WEIGHT is zeroed and the head BIAS becomes a random direction of norm gain('relu') = sqrt(2), so that at
initialisation every class starts from the same generated classifier."""
import torch
import torch.nn as nn


def normc_(t, gain=1.0):
    """Row-normalised gaussian: each row ~ N(0,1) rescaled to norm ``gain``."""
    with torch.no_grad():
        t.normal_(0, 1)
        t.mul_(gain / t.pow(2).sum(1, keepdim=True).sqrt())
    return t


def hyper_weight_layer_init(activation_function, policy_initialisation_str, hyper_layer_dim, input_dim, output_dim,
                            fix_init_b_gain, override_gain=None, adjust_weights=True, adjust_bias=False, use_film=False):
    if use_film or adjust_weights or not adjust_bias or policy_initialisation_str != 'normc':
        raise NotImplementedError("only the configuration FUMI uses (normc, bias-only) is provided")
    gain = nn.init.calculate_gain(activation_function) if override_gain is None else override_gain

    def apply(module):
        with torch.no_grad():
            module.weight.zero_()
            assert module.bias.numel() == input_dim * output_dim
            normc_(module.bias.view(output_dim, input_dim), gain)
        return module
    return apply
