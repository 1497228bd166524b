"""Weight manifest + deterministic NumPy weight recipe.

The manifest is the reference's serialisation surface (names/shapes
as produced by model/STTODE.py:349-366 and the modules it builds).  ``make_weights``
regenerates a full ``state_dict`` anywhere from a seed (no checkpoint ships with the
reference), with per-tensor scales chosen so that the geodesic attention, the GRU
and the MLPs are all non-degenerate (non-zero biases, LayerNorm gains != 1).
The same dict loads into the reference, the CPU oracle and ``sttode_amd.STTODENet``.
"""
from collections import OrderedDict

import numpy as np

_ATT = 'ODE_Encoder.odeblock.odefunc.layers.0.'


def _trunk(prefix, length, D):
    a = prefix + _ATT
    return [
        (prefix + 'input_fc.weight', (D, 4)), (prefix + 'input_fc.bias', (D,)),
        (prefix + 'input_fc2.weight', (D, D * length)), (prefix + 'input_fc2.bias', (D,)),
        (prefix + 'input_fc3.weight', (D, D + 3)), (prefix + 'input_fc3.bias', (D,)),
        (a + 'self_attn.temporal_attention_before.in_proj_weight', (3 * D, D)),
        (a + 'self_attn.temporal_attention_before.in_proj_bias', (3 * D,)),
        (a + 'self_attn.temporal_attention_before.out_proj.weight', (D, D)),
        (a + 'self_attn.temporal_attention_before.out_proj.bias', (D,)),
        (a + 'self_attn.temporal_info.weight', (D, D)), (a + 'self_attn.temporal_info.bias', (D,)),
        (a + 'self_attn.temporal_gate.weight', (D, D)), (a + 'self_attn.temporal_gate.bias', (D,)),
        (a + 'linear1.weight', (1024, D)), (a + 'linear1.bias', (1024,)),
        (a + 'linear2.weight', (D, 1024)), (a + 'linear2.bias', (D,)),
        (a + 'norm1.weight', (D,)), (a + 'norm1.bias', (D,)),
        (a + 'norm2.weight', (D,)), (a + 'norm2.bias', (D,)),
        (prefix + 'pos_encoder.pe', (200, D)),
        (prefix + 'pos_encoder.fc.weight', (D, 2 * D)), (prefix + 'pos_encoder.fc.bias', (D,)),
    ]


def manifest(past_length=8, future_length=12, hidden_dim=64, zdim=32, n_scales=2, num_decompose=2):
    """Ordered name -> shape map of the STTODENet state_dict for the given hyper-parameters."""
    D = hidden_dim
    items = _trunk('past_encoder.', past_length, D)
    items += [('pz_layer.weight', (2 * zdim, (2 + n_scales) * D)), ('pz_layer.bias', (2 * zdim,))]
    items += _trunk('future_encoder.', future_length, D)
    items += [('future_encoder.out_mlp.affine_layers.0.weight', (128, (2 + n_scales) * D)),
              ('future_encoder.out_mlp.affine_layers.0.bias', (128,)),
              ('future_encoder.qz_layer.weight', (2 * zdim, 128)), ('future_encoder.qz_layer.bias', (2 * zdim,))]
    din = 2 * D + zdim + 96
    for i in range(num_decompose):
        p = f'decoder.decompose.{i}.'
        items += [(p + 'conv_past.weight', (32, 2, 3)), (p + 'conv_past.bias', (32,)),
                  (p + 'encoder_past.weight_ih_l0', (288, 32)), (p + 'encoder_past.weight_hh_l0', (288, 96)),
                  (p + 'encoder_past.bias_ih_l0', (288,)), (p + 'encoder_past.bias_hh_l0', (288,))]
        for nm, dout in (('decoder_y', 2 * future_length), ('decoder_x', 2 * past_length)):
            items += [(p + f'{nm}.layers.0.weight', (512, din)), (p + f'{nm}.layers.0.bias', (512,)),
                      (p + f'{nm}.layers.1.weight', (256, 512)), (p + f'{nm}.layers.1.bias', (256,)),
                      (p + f'{nm}.layers.2.weight', (dout, 256)), (p + f'{nm}.layers.2.bias', (dout,))]
    return OrderedDict(items)


def sinusoid_table_np(max_len, d_model):
    """The pe buffer (model/STTODE.py:149-155): sin/cos positional table, evaluated with torch fp32 ops
    so the values are bit-identical to the buffer the reference registers."""
    import torch
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.numpy()


def make_weights(seed=1234, **hp):
    """Deterministic state_dict (name -> float32 ndarray).  hp: see ``manifest``."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in manifest(**hp).items():
        if name.endswith('pos_encoder.pe'):
            # the reference registers pe as a buffer computed by torch; tests load the reference's own
            # buffer values through load_state_dict, so keep the analytic table here.
            out[name] = sinusoid_table_np(*shape)
            continue
        if '.norm' in name:
            w = (1.0 + 0.1 * rng.standard_normal(shape)) if name.endswith('weight') else 0.05 * rng.standard_normal(shape)
        elif name.endswith('bias') or 'bias_' in name:
            w = 0.05 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 1.6 if 'encoder_past' in name or 'in_proj' in name else 1.0
            a = gain * np.sqrt(3.0 / fan_in)
            w = rng.uniform(-a, a, size=shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def sampler_manifest(nk=20, nz=32, qnet_mlp=(512, 256), pred_model_dim=64):
    """Ordered name -> shape map of the stage-2 Sampler state_dict (sampler.py:7-27, utils/mlp.py:5-22)."""
    items, last = [], pred_model_dim
    for i, nh in enumerate(qnet_mlp):
        items += [(f'q_mlp.affine_layers.{i}.weight', (nh, last)), (f'q_mlp.affine_layers.{i}.bias', (nh,))]
        last = nh
    for nm, (o, i) in (('q_A', (nk * nz, last)), ('q_b', (nk * nz, last)), ('q_c', (nz, nk * nz)), ('linear', (64, 128))):
        items += [(f'{nm}.weight', (o, i)), (f'{nm}.bias', (o,))]
    return OrderedDict(items)


def make_sampler_weights(seed=4321, **hp):
    """Deterministic Sampler state_dict: q_A spread enough that the K latent codes of an agent differ."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in sampler_manifest(**hp).items():
        if name.endswith('bias'):
            w = (0.3 if name.startswith(('q_A', 'q_b')) else 0.05) * rng.standard_normal(shape)
        else:
            a = (2.0 if name.startswith(('q_A', 'q_b')) else 1.0) * np.sqrt(3.0 / shape[1])
            w = rng.uniform(-a, a, size=shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def decoder_layer_manifest(d=64, ff=256):
    """state_dict of the repo's (unused) TransformerDecoderLayer (hypertransformer.py:180-198)."""
    items = []
    for att in ('self_attn', 'cross_attn'):
        p = att + '.temporal_attention_before.'
        items += [(p + 'in_proj_weight', (3 * d, d)), (p + 'in_proj_bias', (3 * d,)), (p + 'out_proj.weight', (d, d)),
                  (p + 'out_proj.bias', (d,))]
        for nm in ('temporal_info', 'temporal_gate'):
            items += [(f'{att}.{nm}.weight', (d, d)), (f'{att}.{nm}.bias', (d,))]
    items += [('linear1.weight', (ff, d)), ('linear1.bias', (ff,)), ('linear2.weight', (d, ff)), ('linear2.bias', (d,))]
    for i in (1, 2, 3):
        items += [(f'norm{i}.weight', (d,)), (f'norm{i}.bias', (d,))]
    return OrderedDict(items)


def make_decoder_layer_weights(seed=61, **hp):
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in decoder_layer_manifest(**hp).items():
        if 'norm' in name and name.endswith('weight'):
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith('bias'):
            w = 0.05 * rng.standard_normal(shape)
        else:
            w = rng.standard_normal(shape) / np.sqrt(shape[-1])
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def to_torch_state_dict(weights):
    import torch
    return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in weights.items())
