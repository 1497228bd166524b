"""The GENERIC-DIMENSION form of the hot path: every hyper-parameter the reference's CLI accepts (train.py:25-26,37-40 --past_length,
--future_length, --zdim, --hidden_dim, --num_decompose; every dimension of model/STTODE.py:182-196,242-260,309-318,359-361 derives from them).

The fused forms of csrc/ (chain launch, lagged roles, one-launch scene form, packed fragment streams) are built for the reference's DEFAULT
widths -- hidden_dim 64, zdim 32, two decompose blocks, 2 Tp <= 32, 2 Tf <= 96 -- because their register / LDS budgets are tuned to those
tiles.  A model constructed with anything else takes this form instead: the same HIP kernels the training step runs (csrc/train.hip:
sttode_tlinear MFMA GEMMs over the row-major nn.Parameter storage, conv / GRU-sequence / LayerNorm / geodesic-attention kernels templated
on the model width), layer by layer, driven by sttode_amd.training.Engine.  Results are held to the imported reference at the same 1e-4
(tests/golden/dims.npz); PERFORMANCE CAVEAT: one launch per layer and activations through HBM -- roughly the training forward's rate, far
from the fused forms' -- stated in DESIGN.md.  No CPU / eager fallback here either: every network evaluation is a HIP kernel.
"""
import torch

from . import capi
from .training import EW_CUR_ADD, EW_RSAMPLE, Engine


def unsupported_reason(args):
    """THE one place that lists what STTODENet refuses, and why (None: supported).  Everything else the reference's Namespace can carry is
    accepted -- by the fused forms at the default widths, by the generic form otherwise (``uses_generic``)."""
    D, zd, nd, Tp, Tf = int(args.hidden_dim), int(args.zdim), int(args.num_decompose), int(args.past_length), int(args.future_length)
    if D not in (32, 64, 128):
        return ('hidden_dim must be 32, 64 or 128: the attention / LayerNorm kernels are instantiated for head widths 4, 8, 16 '
                '(8 heads, model/STTODE.py:188); other multiples of 8 need one more instantiation each')
    if zd < 4 or zd % 4:
        return 'zdim must be a positive multiple of 4 (16-byte pieces of the decoder-input rows)'
    if nd < 1:
        return 'num_decompose must be >= 1'
    if Tp < 2 or Tp > 200 or Tf < 1 or Tf > 200:
        return 'past_length in [2, 200], future_length in [1, 200]: the positional table holds 200 rows (model/STTODE.py:141,149)'
    if len(args.hyper_scales) != 2:
        return ('len(hyper_scales) must be 2: FutureEncoder.out_mlp takes (2 + len(hyper_scales)) * hidden_dim inputs but is fed '
                'cat(past_feature, future_feature) = 4 * hidden_dim (model/STTODE.py:258,297): any other length fails in the reference itself')
    if getattr(args, 'learn_prior', False):
        return 'learn_prior is broken in the reference itself (pz_layer in_features (2 + len(hyper_scales)) * hidden_dim != 2 * hidden_dim, model/STTODE.py:361,603)'
    return None


def uses_generic(args):
    """True: outside the widths the fused forms are built for."""
    return not (int(args.hidden_dim) == 64 and int(args.zdim) == 32 and int(args.num_decompose) == 2 and 2 * int(args.past_length) <= 32
                and 2 * int(args.future_length) <= 96)


def _engine(net):
    eng = getattr(net, '_engine', None)
    if eng is None or eng.dev != net.device:
        eng = net._engine = Engine(net)
        net._graphs, net._graph_seen = {}, set()
    eng.P = {k: v for k, v in net.named_parameters()}
    eng._hold = []
    eng.multi = False
    eng._enter(-1)
    return eng


@torch.no_grad()
def encode(net, vel_from_norm):
    """set_data's derived inputs + PastEncoder.forward (model/STTODE.py:214-236) -> (engine, workspace dict, past_feature [n, 2 D], past [n,Tp,2])."""
    eng = _engine(net)
    a = net.args
    n, Tp = net._past.shape[0], a.past_length
    ws = net._frontend(vel_from_norm=vel_from_norm)
    pf = eng.new(n, eng.PFW)
    eng.trunk_fwd('past_encoder.', ws['enc_in'], ws['last'], pf, None)
    past = ws['xpad'][:, :2 * Tp].reshape(n, Tp, 2).contiguous()
    return eng, ws, pf, past


@torch.no_grad()
def inference(net, z):
    """STTODENet.inference (model/STTODE.py:574-623) -> pred [n, K, Tf, 2] in world coordinates; sets net.past_feature / scene_orig / _ws."""
    a = net.args
    K, Tf = a.sample_k, a.future_length
    eng, ws, pf, past = encode(net, 1)
    n = pf.shape[0]
    d = eng.decoder_fwd(pf, z, K, past, ws['cur'], False)
    pred = d['pred']                                                   # [n K, 2 Tf]: sum of the blocks' y_hat + cur_location
    if net._mode == 'scenes':
        eng.ew(EW_CUR_ADD, pred, ws['orig'], i0=2 * Tf, f0=K)          # + scene_orig (:621-622)
    net.past_feature, net._ws = pf, ws
    return pred.view(n, K, Tf, 2)


@torch.no_grad()
def fu_encoder(net, eps_q, eps_p):
    """model/STTODE.py:498-525 on the generic kernels: FutureEncoder trunk, out_mlp, qz_layer, the posterior draw."""
    eng = _engine(net)
    a, P = net.args, eng.P
    n, Tf, zd = net._past.shape[0], a.future_length, a.zdim
    ws = net._ws
    enc_f = eng.new(n, Tf, 4)
    mode = 0 if net._mode == 'scenes' else 1
    capi.call('sttode_frontend_future', net._future, net._past[:, -1].contiguous(), n, Tf, mode, net._N or 1, ws.get('scene_orig'),
              ws.get('agent_scene'), net._scene_ptr if mode == 0 else None, enc_f, eng.st)
    hcat = eng.new(n, 2 * eng.PFW)
    hcat[:, :eng.PFW] = net.past_feature
    eng.trunk_fwd('future_encoder.', enc_f, ws['last'], hcat[:, eng.PFW:], None)
    hq = eng.lin(hcat, P['future_encoder.out_mlp.affine_layers.0.weight'], P['future_encoder.out_mlp.affine_layers.0.bias'], act='relu')
    net.qz_param = eng.lin(hq, P['future_encoder.qz_layer.weight'], P['future_encoder.qz_layer.bias'])
    net.qz_mu, net.qz_logvar = net.qz_param[:, :zd], net.qz_param[:, zd:]
    eps_q = torch.randn(n, zd, device=net.device) if eps_q is None else torch.as_tensor(eps_q, dtype=torch.float32).to(net.device).contiguous()
    net.qz_sampled = eng.new(n, zd)
    eng.ew(EW_RSAMPLE, net.qz_sampled, net.qz_param, eps_q, i0=zd)
    net.pz_sampled = torch.randn(n, zd, device=net.device) if eps_p is None else torch.as_tensor(eps_p, dtype=torch.float32).to(net.device)
    net.future_traj = net._future - ws['orig'][:, None, :]
    return net.qz_param


@torch.no_grad()
def decode(net, z, K, want_recover):
    """Decoder.forward (model/STTODE.py:320-347) in normalised coordinates -> (pred [n,K,Tf,2], recover [n K,Tp,2] | None)."""
    eng = _engine(net)
    a = net.args
    n = net.past_feature.shape[0]
    z = torch.as_tensor(z, dtype=torch.float32).to(net.device).contiguous()
    d = eng.decoder_fwd(net.past_feature, z, K, net.past_traj.contiguous(), net._ws['cur'], want_recover)
    rec = d['rec'].view(n * K, a.past_length, 2) if want_recover else None
    return d['pred'].view(n, K, a.future_length, 2), rec
