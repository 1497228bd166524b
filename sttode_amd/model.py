"""``STTODENet``: host-side mirror of the reference's model API over the HIP C-ABI library.

Drop-in surface (reference: model/STTODE.py:349-623):
    STTODENet(args, device); set_device(); set_data(batch, pre_motion, fut_motion, pre_mask, fut_mask);
    set_data_nba(data); inference(data) -> Tensor[K, B*N, Tf, 2]; step_annealer();
    attributes agent_num, batch_size, scene_orig, past_feature; identical state_dict names / shapes, so a
    reference checkpoint's ``model_dict`` loads with ``load_state_dict(strict=True)``.
Build-defined extension (the batched scene front-end):
    set_scene_batch(past[n,Tp,2], future[n,Tf,2] | None, scene_ptr[S+1]) + inference()  -- many independent
    scenes per call (the reference loops ``for scene: set_data; inference``, test.py:171-184).

All compute runs in hand-written HIP kernels (sttode_amd/csrc); PyTorch only owns device memory and the
stream.  There is no eager / CPU fallback: a missing library or a CPU tensor raises.
"""
import os
import operator

import numpy as np
import torch
from torch import nn

from .dist import Normal
from . import capi, packing
from .weights import sinusoid_table_np

_DEBUG = os.environ.get('STTODE_DEBUG', '0') not in ('', '0')   # debug mode: validate device-resident inputs too (costs a D2H sync per call)


class _HypMHSA(nn.Module):
    """Parameter holder, names of Hyp_mhsa (hyptransformerlib.py:340-381)."""

    def __init__(self, d, h):
        super().__init__()
        self.embed_dim, self.num_heads = d, h
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _HypAttention(nn.Module):  # hypertransformer.py:19-33
    def __init__(self, d, h):
        super().__init__()
        self.temporal_attention_before = _HypMHSA(d, h)
        self.temporal_info = nn.Linear(d, d)
        self.temporal_gate = nn.Linear(d, d)


class _EncoderLayer(nn.Module):  # hypertransformer.py:108-122
    def __init__(self, d, h, ff):
        super().__init__()
        self.self_attn = _HypAttention(d, h)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class _Named(nn.Module):
    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            setattr(self, k, v)


class _PosEnc(nn.Module):  # model/STTODE.py:137-147
    def __init__(self, d, max_t_len=200):
        super().__init__()
        self.fc = nn.Linear(2 * d, d)
        self.register_buffer('pe', torch.from_numpy(sinusoid_table_np(max_t_len, d)))


class _Trunk(nn.Module):
    """PastEncoder / FutureEncoder parameter tree (model/STTODE.py:178-197, 238-261)."""

    def __init__(self, args, length, future=False):
        super().__init__()
        D = args.hidden_dim
        self.input_fc = nn.Linear(4, D)
        self.input_fc2 = nn.Linear(D * length, D)
        self.input_fc3 = nn.Linear(D + 3, D)
        layers = nn.ModuleList([_EncoderLayer(D, 8, 1024)])
        self.ODE_Encoder = _Named(odeblock=_Named(odefunc=_Named(layers=layers)))
        self.pos_encoder = _PosEnc(D)
        if future:
            self.out_mlp = _Named(affine_layers=nn.ModuleList([nn.Linear((2 + len(args.hyper_scales)) * D, 128)]))
            self.qz_layer = nn.Linear(128, 2 * args.zdim)
            for m in (self.out_mlp.affine_layers[0], self.qz_layer):  # initialize_weights (model/utils.py:11-21)
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)


class _MLP(nn.Module):  # model/utils.py:67-79
    def __init__(self, din, dout, hidden=(512, 256)):
        super().__init__()
        dims = [din, *hidden, dout]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))


class _DecomposeBlock(nn.Module):  # model/STTODE.py:20-48
    def __init__(self, past_len, future_len, input_dim):
        super().__init__()
        self.conv_past = nn.Conv1d(2, 32, 3, stride=1, padding=1)
        self.encoder_past = nn.GRU(32, 96, 1, batch_first=True)
        self.decoder_y = _MLP(96 + input_dim, future_len * 2)
        self.decoder_x = _MLP(96 + input_dim, past_len * 2)
        nn.init.kaiming_normal_(self.conv_past.weight)
        nn.init.kaiming_normal_(self.encoder_past.weight_ih_l0)
        nn.init.kaiming_normal_(self.encoder_past.weight_hh_l0)
        nn.init.zeros_(self.conv_past.bias)
        nn.init.zeros_(self.encoder_past.bias_ih_l0)
        nn.init.zeros_(self.encoder_past.bias_hh_l0)


class _Decoder(nn.Module):  # model/STTODE.py:303-318
    def __init__(self, args):
        super().__init__()
        din = 2 * args.hidden_dim + args.zdim
        self.decompose = nn.ModuleList(_DecomposeBlock(args.past_length, args.future_length, din)
                                       for _ in range(args.num_decompose))


_DATA_PTR = torch.Tensor.data_ptr
_VERSION = operator.attrgetter('_version')


def _on(t, device):
    device = torch.device(device)
    return t.device.type == device.type and (device.index is None or t.device.index == device.index)


def _f32(t, device):
    if isinstance(t, torch.Tensor) and t.dtype == torch.float32 and _on(t, device) and t.is_contiguous():
        return t                                     # the per-scene loop passes tensors that are already in place
    return torch.as_tensor(t, dtype=torch.float32).to(device).contiguous()


class _LazyViews(dict):
    """name -> workspace view, materialised on first access (a per-scene inference() call should not pay for ten tensor views
    nobody reads)."""

    def __init__(self, makers):
        super().__init__()
        self._makers = makers

    def __missing__(self, key):
        if key not in self._makers:
            raise KeyError(key)
        v = self[key] = self._makers[key]()
        return v

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


class STTODENet(nn.Module):
    ODE_TIME = 12.0  # ODEG_Encoder(encoder_layers, nlayer, 12): model/STTODE.py:195 -> one Euler step of size 12
    ODE_METHODS = {'euler': 0, 'rk4': 1, 'rk4_classic': 2}

    def __init__(self, args, device):
        super().__init__()
        # what is refused, and why, is listed in ONE place: generic.unsupported_reason.  Everything else the reference's CLI accepts
        # (train.py:25-26,37-40) is taken: by the fused forms at the reference's default widths, by the generic form otherwise
        from . import generic
        why = generic.unsupported_reason(args)
        if why is not None:
            raise NotImplementedError(why)
        self._generic = generic.uses_generic(args)
        self.device = torch.device(device)
        self.args = args
        self.max_train_agent = args.max_train_agent
        self.rand_rot_scene = args.rand_rot_scene
        self.discrete_rot = args.discrete_rot
        scale_num = 2 + len(args.hyper_scales)
        self.past_encoder = _Trunk(args, args.past_length)
        self.pz_layer = nn.Linear(scale_num * args.hidden_dim, 2 * args.zdim)
        self.future_encoder = _Trunk(args, args.future_length, future=True)
        self.decoder = _Decoder(args)
        self.param_annealers = nn.ModuleList()
        self._packed = None
        self._packed_key = None
        self._mode = None
        self._async_calls = 0
        # integrator of the tensor-ODE encoder: the reference runs ONE Euler step (ode_demo.py:186-190) = ('euler', 1); 'rk4' is
        # torchdiffeq's fixed-grid rk4 (3/8 rule), 'rk4_classic' the classical one; steps = uniform steps over [0, 12] (oracle-checked only)
        self.ode_method, self.ode_steps = 'euler', 1
        # EXPLORATORY, opt-in: 'bf16x3' runs the two block-0 decoder MLPs of the fused launch as a three-way bf16 split on the bf16 matrix
        # cores (fp32-class accuracy, fp32 accumulate; csrc/chain32.hip B3M); 'f32' (default) = fp32 MFMA everywhere.  env STTODE_BF16X3=1
        self.mfma_mode = 'bf16x3' if os.environ.get('STTODE_BF16X3', '0') not in ('', '0') else 'f32'
        self._host_futures = {}  # slot -> pinned [n, K, Tf, 2] (futures_to_host_async)
        self.async_depth = 6     # calls in flight of the inference_async pipeline (workspace / prediction slots, <= 8; 2 x pipeline streams)
        # inference_async(z=None) in the lagged form: latents drawn by the call's own launch (csrc/role32.hpp); env STTODE_DEVICE_LATENTS=0: torch.randn
        self.device_latents = os.environ.get('STTODE_DEVICE_LATENTS', '1') != '0'
        self._async_bufs = {}
        self._async_metrics = {}
        self._ext_streams = {}
        self._ptr_cache = {}
        self._pf = self._pf_thunk = None
        self.to(self.device)

    # ------------------------------------------------------------------ plumbing
    @property
    def past_feature(self):
        """[n, 128] encoder output (model/STTODE.py:496); after inference() it is a view into the workspace, made on first read."""
        if self._pf is None and self._pf_thunk is not None:
            self._pf, self._pf_thunk = self._pf_thunk(), None
        return self._pf

    @past_feature.setter
    def past_feature(self, value):
        self._pf, self._pf_thunk = value, None

    # per-call state (tensors of the current batch, views of the workspace): plain attributes.  nn.Module.__setattr__ walks its
    # Parameter / Module / buffer checks on every assignment (~1.5 us each, ~17 per one-scene call of the evaluation loop)
    _PLAIN = frozenset(('_past', '_future', '_scene_ptr', '_mode', 'batch_size', 'agent_num', '_S', '_N', '_G', 'pre_motion_mask', 'fut_motion_mask',
                        'scene_orig', '_pf', '_pf_thunk', '_ws', '_dbg', 'diverse_pred', '_plist'))

    def __setattr__(self, name, value):
        if name in STTODENet._PLAIN:
            self.__dict__[name] = value
        else:
            super().__setattr__(name, value)

    def set_device(self, device):
        self.device = torch.device(device)
        self.to(self.device)

    def step_annealer(self):  # model/STTODE.py:570-572 (no annealers are ever registered)
        for anl in self.param_annealers:
            anl.step()

    def _weights_key(self):
        if getattr(self, '_plist', None) is None:
            self._plist = list(self.parameters()) + list(self.buffers())
        # (C-level maps: this runs on every call of the one-scene evaluation loop)
        return (self.device.index, tuple(map(_DATA_PTR, self._plist)), tuple(map(_VERSION, self._plist)))

    def _apply(self, fn, *a, **k):
        self._plist = None  # .to()/.cuda() may replace parameter storage
        return super()._apply(fn, *a, **k)

    def packed(self):
        """Fragment-ordered device copies of the weights; re-packed whenever a parameter changes."""
        if self._generic:
            return {}                                    # the generic form reads the nn.Parameter storage directly (generic.py)
        key = self._weights_key()
        if self._packed is None or key != self._packed_key:
            if getattr(self, '_native', None) is not None:
                # calls in flight (lagged form: groups still outstanding) read the old packed weights and the workspaces dropped below
                capi.call('sttode_async_flush', self._native.h)
                torch.cuda.synchronize(self.device)
            sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
            a = self.args
            host = {'past': packing.pack_trunk(sd, 'past_encoder.', a.past_length),
                    'blk0': packing.pack_block(sd, 0, a.past_length, a.future_length, first=True),
                    'blk1': packing.pack_block(sd, 1, a.past_length, a.future_length, first=False),
                    'future': packing.pack_trunk(sd, 'future_encoder.', a.future_length),
                    'post': packing.pack_posterior(sd),
                    'chain': packing.chain_stream(sd, a.past_length, a.future_length),
                    'gru0s': packing.gru32_stream(sd, 0, a.past_length),
                    'role32': packing.role_stream(sd, a.past_length)}
            self._packed = {g: {k: (torch.from_numpy(np.ascontiguousarray(v)).to(self.device) if isinstance(v, np.ndarray) else v)
                                for k, v in d.items()} for g, d in host.items()}
            self._packed_key = key
            self._native = capi.NativeModel(self._packed, a.past_length, a.future_length, a.sample_k) if self.device.type == 'cuda' else None
            self._wscache = {}
            self._async_bufs = {}
        return self._packed

    def release_native(self):
        """Drop the native pipeline handle (its events; the pipeline's streams are process-wide), the packed weights and the cached
        workspaces; all are rebuilt on the next call."""
        if self.device.type == 'cuda':
            if getattr(self, '_native', None) is not None:
                capi.call('sttode_async_flush', self._native.h)
            torch.cuda.synchronize(self.device)
        self._native, self._packed, self._packed_key = None, None, None
        self._wscache, self._async_bufs = {}, {}

    def native(self, check_weights=True):
        """The native pipeline handle on the current weights.  ``check_weights=False``: skip the weight-version comparison (10 us over 88
        parameters) -- for callers that make it themselves AFTER their launch is enqueued and launch again if it fails (inference())."""
        if self._generic:
            raise capi.SttodeError('the native pipeline (fused launches, packed weight streams) is built for the reference widths hidden_dim 64, '
                                   'zdim 32, num_decompose 2, 2 Tp <= 32, 2 Tf <= 96; this model runs the generic form (sttode_amd/generic.py)')
        if check_weights or self._packed is None or self._native is None:
            self.packed()
        ode = (self.ODE_METHODS[self.ode_method], int(self.ode_steps))
        if self._native is not None and getattr(self._native, '_ode', (0, 1)) != ode:
            self._native.set_ode(*ode)
            self._native._ode = ode
        if self._native is not None and getattr(self._native, '_mfma', None) != self.mfma_mode:
            if self.mfma_mode not in ('f32', 'bf16x3'):
                raise ValueError("mfma_mode must be 'f32' or 'bf16x3'")
            if self.mfma_mode == 'bf16x3' and 'chain_b3' not in self._packed:
                # the opt-in mode's weight stream (1.5x the fp32 stream, ~20 k small NumPy ops) is packed on FIRST use, not with every weight set
                sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
                b3 = packing.chain_stream_b3(sd, self.args.past_length, self.args.future_length)
                self._native.set_weights('chain_b3', {k: (torch.from_numpy(np.ascontiguousarray(v)).to(self.device) if isinstance(v, np.ndarray) else v)
                                                      for k, v in b3.items()})
            self._native.set_mfma_mode(1 if self.mfma_mode == 'bf16x3' else 0)
            self._native._mfma = self.mfma_mode
        return self._native

    def _workspace(self, n, S):
        """Cached workspace tensor + named views (layout from sttode_workspace_layout)."""
        key = (n, S)
        if key not in self._wscache:
            if len(self._wscache) > 8:
                self._wscache.clear()
            off, tot = self._native.layout(n, S)
            buf = torch.empty(tot, dtype=torch.float32, device=self.device)
            self._native.init_workspace(buf, n, S)        # once per workspace: the hand-off flag words start (and are kept) zero
            self._wscache[key] = (buf, off)
        return self._wscache[key]

    def _view(self, buf, off, name, *shape, dtype=torch.float32):
        cnt = int(np.prod(shape))
        v = buf[off[name]: off[name] + cnt]
        if dtype != torch.float32:
            v = v.view(dtype)
        return v.view(*shape)

    def _require_gpu(self):
        if self.device.type != 'cuda':
            raise capi.SttodeError('STTODENet compute runs only on a HIP device (no CPU fallback); got device=%s' % self.device)

    # ------------------------------------------------------------------ data entry
    def set_data(self, batch, pre_motion, fut_motion, pre_motion_mask=None, fut_motion_mask=None, theta=None):
        """model/STTODE.py:397-461: one scene, pre_motion [N,2,Tp], fut_motion [N,2,Tf] (loader layout).
        ``batch`` is ignored, as in the reference.  In ``train()`` mode the reference's augmentation applies (:405-426):
        random sub-sampling to ``max_train_agent`` agents (np.random.choice, with replacement) and a random rotation of the
        scene about ``scene_orig`` (``theta`` may be injected; otherwise torch.rand(1)*2pi, or a multiple of pi/12 when
        ``discrete_rot``).  This is data preparation on a [N, T, 2] track, done with torch ops before the kernels run."""
        dev = self.device
        # the evaluation loop's call (test.py:171-188: eval mode, the loader's float32 host tensors): staged natively, attributes set directly --
        # every microsecond here is on the loop's critical path (the GPU idles until inference() has enqueued its launch)
        if (not self.training and theta is None and type(pre_motion) is torch.Tensor and type(fut_motion) is torch.Tensor
                and pre_motion.dtype is torch.float32 and fut_motion.dtype is torch.float32 and not pre_motion.is_cuda and not fut_motion.is_cuda
                and dev.type == 'cuda' and pre_motion.dim() == 3 and pre_motion.shape[1] == 2 and pre_motion.shape[2] == self.args.past_length
                and pre_motion.shape[0] > 0 and fut_motion.shape == (pre_motion.shape[0], 2, self.args.future_length)):
            N = pre_motion.shape[0]
            past, fut = self._stage_scene(pre_motion, fut_motion)
            ptr = self._ptr_cache.get(N)
            if ptr is None or not _on(ptr, dev):
                ptr = self._ptr_cache[N] = torch.tensor([0, N], dtype=torch.int32).to(dev)
            d = self.__dict__
            d['_past'], d['_future'], d['_scene_ptr'], d['_mode'], d['batch_size'], d['agent_num'], d['_S'], d['_N'] = past, fut, ptr, 'scenes', 1, N, 1, 0
            d['_G'] = 1
            d['pre_motion_mask'], d['fut_motion_mask'] = pre_motion_mask, fut_motion_mask
            return

        def to_dev(x):      # [N, 2, T] loader layout -> [N, T, 2]; a host tensor is transposed on the host (one H2D copy, no kernel)
            x = torch.as_tensor(x, dtype=torch.float32)
            return x.permute(0, 2, 1).contiguous().to(dev)
        pre_motion = torch.as_tensor(pre_motion, dtype=torch.float32)
        fut_motion = torch.as_tensor(fut_motion, dtype=torch.float32) if fut_motion is not None else None
        if torch.device(dev).type == 'cuda' and not pre_motion.is_cuda and (fut_motion is None or not fut_motion.is_cuda):
            past, fut = self._stage_scene(pre_motion, fut_motion)
        else:
            past = to_dev(pre_motion)
            fut = to_dev(fut_motion) if fut_motion is not None else None
        if self.training and past.shape[0] > self.max_train_agent:
            ind = torch.tensor(np.random.choice(past.shape[0], self.max_train_agent).tolist(), device=dev)
            past = past.index_select(0, ind).contiguous()
            fut = fut.index_select(0, ind).contiguous() if fut is not None else None
            pre_motion_mask = pre_motion_mask.to(dev).index_select(0, ind) if pre_motion_mask is not None else None
            fut_motion_mask = fut_motion_mask.to(dev).index_select(0, ind) if fut_motion_mask is not None else None
        if (self.training and self.rand_rot_scene) or theta is not None:
            if theta is None:
                theta = (torch.randint(high=24, size=(1,)) * (np.pi / 12)) if self.discrete_rot else torch.rand(1) * np.pi * 2
            th = float(theta)
            c, s_ = float(np.cos(np.float32(th))), float(np.sin(np.float32(th)))
            if past.is_cuda:                                     # one small kernel (a dozen torch ops were ~0.1 ms of host time per training step)
                past = past.contiguous()                         # (fresh [N,T,2] copies of the loader's [N,2,T] tracks: rotated in place)
                fut = fut.contiguous() if fut is not None else None
                capi.call('sttode_rotate_scene', past, fut, past.shape[0], past.shape[1], fut.shape[1] if fut is not None else 0, c, s_,
                          capi.stream_ptr())
            else:
                R = torch.tensor([[c, -s_], [s_, c]], dtype=torch.float32, device=dev)        # rotation_2d_torch (:6-14)
                orig = past[:, -1].mean(dim=0)                       # scene_orig (:417); invariant under the rotation about itself
                rot = lambda x: (((x - orig).unsqueeze(-2) * R).sum(-1) + orig).contiguous()  # x'_i = sum_j R[i][j] (x - orig)_j + orig_i
                past = rot(past)
                fut = rot(fut) if fut is not None else None
        N = past.shape[0]
        if N == 0:
            raise ValueError('empty scene')
        ptr = self._ptr_cache.get(N)                             # device-resident [0, N] CSR, built once per scene size
        if ptr is None or not _on(ptr, dev):
            ptr = self._ptr_cache[N] = torch.tensor([0, N], dtype=torch.int32).to(dev)
        self.set_scene_batch(past, fut, ptr)
        self.batch_size = 1
        self.pre_motion_mask, self.fut_motion_mask = pre_motion_mask, fut_motion_mask

    def _stage_scene(self, pre, fut):
        """Host tensors of one scene (loader layout [N,2,T]) -> device [N,T,2] tensors through ONE asynchronous H2D copy: both tracks
        are transposed into a pinned staging buffer (a ring of four, each guarded by an event; native) and travel together.  ``.to(device)`` of
        a pageable tensor is a synchronous copy that also waits for the stream's earlier kernels -- two of them per scene were a third
        of the per-scene loop of test.py:171-188."""
        N, Tp = pre.shape[0], pre.shape[2]
        Tf = fut.shape[2] if fut is not None else 0
        if N == 0:
            raise ValueError('empty scene')
        need = N * (Tp + Tf) * 2
        dev = torch.empty(need, dtype=torch.float32, device=self.device)
        # transposes into a pinned ring slot + ONE asynchronous H2D copy, natively (csrc/frontend.hip: sttode_stage_scene)
        if capi.TIMING is None and fut is not None:
            if capi.lib().sttode_stage_scene(pre.contiguous().data_ptr(), fut.contiguous().data_ptr(), N, Tp, Tf, dev.data_ptr(), capi.stream_ptr()):
                raise capi.SttodeError('sttode_stage_scene failed: ' + capi.lib().sttode_last_error().decode())
        else:
            capi.call('sttode_stage_scene', pre.contiguous(), fut.contiguous() if fut is not None else None, N, Tp, Tf, dev, capi.stream_ptr())
        return dev[:N * Tp * 2].view(N, Tp, 2), (dev[N * Tp * 2:].view(N, Tf, 2) if fut is not None else None)

    def set_scene_batch(self, past, future, scene_ptr):
        """Many independent scenes: past [n,Tp,2] / future [n,Tf,2] world coordinates, agent-major;
        scene_ptr [S+1] CSR offsets (== seq_start_end of utils/dataloader.py:177-181)."""
        a, dev = self.args, self.device
        self._past = _f32(past, dev)
        self._future = _f32(future, dev) if future is not None else None
        if isinstance(scene_ptr, torch.Tensor) and scene_ptr.dtype == torch.int32 and _on(scene_ptr, dev) and scene_ptr.is_contiguous():
            self._scene_ptr = scene_ptr
        else:
            self._scene_ptr = torch.as_tensor(scene_ptr, dtype=torch.int32).to(dev).contiguous()
        if self._past.dim() != 3 or self._past.shape[1] != a.past_length or self._past.shape[2] != 2:
            raise ValueError(f'past must be [n, {a.past_length}, 2], got {tuple(self._past.shape)}')
        if self._past.shape[0] == 0 or self._scene_ptr.numel() < 2:
            raise ValueError('empty batch: need at least one scene with at least one agent')
        if not (isinstance(scene_ptr, torch.Tensor) and scene_ptr.is_cuda) or _DEBUG:
            # host-side CSR is validated here; a device-resident CSR is trusted (validating it would force a D2H sync
            # per call -- callers on the hot loop keep their batches resident, bench.py) unless STTODE_DEBUG=1
            sp = torch.as_tensor(scene_ptr).cpu()
            if int(sp[0]) != 0 or int(sp[-1]) != self._past.shape[0] or bool((sp[1:] <= sp[:-1]).any()):
                raise ValueError('scene_ptr must start at 0, end at n and be strictly increasing (no empty scenes)')
        self._mode = 'scenes'
        self._G = 1
        self.batch_size = 1
        self.agent_num = self._past.shape[0]
        self._S = self._scene_ptr.numel() - 1
        self._N = 0

    def set_data_nba(self, data):
        """model/STTODE.py:463-486: dict with past_traj [B,N,Tp,2], future_traj [B,N,Tf,2].
        Build-defined: past_traj [G,B,N,Tp,2] (future_traj [G,B,N,Tf,2]) = G forward-call batches in ONE call -- the attention runs within each
        batch of B scenes (what G reference calls compute, test.py:520-524), everything else over all G B N agents (include/sttode_hip.h
        sttode_inference_nba_groups); ``batch_size`` / ``agent_num`` stay B / N, results come back for G B N agents in (g, b, n) order."""
        a, dev = self.args, self.device
        pt, ft = data['past_traj'], (data.get('future_traj') if hasattr(data, 'get') else None)
        if (dev.type == 'cuda' and type(pt) is torch.Tensor and pt.dtype is torch.float32 and not pt.is_cuda and pt.is_contiguous() and pt.numel() > 0
                and (ft is None or (type(ft) is torch.Tensor and ft.dtype is torch.float32 and not ft.is_cuda and ft.is_contiguous()))
                and dev.index in (None, torch.cuda.current_device())):
            # the loader's pageable host tensors (train.py:61): both through a pinned ring slot and ONE asynchronous copy (csrc/frontend.hip
            # sttode_stage_rows) -- `.to(device)` of a pageable tensor first waits for everything queued on the stream, i.e. for the previous
            # training step's whole backward pass and optimizer step
            na, nb = pt.numel(), (ft.numel() if ft is not None else 0)
            na4 = (na + 3) // 4 * 4
            buf = torch.empty(na4 + nb, dtype=torch.float32, device=dev)
            capi.call('sttode_stage_rows', pt, na, ft, nb, buf, capi.stream_ptr())
            pt, ft = buf[:na].view(pt.shape), (buf[na4:].view(ft.shape) if ft is not None else None)
        else:
            pt = _f32(pt, dev)
        self.data = data
        self._G = 1
        if pt.dim() == 5:
            self._G = pt.shape[0]
            pt = pt.reshape(-1, *pt.shape[2:])
        self.batch_size, self.agent_num = pt.shape[0] // self._G, pt.shape[1]
        self._past = pt.reshape(pt.shape[0] * self.agent_num, a.past_length, 2).contiguous()
        self._future = _f32(ft, dev).reshape(-1, a.future_length, 2).contiguous() if ft is not None else None
        self._mode = 'nba'
        self._N = self.agent_num
        self.scene_orig = self._past  # sic (model/STTODE.py:473)

    # ------------------------------------------------------------------ compute (staged API, kernel by kernel)
    def _f(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def _frontend(self, vel_from_norm):
        a, dev = self.args, self.device
        n, Tp = self._past.shape[0], a.past_length
        TPX = (2 * Tp + 15) // 16
        ws = {'xpad': self._f(n, 16 * TPX), 'enc_in': self._f(n, Tp, 4), 'cur': self._f(n, 2), 'orig': self._f(n, 2),
              'last': torch.empty(n, dtype=torch.int32, device=dev)}
        st = capi.stream_ptr()
        if self._mode == 'scenes':
            ws['scene_orig'] = self._f(self._S, 2)
            ws['agent_scene'] = torch.empty(n, dtype=torch.int32, device=dev)
            capi.call('sttode_frontend_scenes', self._past, self._scene_ptr, n, self._S, Tp, TPX, int(vel_from_norm), ws['scene_orig'],
                      ws['agent_scene'], ws['xpad'], ws['enc_in'], ws['cur'], ws['orig'], ws['last'], st)
            self.scene_orig = ws['scene_orig'][0] if self._S == 1 else ws['scene_orig']
        else:
            capi.call('sttode_frontend_nba', self._past, n, self._N, Tp, TPX, ws['xpad'], ws['enc_in'], ws['cur'], ws['orig'],
                      ws['last'], st)
        return ws

    def _encode(self, W, enc_in, last, Tlen):
        """Trunk forward (PastEncoder / FutureEncoder shared part): enc_in [n,Tlen,4] -> [n,128]."""
        n = enc_in.shape[0]
        L, Nslots = (self.batch_size, self._N) if self._mode == 'nba' else (1, 1)
        st = capi.stream_ptr()
        g, qkv = self._f(n, 64), self._f(n, 192)
        capi.call('sttode_embed_qkv', W['fc1P'], W['fc1b'], W['posP'], W['peb'], W['fc2P'], W['fc2b'], W['fc3P'], W['fc3b'],
                  W['fc3last'], W['inP'], W['inb'], enc_in, last, g, qkv, n, Tlen, st)
        if L > 1:
            # self-attention with L == S: scores are used untransposed (hyptransformerlib.py:261-265), i.e.
            # rows = keys, columns = queries, values indexed by the column:  out_i = sum_j softmax_j(-d(k_i, q_j)) v_j
            attn = self._f(n, 64)
            e = qkv.element_size()
            G = getattr(self, '_G', 1)                    # attention groups of the call (set_data_nba with [G,B,N,...]); group stride = B N rows
            capi.call('sttode_mhgsa_attn_groups', qkv.data_ptr() + 64 * e, qkv.data_ptr(), qkv.data_ptr() + 128 * e, attn, G, L * Nslots * 192,
                      L * Nslots * 192, L * Nslots * 192, L * Nslots * 64, L, L, Nslots, Nslots * 192, 192, Nslots * 192, 192, Nslots * 192, 192,
                      Nslots * 64, 64, 1.0, 8.0 ** -0.5, 8, st)
            attn_ptr, ld = attn, 64
        else:
            attn_ptr, ld = qkv.data_ptr() + 128 * qkv.element_size(), 192  # softmax over one element == 1  =>  output == v
        pf = self._f(n, 128)
        if (self.ode_method, self.ode_steps) != ('euler', 1):
            if L > 1 and getattr(self, '_G', 1) > 1:
                raise NotImplementedError('staged API: non-default integrators with several attention groups per call go through inference()')
            if L > 1:
                # every stage is a pass over the whole attention group: in-projection of the state -> geodesic attention -> f(y); the same
                # stage algebra the native pipeline enqueues (csrc/pipeline.hip stage_agents) and hypertransformer.ode_integrate spells out
                from .hypertransformer import ode_integrate
                e = qkv.element_size()

                def rhs(y):
                    q2, a2, k = self._f(n, 192), self._f(n, 64), self._f(n, 64)
                    capi.call('sttode_linear_cols', y, 64, 64, None, 0, 0, W['inP'], W['inb'], q2, 192, n, 192, 0, st)
                    capi.call('sttode_mhgsa_attn', q2.data_ptr() + 64 * e, q2.data_ptr(), q2.data_ptr() + 128 * e, a2, None, None, L, L,
                              Nslots, Nslots * 192, 192, Nslots * 192, 192, Nslots * 192, 192, Nslots * 64, 64, 1.0, 8.0 ** -0.5, st)
                    capi.call('sttode_post_attn_rhs', W['outP'], W['outb'], W['infoP'], W['infob'], W['gateP'], W['gateb'], W['ln1w'], W['ln1b'],
                              W['l1P'], W['l1b'], W['l2P'], W['l2b'], W['ln2w'], W['ln2b'], y, a2, 64, k, n, st)
                    self._keep_ode = (q2, a2)
                    return k
                yT = ode_integrate(rhs, g, self.ODE_TIME, self.ode_method, int(self.ode_steps))
                capi.call('sttode_ode_state_to_pf', g, yT, pf, n, st)
                self._keep = (g, qkv)
                return pf
            capi.call('sttode_post_attn_ode', W['outP'], W['outb'], W['infoP'], W['infob'], W['gateP'], W['gateb'], W['ln1w'], W['ln1b'],
                      W['l1P'], W['l1b'], W['l2P'], W['l2b'], W['ln2w'], W['ln2b'], W['inP'], W['inb'], g, pf, n, self.ODE_TIME,
                      self.ODE_METHODS[self.ode_method], int(self.ode_steps), st)
        else:
            capi.call('sttode_post_attn', W['outP'], W['outb'], W['infoP'], W['infob'], W['gateP'], W['gateb'], W['ln1w'], W['ln1b'],
                      W['l1P'], W['l1b'], W['l2P'], W['l2b'], W['ln2w'], W['ln2b'], g, attn_ptr, ld, pf, n, self.ODE_TIME, st)
        self._keep = (g, qkv)
        return pf

    @torch.no_grad()
    def encode_history(self):
        """model/STTODE.py:488-496 (self.inputs uses velocities of the un-normalised track, :432-433,456)."""
        self._require_gpu()
        if self._generic:
            from . import generic
            _, self._ws, self.past_feature, self.past_traj = generic.encode(self, 0)
            self.cur_location = self.past_traj[:, -1:]
            return self.past_feature
        P = self.packed()
        self._ws = self._frontend(vel_from_norm=0)
        self.past_feature = self._encode(P['past'], self._ws['enc_in'], self._ws['last'], self.args.past_length)
        n, Tp = self._past.shape[0], self.args.past_length
        self.past_traj = self._ws['xpad'][:, :2 * Tp].reshape(n, Tp, 2)
        self.cur_location = self.past_traj[:, -1:]
        return self.past_feature

    @torch.no_grad()
    def fu_encoder(self, eps_q=None, eps_p=None):
        """model/STTODE.py:498-525: posterior q(z | past, future) and prior samples."""
        if self._future is None:
            raise capi.SttodeError('fu_encoder needs the future (set_data / set_scene_batch with future, or set_data_nba)')
        if self._generic:
            from . import generic
            return generic.fu_encoder(self, eps_q, eps_p)
        a, P = self.args, self.packed()
        n, Tf = self._past.shape[0], a.future_length
        st = capi.stream_ptr()
        enc_f = self._f(n, Tf, 4)
        last_past = self._past[:, -1].contiguous()
        mode = 0 if self._mode == 'scenes' else 1
        capi.call('sttode_frontend_future', self._future, last_past, n, Tf, mode, self._N or 1, self._ws.get('scene_orig'),
                  self._ws.get('agent_scene'), self._scene_ptr if mode == 0 else None, enc_f, st)
        ff = self._encode(P['future'], enc_f, self._ws['last'], Tf)
        h = self._f(n, 128)
        capi.call('sttode_linear_cols', self.past_feature, 128, 128, ff, 128, 128, P['post']['outP'], P['post']['outb'], h, 128, n, 128, 1, st)
        self.qz_param = self._f(n, 2 * a.zdim)
        capi.call('sttode_linear_cols', h, 128, 128, None, 0, 0, P['post']['qzP'], P['post']['qzb'], self.qz_param, 2 * a.zdim, n,
                  2 * a.zdim, 0, st)
        self.qz_mu, self.qz_logvar = self.qz_param[:, :a.zdim], self.qz_param[:, a.zdim:]
        eps_q = torch.randn(n, a.zdim, device=self.device) if eps_q is None else _f32(eps_q, self.device)
        self.qz_sampled = self._f(n, a.zdim)                                                    # Normal.rsample, :89-93
        capi.call('sttode_train_ewise', 5, self.qz_sampled, self.qz_param, eps_q.contiguous(), None, None, n * a.zdim, a.zdim, 0.0, st)
        self.pz_sampled = torch.randn(n, a.zdim, device=self.device) if eps_p is None else _f32(eps_p, self.device)
        orig = self._ws['orig']
        self.future_traj = self._future - orig[:, None, :]
        return self.qz_param

    def _decode(self, pf, z, ws, K, orig=None, want_recover=False):
        """Decoder.forward (model/STTODE.py:320-347) kernel by kernel; returns pred [n,K,Tf,2] (+ recover [n*K,Tp,2])."""
        a = self.args
        P = self.packed()
        b0, b1 = P['blk0'], P['blk1']
        n, Tp, Tf = pf.shape[0], a.past_length, a.future_length
        TPX, NOY = packing.tiles_x(Tp), packing.tiles_y(Tf)
        m = n * K
        st = capi.stream_ptr()
        z = _f32(z, self.device)
        orig = ws['orig'] if orig is None else orig
        state0 = self._f(n, 96)
        capi.call('sttode_gru_cols', ws['xpad'], b0['convP'], b0['convB'], b0['wihP'], b0['whhP'], b0['gbias'], state0, n, Tp, TPX, st)
        A0x, A0y, A1y = self._f(n, 512), self._f(n, 512), self._f(n, 512)
        capi.call('sttode_agent_preact', pf, state0, b0['x_WA'], b0['x_b1'], b0['y_WA'], b0['y_b1'], b1['y_WA'], b1['y_b1'], A0x, A0y, A1y, n, st)
        dbuf, ybuf = self._f(m, 16 * TPX), self._f(m, 16 * NOY)
        capi.call('sttode_mlp_block0', A0x, A0y, b0['stream'], b0['n_chunks'], z, ws['xpad'], dbuf, ybuf, m, K, TPX, NOY, st)
        state1 = self._f(m, 96)
        capi.call('sttode_gru_cols', dbuf, b1['convP'], b1['convB'], b1['wihP'], b1['whhP'], b1['gbias'], state1, m, Tp, TPX, st)
        pred = self._f(n, K, Tf, 2)
        capi.call('sttode_mlp_block1', A1y, b1['stream'], b1['n_chunks'], z, state1, ybuf, ws['cur'], orig, pred, m, K, Tf,
                  NOY, st)
        self._dbg = dict(state0=state0, dbuf=dbuf, ybuf=ybuf, state1=state1)
        if not want_recover:
            return pred
        # reconstruction = x_hat0 + x_hat1 (the last block's decoder_x is live in training, :337-341)
        A1x = self._f(n, 512)
        capi.call('sttode_linear_cols', pf, 128, 128, None, 0, 0, b1['x_WA'], b1['x_b1'], A1x, 512, n, 512, 0, st)
        xh1 = self._f(m, 16 * TPX)
        capi.call('sttode_mlp_cols', A1x, b1['x_stream'], b1['x_n_chunks'], z, state1, xh1, m, K, TPX, st)
        x_true = ws['xpad'].repeat_interleave(K, dim=0) if K > 1 else ws['xpad']
        x_hat0 = x_true - dbuf
        recover = (x_hat0 + xh1)[:, :2 * Tp].reshape(m, Tp, 2)
        return pred, recover

    @torch.no_grad()
    def decoder_future_0(self, qz_sampled, eps20=None):
        """model/STTODE.py:534-551: K = 1 decode with the posterior sample (normalised coordinates), then draws the 20 prior samples."""
        a = self.args
        if self._generic:
            from . import generic
            pred, rec = generic.decode(self, qz_sampled, 1, True)
        else:
            zeros = torch.zeros_like(self._ws['orig'])
            pred, rec = self._decode(self.past_feature, qz_sampled, self._ws, 1, orig=zeros, want_recover=True)
        self.pred_traj = pred.reshape(pred.shape[0], a.future_length, 2)
        self.recover_traj = rec
        n = self.past_feature.shape[0]
        self.past_feature_repeat = self.past_feature.repeat_interleave(20, dim=0)
        self.pz_sampled = torch.randn(n * 20, a.zdim, device=self.device) if eps20 is None else _f32(eps20, self.device)
        zero = torch.zeros(n * 20, a.zdim, device=self.device)
        self.pz_dis = Normal(mu=zero, logvar=zero)          # N(0, I) prior over the 20 samples (:541-551)

    @torch.no_grad()
    def decoder_future_1(self, pz_sampled):
        """model/STTODE.py:529-532: K = 20 decode with prior samples -> diverse_pred_traj [n,20,Tf,2] (normalised)."""
        if self._generic:
            from . import generic
            self.diverse_pred_traj = generic.decode(self, pz_sampled, 20, False)[0]
        else:
            zeros = torch.zeros_like(self._ws['orig'])
            self.diverse_pred_traj = self._decode(self.past_feature, pz_sampled, self._ws, 20, orig=zeros)
        self.attn_weights = None

    def forward(self, eps_q=None, eps_p=None, eps20=None, drop_past=None, drop_future=None):
        """Training objective (model/STTODE.py:553-568, losses :372-395): (total_loss, loss_pred, loss_recover, loss_kl,
        loss_diverse).  With autograd enabled (the train.py case) the step runs on the training kernels of csrc/train.hip and
        ``total_loss.backward()`` fills ``.grad`` of every parameter (sttode_amd/training.py); under ``torch.no_grad()`` the
        same values come from the fused inference kernels.  ``eps_*`` / ``drop_*`` inject the noises / dropout masks the
        reference draws from torch's generator (Normal.rsample, nn.Dropout(0.1) of the positional encoders)."""
        self._require_gpu()
        if getattr(self, '_G', 1) > 1:
            raise NotImplementedError('forward(): one forward-call batch per step, as train.py:59-95 (several attention groups per call are an '
                                      'evaluation feature: set_data_nba with [G,B,N,...] + inference())')
        if self._generic or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            from .training import training_forward        # (generic widths: the training kernels serve forward() with and without autograd)
            return training_forward(self, eps_q, eps_p, eps20, drop_past, drop_future)
        with torch.no_grad():
            return self._forward_values(eps_q, eps_p, eps20)

    def _forward_values(self, eps_q=None, eps_p=None, eps20=None):
        a = self.args
        B = self.batch_size if self._mode == 'nba' else 1
        N = self.agent_num
        self.encode_history()
        self.fu_encoder(eps_q, eps_p)
        self.decoder_future_0(self.qz_sampled, eps20)
        n, Tp, Tf = self.past_feature.shape[0], a.past_length, a.future_length
        st = capi.stream_ptr()
        losses, scratch = self._f(4), self._f(max(n, 1))
        pred1, rec1 = self.pred_traj.contiguous(), self.recover_traj.contiguous()
        fut, past = self.future_traj.contiguous(), self.past_traj.contiguous()
        # calculate_loss_pred / _recover / _kl (model/STTODE.py:372-388), on the loss kernels of csrc/train.hip
        capi.call('sttode_loss_sqerr', pred1, fut, n * 2 * Tf, 1.0 / (B * Tf), losses[0:], None, st)
        capi.call('sttode_loss_sqerr', rec1, past, n * 2 * Tp, 1.0 / (B * Tp), losses[1:], None, st)
        seg = self._mode == 'scenes' and self._S > 1          # batch of scenes: sum of the per-scene objectives (see training.py)
        sp, ags, S = (self._scene_ptr, self._ws['agent_scene'], self._S) if seg else (None, None, 0)
        scratch = self._f(max(n, S, 1))
        capi.call('sttode_loss_kl', self.qz_param, sp, S, n, a.zdim, float(B * N), float(a.min_clip), losses[2:], None, scratch, st)
        self.decoder_future_1(self.pz_sampled)
        capi.call('sttode_loss_diverse', self.diverse_pred_traj.contiguous(), fut, sp, ags, n, 20, 2 * Tf, losses[3:], None, scratch, st)   # :390-395
        lv = losses.tolist()
        return losses.sum(), lv[0], lv[1], lv[2], lv[3]

    @torch.no_grad()
    def inference(self, data=None, z=None):
        """model/STTODE.py:574-623 -> [K, n, Tf, 2] in world coordinates (a permuted view, as in the reference).
        ``z`` ([n*K, zdim], row = agent*K + k) may be injected; otherwise drawn from torch's generator like
        Normal.rsample (model/STTODE.py:89-93,609-616)."""
        self._require_gpu()
        a = self.args
        if a.learn_prior:
            raise NotImplementedError('learn_prior is broken in the reference (pz_layer in_features 256 != 128, model/STTODE.py:361,603)')
        if a.dataset == 'nba' and data is not None and self._mode != 'nba':
            self.set_data_nba(data)
        if self._mode is None:
            raise capi.SttodeError('call set_data / set_data_nba / set_scene_batch before inference()')
        K = a.sample_k
        if self._generic:                                                # widths outside the fused forms: the layer-by-layer form (generic.py)
            if (self.ode_method, self.ode_steps) != ('euler', 1):
                raise NotImplementedError('non-default integrators are built for the reference widths (hidden_dim 64, zdim 32, two blocks)')
            n = self._past.shape[0]
            z = torch.randn(n * K, a.zdim, device=self.device) if z is None else _f32(z, self.device)
            if tuple(z.shape) != (n * K, a.zdim):
                raise ValueError(f'z must be [{n * K}, {a.zdim}], got {tuple(z.shape)}')
            from . import generic
            pred = generic.inference(self, z)
            if self._mode == 'scenes':
                so = self._ws['scene_orig']
                self.scene_orig = so[0] if self._S == 1 else so
            self.diverse_pred = pred
            return pred.permute(1, 0, 2, 3)
        # weights compared AFTER the launch is enqueued (below): only on the one-scene evaluation loop (test.py:171-188), where the host is the
        # critical path up to the launch and no optimizer runs between calls; everywhere else (batches, the NBA path, train / eval alternation)
        # the comparison comes first -- a stale launch there would double the GPU work of every first call after an optimizer step
        late = (self._packed is not None and self._native is not None and self._mode == 'scenes' and self._S == 1 and not self.training)
        nat = self.native(check_weights=not late)
        if nat.timeout_word.value:                                       # an EARLIER launch gave up on its hand-off: its futures were NaN (host load, no sync)
            nat.raise_if_timed_out()
        n = self._past.shape[0]
        if z is None:
            z = torch.randn(n * K, a.zdim, device=self.device)
        elif not (isinstance(z, torch.Tensor) and z.is_cuda and z.dtype == torch.float32 and z.is_contiguous()):
            z = _f32(z, self.device)
        if tuple(z.shape) != (n * K, a.zdim):
            raise ValueError(f'z must be [{n * K}, {a.zdim}], got {tuple(z.shape)}')
        Tp, Tf = a.past_length, a.future_length
        TPX, NOY = packing.tiles_x(Tp), packing.tiles_y(Tf)
        S = self._S if self._mode == 'scenes' else 0
        buf, off = self._workspace(n, S)
        pred = torch.empty(n, K, Tf, 2, dtype=torch.float32, device=self.device)
        st = capi.stream_ptr()
        for attempt in (0, 1):
            if self._mode == 'scenes':
                if capi.TIMING is None:                            # (the generic capi.call: ~3 us of argument conversion on the loop's critical path)
                    if capi.lib().sttode_inference_scenes(nat.h, self._past.data_ptr(), self._scene_ptr.data_ptr(), n, S, z.data_ptr(),
                                                          buf.data_ptr(), pred.data_ptr(), st):
                        raise capi.SttodeError('sttode_inference_scenes failed: ' + capi.lib().sttode_last_error().decode())
                else:
                    capi.call('sttode_inference_scenes', nat.h, self._past, self._scene_ptr, n, S, z, buf, pred, st)
            elif getattr(self, '_G', 1) > 1:
                capi.call('sttode_inference_nba_groups', nat.h, self._past, self._G, self.batch_size, self._N, z, buf, pred, st)
            else:
                capi.call('sttode_inference_nba', nat.h, self._past, self.batch_size, self._N, z, buf, pred, st)
            # the weight-version comparison runs while the GPU works: on the one-scene evaluation loop (test.py:171-188) the host is the
            # critical path up to the launch.  A parameter that changed since the weights were packed (rare: an optimizer step, a
            # load_state_dict) is found here: re-pack (packed() first waits for the launch just made) and launch again.
            if not late or attempt or self._weights_key() == self._packed_key:
                break
            nat = self.native()
            buf, off = self._workspace(n, S)
        if self._mode == 'scenes':
            so = self._view(buf, off, 'scene_orig', S, 2)
            self.scene_orig = so[0] if S == 1 else so
        m = n * K
        self._pf, self._pf_thunk = None, (lambda: self._view(buf, off, 'pf', n, 128))
        self._ws = _LazyViews({'xpad': lambda: self._view(buf, off, 'xpad', n, 16 * TPX), 'enc_in': lambda: self._view(buf, off, 'enc_in', n, Tp, 4),
                               'cur': lambda: self._view(buf, off, 'cur', n, 2), 'orig': lambda: self._view(buf, off, 'orig', n, 2)})
        self._dbg = _LazyViews({'state0': lambda: self._view(buf, off, 'state0', n, 96), 'dbuf': lambda: self._view(buf, off, 'dbuf', m, 16 * TPX),
                                'ybuf': lambda: self._view(buf, off, 'ybuf', m, 16 * NOY), 'state1': lambda: self._view(buf, off, 'state1', m, 96)})
        self.diverse_pred = pred
        return pred.permute(1, 0, 2, 3)

    @torch.no_grad()
    def inference_nba_sharded(self, data_local, z=None, gather=None):
        """NBA path with ONE attention group spanning ranks (SURVEY.md §8e, config 5 read literally): this rank holds
        ``data_local['past_traj'] [B_r, N, Tp, 2]``, scenes ordered by rank.  The only exchange is one all-gather of the raw
        q|k|v rows [B_r*N, 192] (RCCL; ``gather`` may be injected for single-process tests).  Because self-attention uses the
        scores untransposed (rows = keys, columns = queries, hyptransformerlib.py:261-265) every LOCAL key row needs ALL
        queries and values; everything downstream is local.  Returns this rank's [K, B_r*N, Tf, 2]."""
        self._require_gpu()
        a, P = self.args, self.packed()
        self.set_data_nba(data_local)
        ws = self._frontend(vel_from_norm=1)
        n, N, W = self._past.shape[0], self._N, P['past']
        st = capi.stream_ptr()
        g, qkv = self._f(n, 64), self._f(n, 192)
        capi.call('sttode_embed_qkv', W['fc1P'], W['fc1b'], W['posP'], W['peb'], W['fc2P'], W['fc2b'], W['fc3P'], W['fc3b'],
                  W['fc3last'], W['inP'], W['inb'], ws['enc_in'], ws['last'], g, qkv, n, a.past_length, st)
        if gather is None:
            from . import parallel
            gather = parallel.gather_futures  # variable-size row all-gather in rank order
        qkv_all = gather(qkv).contiguous()
        L_all, L_loc = qkv_all.shape[0] // N, n // N
        attn = self._f(n, 64)
        e = qkv.element_size()
        capi.call('sttode_mhgsa_attn', qkv.data_ptr() + 64 * e, qkv_all.data_ptr(), qkv_all.data_ptr() + 128 * e, attn, None, None,
                  L_loc, L_all, N, N * 192, 192, N * 192, 192, N * 192, 192, N * 64, 64, 1.0, 8.0 ** -0.5, st)
        pf = self._f(n, 128)
        capi.call('sttode_post_attn', W['outP'], W['outb'], W['infoP'], W['infob'], W['gateP'], W['gateb'], W['ln1w'], W['ln1b'],
                  W['l1P'], W['l1b'], W['l2P'], W['l2b'], W['ln2w'], W['ln2b'], g, attn, 64, pf, n, self.ODE_TIME, st)
        self.past_feature, self._ws = pf, ws
        K = a.sample_k
        if z is None:
            z = torch.randn(n * K, a.zdim, device=self.device)
        pred = self._decode(pf, z, ws, K)
        self._keep = (g, qkv, qkv_all)
        return pred.permute(1, 0, 2, 3)

    @torch.no_grad()
    def inference_async(self, z=None, metrics_gt=None, metrics_scale=1.0, pred_host=False):
        """Pipelined inference (build-defined): enqueue this batch and return a handle immediately.  ``async_depth`` (default 6, at most 8)
        workspace / prediction slots rotate, so at most that many calls may be in flight: call ``wait(handle)`` (which returns the
        [K, n, Tf, 2] view) before the ``async_depth``-th next call.  Inputs set by set_data / set_scene_batch / set_data_nba must stay
        unmodified until then.
        Batches whose per-trajectory stage takes the chain run in the LAGGED form (include/sttode_hip.h, csrc/role32.hpp): the launch a call
        enqueues carries its per-agent stage and the trajectory groups of the call made three calls earlier (same pipeline stream), so a call's predictions are
        produced when a later call -- or ``wait`` / ``best_of_k_async`` -- enqueues them.  Agrees with inference() to fp32 rounding
        (``native().set_lagged(0)``: the round-3 forms, bitwise inference()).
        ``metrics_gt`` [n, Tf, 2] (contiguous float32 device tensor, e.g. the futures set with the batch): in the lagged form the call's own
        trajectory groups also compute its min-over-K ADE / FDE (utils/metrics.py:7-26) -- ``best_of_k_async(handle)`` then returns them
        without launching a kernel (the values of best_of_k on the same predictions, bit for bit).
        ``pred_host=True`` (lagged form only; raises otherwise): the call's futures are written by the launch STRAIGHT to pinned host
        memory (``handle['pred']`` is then a pinned CPU tensor [n, K, Tf, 2]; ``wait_host(handle)`` makes the host wait for it) -- what
        test.py:186-188 does with a .cpu() per call, without a D2H copy."""
        self._require_gpu()
        a = self.args
        if self._mode is None:
            raise capi.SttodeError('call set_data / set_data_nba / set_scene_batch before inference_async()')
        if self._generic:
            # widths outside the fused forms have no pipelined form: the call runs serially on the caller's stream; the handle keeps the
            # callers of the pipelined API (evaluate.eval_scenes / eval_nba) working unchanged
            if pred_host:
                raise capi.SttodeError('pred_host=True needs the lagged pipelined form (reference widths)')
            out = self.inference(None, z=z)
            return {'generic': True, 'fused_metrics': None, 'slot': -1, 'pred': self.diverse_pred, 'z': None, 'metrics': None,
                    'gt_default': self._future, 'stream': None, 'inputs': (self._past, getattr(self, '_scene_ptr', None))}
        nat = self.native()
        nat.raise_if_timed_out()
        K, n = a.sample_k, self._past.shape[0]
        S = self._S if self._mode == 'scenes' else 0
        # every argument is validated BEFORE a slot is taken or anything native is touched (round-4 advice: a check that failed after the
        # native model had been armed left the request armed for the next call)
        lagged = bool(capi.lib().sttode_async_is_lagged(nat.h, n))
        if z is not None:
            if not (isinstance(z, torch.Tensor) and z.is_cuda and z.dtype == torch.float32 and z.is_contiguous()):
                z = _f32(z, self.device)
            if tuple(z.shape) != (n * K, a.zdim):
                raise ValueError(f'z must be [{n * K}, {a.zdim}], got {tuple(z.shape)}')
        if metrics_gt is not None and not (isinstance(metrics_gt, torch.Tensor) and metrics_gt.is_cuda and metrics_gt.dtype == torch.float32
                                           and metrics_gt.is_contiguous() and tuple(metrics_gt.shape) == (n, a.future_length, 2)):
            raise ValueError(f'metrics_gt must be a contiguous float32 device tensor [{n}, {a.future_length}, 2]')
        if pred_host and not lagged:
            raise capi.SttodeError('pred_host=True needs the lagged pipelined form (a chain-sized batch, reference integrator)')
        slot = self._async_calls % max(2, min(8, int(self.async_depth)))
        key = (n, S, slot)
        if key not in self._async_bufs:
            if len(self._async_bufs) > 16:
                raise capi.SttodeError('too many distinct batch shapes in flight for the async pipeline; call reset_async()')
            _, tot = nat.layout(n, S)
            ws = torch.empty(tot, dtype=torch.float32, device=self.device)
            nat.init_workspace(ws, n, S)
            self._async_bufs[key] = (ws, torch.empty(n, K, a.future_length, 2, dtype=torch.float32, device=self.device),
                                     torch.empty(n * K, a.zdim, dtype=torch.float32, device=self.device))
        self._async_calls += 1
        opts = capi.AsyncOpts()
        if z is None:
            # Latents like Normal.rsample (model/STTODE.py:89-93,609-616).  Lagged form: the call's own launch draws them (Philox4x32-10 on
            # device, csrc/role32.hpp) into the slot's latent buffer, keyed by 64 bits taken from torch's generator here -- reproducible
            # under torch.manual_seed, not the sequence torch.randn would give (device_latents = False: torch.randn on the caller's stream).
            # Every other form: torch.randn.
            if self.device_latents and lagged:
                opts.device_latents, opts.zkey = 1, int(torch.empty((), dtype=torch.int64).random_()) & 0x7fffffffffffffff
                z = self._async_bufs[key][2]
            else:
                z = torch.randn(n * K, a.zdim, device=self.device)
        buf, pred = self._async_bufs[key][:2]
        if pred_host:
            hk = ('host',) + key
            if hk not in self._async_bufs:
                self._async_bufs[hk] = torch.empty(n, K, a.future_length, 2, dtype=torch.float32).pin_memory()
            pred = self._async_bufs[hk]
        st = capi.stream_ptr()
        pstream = self.next_async_stream(n)                     # the pipeline stream this call's launches (and its metrics) run on, or None
        mb = self._async_metrics.get(key)
        if mb is None:                                           # per-slot best-of-K outputs (best_of_k_async): no allocation per call
            t = torch.empty(2, n, dtype=torch.float32, device=self.device)
            mb = self._async_metrics[key] = (t[0], t[1])
        fused = None
        if metrics_gt is not None and lagged:
            opts.metrics_gt, opts.ade, opts.fde, opts.metrics_scale = metrics_gt.data_ptr(), mb[0].data_ptr(), mb[1].data_ptr(), float(metrics_scale)
            fused = (metrics_gt, float(metrics_scale))
        import ctypes
        if self._mode == 'scenes':
            capi.call('sttode_inference_scenes_async', nat.h, self._past, self._scene_ptr, n, S, z, buf, pred, slot, ctypes.addressof(opts), st)
        else:
            opts.nba_groups = getattr(self, '_G', 1)
            capi.call('sttode_inference_nba_async', nat.h, self._past, self.batch_size, self._N, z, buf, pred, slot, ctypes.addressof(opts), st)
        return {'fused_metrics': fused, 'slot': slot, 'pred': pred, 'z': z, 'inputs': (self._past, getattr(self, '_scene_ptr', None)), 'metrics': mb,
                'gt_default': self._future, 'stream': pstream}

    def wait(self, handle):
        """Make the current stream wait for an inference_async() result; returns predictions [K, n, Tf, 2]."""
        if handle.get('generic'):
            return handle['pred'].permute(1, 0, 2, 3)
        nat = self.native()
        nat.raise_if_timed_out()
        capi.call('sttode_wait', nat.h, handle['slot'], capi.stream_ptr())
        return handle['pred'].permute(1, 0, 2, 3)

    def wait_host(self, handle):
        """The HOST waits for an inference_async() result (for ``pred_host=True`` calls: the pinned tensor may be read afterwards);
        returns predictions [K, n, Tf, 2]."""
        nat = self.native()
        capi.call('sttode_wait_host', nat.h, handle['slot'])
        nat.raise_if_timed_out()                                 # (the host has waited: a give-up of that call is visible now)
        return handle['pred'].permute(1, 0, 2, 3)

    def futures_to_host_async(self, handle, out=None, workgroups=8):
        """Enqueue the device -> host copy of an inference_async() call's futures behind its trajectory groups, on the pipeline stream the call
        runs on, and return the pinned tensor [n, K, Tf, 2] (``out``: a pinned float32 tensor of that shape to copy into; default: one
        per slot, reused by the slot's next call).  The copy is made by ``workgroups`` persistent workgroups (csrc/frontend.hip
        sttode_copy_to_host), not by hipMemcpyAsync: beside a full chip the latter costs the pipeline the copy's whole duration
        (62.9 against 76.0 M trajectories/s at 512 scenes), a few workgroups cost 1-2 % (profiles/r04/d2h_copy_kernel_ab.txt).  The host may
        read the tensor after ``wait_host_copy(handle)``; what test.py:186-188 does with a ``.cpu()`` per call."""
        pred = handle['pred']
        if out is None:
            out = self._host_futures.get(handle['slot'])
            if out is None or out.shape != pred.shape:
                out = self._host_futures[handle['slot']] = torch.empty(pred.shape, dtype=torch.float32).pin_memory()
        elif not (out.is_pinned() and out.dtype == torch.float32 and out.is_contiguous() and out.shape == pred.shape):
            raise ValueError('futures_to_host_async: out must be a pinned contiguous float32 tensor of shape %s' % (tuple(pred.shape),))
        nat = self.native()
        capi.call('sttode_async_enqueue', nat.h, handle['slot'])        # (lagged form: the call's groups may still be waiting for a later call)
        st = handle['stream']
        raw = st.cuda_stream if st is not None else capi.stream_ptr()
        if st is None:
            capi.call('sttode_wait', nat.h, handle['slot'], raw)
        if (pred.numel() * 4) % 16 == 0:
            capi.call('sttode_copy_to_host', out, pred, pred.numel() * 4, int(workgroups), raw)
        else:                                                        # (an odd number of floats: the copy kernel moves 16-byte pieces)
            with torch.cuda.stream(st if st is not None else torch.cuda.current_stream(self.device)):
                out.copy_(pred, non_blocking=True)
        ev = handle.get('host_event')
        if ev is None:
            ev = handle['host_event'] = torch.cuda.Event()
        ev.record(st if st is not None else torch.cuda.current_stream(self.device))
        handle['host'] = out
        return out

    def wait_host_copy(self, handle):
        """The HOST waits for the copy enqueued by futures_to_host_async(); returns the pinned tensor as [K, n, Tf, 2]."""
        handle['host_event'].synchronize()
        self.native().raise_if_timed_out()                       # (the copy has drained: a give-up of that call is visible now)
        return handle['host'].permute(1, 0, 2, 3)

    def next_async_stream(self, n):
        if self._generic:
            return None
        return self._next_async_stream(n)

    def _next_async_stream(self, n):
        """torch stream (an ExternalStream over the pipeline's own) the next inference_async() call of ``n`` agents will run on, or None
        when that call will not take the one-stream fused form.  Work enqueued there before the call -- the H2D copy of its inputs, the
        latents -- is ordered in front of it without any cross-stream event:  ``with torch.cuda.stream(s): load(); h = m.inference_async()``.
        Buffers the caller owns per call: the inputs (past, scene_ptr) are read by the call's own launch, the ground truth of fused
        metrics by the launch of the call made <pipeline streams> calls later -- keep one set per slot (``async_depth`` of them) and reuse a
        set only for the call that reuses its slot."""
        import ctypes
        out = ctypes.c_void_p(0)
        capi.call('sttode_async_next_stream', self.native().h, int(n), ctypes.addressof(out))
        if not out.value:
            return None
        st = self._ext_streams.get(out.value)
        if st is None:
            st = self._ext_streams[out.value] = torch.cuda.ExternalStream(out.value, device=self.device)
        return st

    def best_of_k_async(self, handle, gt=None, scale=1.0):
        """Min-over-K ADE / FDE per agent of an inference_async() call, enqueued on the pipeline stream the call runs on (they start the
        moment the call's launch drains; nothing goes onto the caller's stream).  Returns (ade [n], fde [n]): views of the slot's metric
        buffers, valid after ``wait(handle)`` and until the slot's next call.  ``gt`` [n, Tf, 2] must have been written before the
        inference_async() call (default: the futures set with the batch).  Further work on the call's results -- a D2H copy of its
        futures -- may follow on ``handle['stream']`` (stream order: no event; ``wait(handle)`` still covers the metrics only)."""
        if handle.get('generic'):
            return self.best_of_k(handle['pred'], gt=handle.get('gt_default') if gt is None else gt, scale=scale)
        fm = handle.get('fused_metrics')
        if fm is not None and (gt is None or gt.data_ptr() == fm[0].data_ptr()) and float(scale) == fm[1]:
            capi.call('sttode_async_enqueue', self.native().h, handle['slot'])   # the call's groups compute them: make sure they are enqueued
            return handle['metrics']
        gt = handle.get('gt_default') if gt is None else gt
        if not (isinstance(gt, torch.Tensor) and gt.is_cuda and gt.dtype == torch.float32 and gt.is_contiguous()):
            raise ValueError('best_of_k_async needs a contiguous float32 device tensor gt [n, Tf, 2] that was written before the call')
        pred = handle['pred']                                    # contiguous [n, K, Tf, 2]
        n, K, Tf = pred.shape[:3]
        mb = handle['metrics']
        capi.call('sttode_async_best_of_k', self.native().h, handle['slot'], pred, gt, n, K, Tf, float(scale), mb[0], mb[1])
        return mb[0], mb[1]

    @torch.no_grad()
    def horizon_metrics(self, pred_nk, gt=None, scale=1.0):
        """The NBA evaluation's per-horizon metric (test.py:530-551) on device: pred_nk [n,K,Tf,2], gt [n,Tf,2] -> [n,Tf,2] with
        [a, h-1] = (min_k mean_{t<h} |scale (pred - gt)|, min_k |scale (pred_h - gt_h)|)."""
        gt = self._future if gt is None else _f32(gt, self.device)
        pred_nk = pred_nk.contiguous()
        n, K, Tf = pred_nk.shape[:3]
        out = torch.empty(n, Tf, 2, dtype=torch.float32, device=self.device)
        capi.call('sttode_horizon_metrics', pred_nk, gt, n, K, Tf, float(scale), out, capi.stream_ptr())
        return out

    def horizon_metrics_async(self, handle, gt=None, scale=1.0, out=None):
        """horizon_metrics of an inference_async() call, enqueued on the pipeline stream the call runs on (behind its trajectory groups; nothing
        goes onto the caller's stream).  Returns [n,Tf,2], valid after ``wait(handle)``."""
        gt = handle.get('gt_default') if gt is None else gt
        if not (isinstance(gt, torch.Tensor) and gt.is_cuda and gt.dtype == torch.float32 and gt.is_contiguous()):
            raise ValueError('horizon_metrics_async needs a contiguous float32 device tensor gt [n, Tf, 2] that was written before the call')
        pred = handle['pred']
        n, K, Tf = pred.shape[:3]
        if handle.get('generic'):
            return self.horizon_metrics(pred, gt, scale)
        if out is None:
            out = torch.empty(n, Tf, 2, dtype=torch.float32, device=self.device)
        capi.call('sttode_async_horizon_metrics', self.native().h, handle['slot'], pred, gt, n, K, Tf, float(scale), out)
        return out

    def reset_async(self):
        if getattr(self, '_native', None) is not None:
            capi.call('sttode_async_flush', self._native.h)      # outstanding groups of lagged calls read the buffers dropped below
        torch.cuda.synchronize(self.device)
        self._async_bufs = {}
        self._async_metrics = {}

    @torch.no_grad()
    def best_of_k(self, pred_nk, gt=None, scale=1.0):
        """Device-side min-over-K ADE / FDE per agent (utils/metrics.py:7-26). pred_nk [n,K,Tf,2], gt [n,Tf,2]."""
        gt = self._future if gt is None else _f32(gt, self.device)
        pred_nk = pred_nk.contiguous()
        n, K, Tf = pred_nk.shape[:3]
        ade = torch.empty(n, dtype=torch.float32, device=self.device)
        fde = torch.empty_like(ade)
        capi.call('sttode_best_of_k', pred_nk, gt, n, K, Tf, float(scale), ade, fde, capi.stream_ptr())
        return ade, fde
