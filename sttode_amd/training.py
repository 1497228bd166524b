"""Training step on the HIP kernels: forward-with-tape and backward of ``STTODENet.forward()``.

Reference: model/STTODE.py:553-568 (objective), :372-395 (losses), :214-236 / :276-300 (encoders), :320-347 / :51-77
(decoder), hypertransformer.py:134-153 + hyptransformerlib.py:191-300 (encoder layer / geodesic attention),
ode_demo.py:188,228 (one Euler step of size 12 + relu); what ``train.py:81-87`` drives through ``total_loss.backward()``.

Every network evaluation, loss and gradient runs in ``csrc/train.hip`` kernels (generic MFMA linear / weight-gradient
kernels over the row-major nn.Parameter storage, plus the element-wise pieces).  PyTorch only allocates buffers and, via
one ``autograd.Function``, hands the finished gradients to ``.grad`` so that ``optimizer.step()`` works unchanged.
There is no eager / CPU fallback.
"""
import contextlib
import ctypes
import os

import torch

from . import capi
from .capi import SttodeError

# backward over the decoder columns that carry a gradient (Engine.decoder_live); STTODE_TRAIN_LIVE=0: over all 21 per agent, as rounds 1-4
_LIVE_COLUMNS = os.environ.get('STTODE_TRAIN_LIVE', '1') != '0'
_GATHER_MAX = 32


class _GatherItem(ctypes.Structure):                 # csrc/train.hip GatherItem (include/sttode_hip.h sttode_live_rows_gather)
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('src_plane', ctypes.c_long),
                ('dst_plane', ctypes.c_long), ('row', ctypes.c_int), ('outer', ctypes.c_int)]


EW_MUL, EW_AXPY, EW_GATE_BWD, EW_EULER_FWD, EW_EULER_BWD, EW_RSAMPLE, EW_RELU_BWD, EW_FILL, EW_RSAMPLE_BWD, EW_CUR_ADD = range(10)
EW_TANH_BWD, EW_LATENT_BWD, EW_SUM_CUR, EW_EULER_BWD_CAT, EW_SCALE_ADD, EW_AXPY_ROWS = 10, 11, 12, 13, 14, 15
ACT = {None: 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3}
_ATT = 'ODE_Encoder.odeblock.odefunc.layers.0.'


_AGENT_GRU = os.environ.get('STTODE_TRAIN_AGENT_GRU', '1') != '0'   # the first block's conv + GRU once per agent (0: per trajectory column, A/B)
_PAIRED = os.environ.get('STTODE_TRAIN_PAIRED', '1') != '0'   # decoder_x / decoder_y of a block layer by layer, grouped launches (0: A/B)
# Layer 1 of the decoder MLPs split like the inference chain's (round 5): W1 cat(pf_rep, z, state) = (W1[:, pf] pf + b1) per AGENT -- a table of
# n rows shared by the agent's K samples -- + W1[:, z | state] [z | state] per trajectory: half the layer's products forward and in both
# gradient products (the pf part of dW1 and of dX is formed from the K-summed gradient rows, n of them).  At batch sizes only: below this
# many trajectory columns the step is bound by its number of launches and the split adds four per block (STTODE_TRAIN_L1SPLIT=0: A/B).
# MEASURED NEUTRAL and therefore OFF by default (STTODE_TRAIN_L1SPLIT=1 enables it): 2.301 / 2.308 ms per NBA-size step with, 2.304 / 2.291 ms without
# (profiles/r05/train_l1split_ab.txt) -- halving the layer's products does not shorten its launches: at 7 392 columns they are bound by one-round
# quantisation, tile start and the 15 MB tape store, not by the matrix pipe.
_L1SPLIT_MIN_COLS = int(os.environ.get('STTODE_TRAIN_L1SPLIT_MIN', '2048')) if os.environ.get('STTODE_TRAIN_L1SPLIT', '0') != '0' else 1 << 60
_SCRATCH_BATCH = 32 << 20     # floats (128 MB): split sums of one backward pass at batch sizes (more than 2048 GEMM columns)


def _ld(t):
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0)


class Engine:
    """One training step: ``run_forward`` builds the tape and the loss values, ``run_backward`` returns name -> gradient."""

    def __init__(self, net):
        self.net = net
        self.dev = net.device
        self.scratch = self.main_scratch = torch.empty(4 << 20, dtype=torch.float32, device=self.dev)
        self.red_scratch = None     # split sums of a backward pass's deferred weight-gradient reductions (batch sizes; allocated on first use)
        self.param_grads = True     # False: skip the weight-gradient GEMMs (stage-2 sampler training keeps this net frozen)
        # The step is written as segments (forward_segments / backward_segments); optionally the future trunk's segments run on a side
        # stream with its own split-k scratch (_use_streams: off by default).
        self.side = None            # [stream, stream], created on first use
        self.side_scratch = None
        self.multi = False
        self._hold = []
        self.fused_trunk = os.environ.get('STTODE_TRUNK_FUSED', '1') != '0'   # scene batches: a trunk's forward + tape in one launch
        # dimensions (train.py:37-40 --zdim / --hidden_dim / --num_decompose; 8 heads, ff 1024, conv 32 and GRU 96 are fixed in the reference:
        # model/STTODE.py:23-26,188-189): D model width, HD head width, ZD latent width, PFW = width of past_feature = cat(ftraj_input, ode),
        # ST = column of the GRU state in a decompose block's input cat(pf, z, state), IN = its width
        a = net.args
        self.D, self.ZD, self.NBLK = int(a.hidden_dim), int(a.zdim), int(a.num_decompose)
        self.HD, self.PFW = self.D // 8, 2 * self.D
        self.ST = self.PFW + self.ZD
        self.IN = self.ST + 96
        self.ZS = self.ZD + 96                    # width of a trajectory's own layer-1 input [z | state] (layer-1 split)
        self.split = False                        # set per decoder pass (decoder_fwd) from the pass's trajectory count

    # ---------------------------------------------------------------- streams
    def _use_streams(self, n):
        """The future trunk's segments on a side stream (STTODE_TRAIN_STREAMS=1; default off).  Measured on one-scene steps: in the eager form
        the host enqueues too slowly for two streams to overlap (and holding every temporary alive costs more than it buys: 6.5 vs 4.0 ms),
        in the captured form hipGraphLaunch feeds kernels at the rate one queue executes them (see _GraphedStep._capture)."""
        on = os.environ.get('STTODE_TRAIN_STREAMS', '0') not in ('', '0') and n * 20 <= 32768
        if on and self.side is None:
            self.side = [torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev)]
            self.side_scratch = [torch.empty(4 << 20, dtype=torch.float32, device=self.dev) for _ in self.side]
        return on

    def _enter(self, idx):
        """Bind the launch state to the stream a segment runs on (idx -1: the caller's stream, 0 / 1: side streams)."""
        self.st = capi.stream_ptr()
        self.scratch = self.main_scratch if idx < 0 else self.side_scratch[idx]

    def run_segments(self, segs, launch=None):
        """A step is a list of segments (stream, waits, fn): fn enqueues a chain of kernels on its stream; a side-stream segment starts
        after everything enqueued so far on the caller's stream, a caller's-stream segment after the side streams named in ``waits``.
        ``launch(i)`` replaces fn (replay of the segment's captured hipGraph).  With streams off everything runs in list order."""
        main = torch.cuda.current_stream()
        used = set()
        for i, (stream, waits, fn) in enumerate(segs):
            if not self.multi or stream < 0:
                if self.multi:
                    for w in waits:
                        main.wait_stream(self.side[w])
                        used.discard(w)
                self._enter(-1)
                fn() if launch is None else launch(i)
            else:
                sd = self.side[stream]
                sd.wait_stream(main)
                used.add(stream)
                with torch.cuda.stream(sd):
                    self._enter(stream)
                    fn() if launch is None else launch(i)
        for w in used:                       # the caller's stream ends behind every side stream
            main.wait_stream(self.side[w])
        self._enter(-1)

    def hold(self, t):
        """Keeps a tensor that crosses streams alive until the step's last kernel is enqueued AND the next step begins: the caching
        allocator would otherwise hand its block to a later allocation of the allocating stream while another stream still uses it."""
        if self.multi:
            self._hold.append(t)
        return t

    # ---------------------------------------------------------------- primitives
    def new(self, *shape):
        return self.hold(torch.empty(*shape, dtype=torch.float32, device=self.dev))

    def zeros(self, *shape):
        return self.hold(torch.zeros(*shape, dtype=torch.float32, device=self.dev))

    def lin(self, X, W, b, act=None, xdiv=1, out=None, cols=None):
        """out[c] = act(W X[c / xdiv] + b); X [rows, J] (row stride free), W [I, J] row-major view."""
        cols = X.shape[0] * xdiv if cols is None else cols
        I, J = W.shape
        assert X.shape[1] == J
        out = self.new(cols, I) if out is None else out
        capi.call('sttode_tlinear', X, _ld(X), xdiv, W, _ld(W), 0, b, None, 0, out, _ld(out), cols, J, I, ACT[act], 0, self.st)
        return out

    def lin_tab(self, X, W, tab, tdiv, act=None):
        """out[c] = act(W X[c] + tab[c / tdiv]) (sttode_tlinear_tab: the decoder MLPs' layer 1 with its per-agent part as a table)."""
        cols = X.shape[0]
        I, J = W.shape
        assert X.shape[1] == J and tab.shape[1] == I and tab.shape[0] * tdiv == cols
        out = self.new(cols, I)
        capi.call('sttode_tlinear_tab', X, _ld(X), W, _ld(W), None, tab, _ld(tab), tdiv, out, _ld(out), cols, J, I, ACT[act], self.st)
        return out

    def lin_dx(self, dY, W, mask=None, out=None, accumulate=False, in_features=None):
        """dX = (dY W[:, :in_features]) (* (mask > 0));  W [N, K] is the forward weight."""
        cols, N = dY.shape
        K = W.shape[1] if in_features is None else in_features
        out = self.new(cols, K) if out is None else out
        capi.call('sttode_tlinear', dY, _ld(dY), 1, W, _ld(W), 1, None, mask, _ld(mask) if mask is not None else 0, out, _ld(out),
                  cols, N, K, 0, int(accumulate), self.st)
        return out

    def wgrad(self, dY, X, gW, gb, xdiv=1):
        if not self.param_grads:
            return
        cols, N = dY.shape
        K = X.shape[1]
        assert gW.shape[0] == N and gW.shape[1] == K, (gW.shape, N, K)
        capi.call('sttode_twgrad', dY, _ld(dY), X, _ld(X), xdiv, gW, _ld(gW), gb, cols, N, K, self.scratch, self.scratch.numel(), self.st)

    def lin_bwd(self, dY, W, X, gW, gb, mask=None, out=None, accumulate=False, in_features=None):
        """One layer's backward: returns dX = (dY W[:, :in_features]) (* (mask > 0)) (+ out) and accumulates gW += dY^T X,
        gb += sum dY -- one launch at scene sizes (sttode_tlinear_bwd)."""
        if not self.param_grads:
            return self.lin_dx(dY, W, mask=mask, out=out, accumulate=accumulate, in_features=in_features)
        cols, N = dY.shape
        K = W.shape[1]
        Kdx = K if in_features is None else in_features
        out = self.new(cols, Kdx) if out is None else out
        assert X.shape[1] == K and gW.shape[0] == N and gW.shape[1] == K
        capi.call('sttode_tlinear_bwd', dY, _ld(dY), W, _ld(W), mask, _ld(mask) if mask is not None else 0, out, _ld(out), Kdx,
                  int(accumulate), X, _ld(X), 1, gW, _ld(gW), gb, cols, N, K, self.scratch, self.scratch.numel(), self.st)
        return out

    def ew(self, op, p0, p1=None, p2=None, p3=None, p4=None, i0=0, f0=0.0, count=None):
        capi.call('sttode_train_ewise', op, p0, p1, p2, p3, p4, p0.numel() if count is None else count, i0, float(f0), self.st)

    def grad(self, name):
        """Gradient buffer of a parameter: a view into one flat buffer that is zeroed once per step."""
        self.touched.add(name)
        return self.G[name]

    def _grad_views(self):
        # a FRESH flat buffer per backward (one memset): autograd may adopt the returned views as .grad, so they must not
        # alias the next step's buffer
        tot = sum(((v.numel() + 3) // 4) * 4 for v in self.P.values())           # 16-byte aligned sub-buffers
        self.Gflat = torch.zeros(tot, dtype=torch.float32, device=self.dev)
        self.G, off = {}, 0
        for k, v in self.P.items():
            self.G[k] = self.Gflat[off: off + v.numel()].view(v.shape)
            off += ((v.numel() + 3) // 4) * 4
        self.touched = set()

    # ---------------------------------------------------------------- encoder trunk (PastEncoder / FutureEncoder shared part)
    def trunk_fwd(self, pre, enc_in, last, feat, drop_mask):
        """enc_in [n,T,4] -> feat[:, :64] = ftraj_input, feat[:, 64:128] = ODE encoder output.  Returns the tape."""
        return self.trunk_fwd_multi([(pre, enc_in, last, feat, drop_mask)])[0]

    def trunk_fwd_multi(self, items):
        """Forward of one or several trunks (past and future encoder: the same layers, separate weights, independent of each other).  Scene
        batches with T <= 12: each trunk is ONE launch (csrc/train_trunk.hip), two of them inside a group one launch together.  Otherwise
        (NBA: attention over the batch) layer by layer, layer i of every trunk inside one group -- one launch at scene sizes."""
        P, net = self.P, self.net
        D, HD = self.D, self.HD
        S = []
        for pre, enc_in, last, feat, drop_mask in items:
            n, T = enc_in.shape[0], enc_in.shape[1]
            S.append(dict(t={'n': n, 'T': T, 'pre': pre, 'feat': feat}, pre=pre, a=pre + _ATT, n=n, T=T, X0=enc_in.reshape(n * T, 4), last=last, feat=feat,
                          drop=drop_mask))
        L, Nb = (net.batch_size, net._N) if net._mode == 'nba' else (1, None)
        if all(s['T'] <= 12 for s in S) and self.fused_trunk and D == 64:
            if L == 1:
                with (self.group() if len(S) > 1 else contextlib.nullcontext()):
                    return [self._trunk_fwd_fused(s['t'], s['X0'], s['last'], s['feat'], s['drop']) for s in S]
            # attention over the forward-call batch (the NBA branch): each trunk as TWO launches -- up to the in-projection, then (the attention
            # kernel between) from the attention output on -- instead of twelve layer launches; both trunks of a phase in one launch
            with (self.group() if len(S) > 1 else contextlib.nullcontext()):
                tabs = [self._trunk_fwd_fused(s['t'], s['X0'], s['last'], s['feat'], s['drop'], phase=1) for s in S]
            G = getattr(net, '_G', 1)
            for s, (t, src) in zip(S, tabs):
                qkv, e = src['qkv'], 4
                capi.call('sttode_mhgsa_attn_groups', qkv.data_ptr() + 64 * e, qkv.data_ptr(), qkv.data_ptr() + 128 * e, src['attn'], G, L * Nb * 192,
                          L * Nb * 192, L * Nb * 192, L * Nb * 64, L, L, Nb, Nb * 192, 192, Nb * 192, 192, Nb * 192, 192, Nb * 64, 64, 1.0, 8.0 ** -0.5, 8, self.st)
            with (self.group() if len(S) > 1 else contextlib.nullcontext()):
                for s, (t, src) in zip(S, tabs):
                    self._trunk_launch(src, s['n'], s['T'], s['feat'], 2)
            for t, src in tabs:
                t.update(L=L, Nb=Nb, attn=src['attn'])
            return [t for t, _ in tabs]
        with self.group():
            for s in S:
                s['posin'] = self.new(s['n'] * s['T'], 2 * D)
                self.lin(s['X0'], P[s['pre'] + 'input_fc.weight'], P[s['pre'] + 'input_fc.bias'], out=s['posin'][:, :D])
        for s in S:
            pe = getattr(net, s['pre'][:-1]).pos_encoder.pe
            capi.call('sttode_rows_copy', s['posin'][:, D:], 2 * D, pe, D, s['n'] * s['T'], D, 1, s['T'], self.st)
        with self.group():
            for s in S:
                s['tp'] = self.lin(s['posin'], P[s['pre'] + 'pos_encoder.fc.weight'], P[s['pre'] + 'pos_encoder.fc.bias'])
        with self.group():
            for s in S:
                if s['drop'] is not None:                        # nn.Dropout(0.1) of PositionalAgentEncoding (model/STTODE.py:140,176)
                    self.ew(EW_MUL, s['tp'], s['tp'], s['drop'])
        for s in S:
            s['h3in'] = self.zeros(s['n'], D + 4)
            s['h3in'][:, D + 2] = self.hold(s['last'].to(torch.float32))   # add_category: [0, 0, 1] for the last agent (model/STTODE.py:199-210)
        with self.group():
            for s in S:
                self.lin(s['tp'].view(s['n'], s['T'] * D), P[s['pre'] + 'input_fc2.weight'], P[s['pre'] + 'input_fc2.bias'], out=s['h3in'][:, :D])
        with self.group():
            for s in S:
                s['x'] = s['feat'][:, :D]
                self.lin(s['h3in'][:, :D + 3], P[s['pre'] + 'input_fc3.weight'], P[s['pre'] + 'input_fc3.bias'], out=s['x'])
        with self.group():
            for s in S:
                sa = s['a'] + 'self_attn.temporal_attention_before.'
                s['qkv'] = self.lin(s['x'], P[sa + 'in_proj_weight'], P[sa + 'in_proj_bias'])
        for s in S:
            n, qkv = s['n'], s['qkv']
            if L > 1:
                s['attn'] = self.new(n, D)
                e = qkv.element_size()
                G = getattr(net, '_G', 1)                          # forward-call batches per call (set_data_nba with [G,B,N,...]): attention within each
                gq, go = L * Nb * 3 * D, L * Nb * D
                capi.call('sttode_mhgsa_attn_groups', qkv.data_ptr() + D * e, qkv.data_ptr(), qkv.data_ptr() + 2 * D * e, s['attn'], G, gq, gq, gq, go, L, L,
                          Nb, Nb * 3 * D, 3 * D, Nb * 3 * D, 3 * D, Nb * 3 * D, 3 * D, Nb * D, D, 1.0, float(HD) ** -0.5, HD, self.st)
            else:
                s['attn'] = qkv[:, 2 * D:]                        # softmax over a single key == 1  =>  output == v
        with self.group():
            for s in S:
                sa = s['a'] + 'self_attn.temporal_attention_before.'
                s['ao'] = self.lin(s['attn'], P[sa + 'out_proj.weight'], P[sa + 'out_proj.bias'])
        with self.group():                                       # temporal_info and temporal_gate read the same input: four layers, one launch
            for s in S:
                a = s['a']
                s['tt'] = self.lin(s['ao'], P[a + 'self_attn.temporal_info.weight'], P[a + 'self_attn.temporal_info.bias'], act='tanh')
                s['ss'] = self.lin(s['ao'], P[a + 'self_attn.temporal_gate.weight'], P[a + 'self_attn.temporal_gate.bias'], act='sigmoid')
        with self.group():
            for s in S:
                s['gated'] = self.new(s['n'], D)
                self.ew(EW_MUL, s['gated'], s['tt'], s['ss'])
        for s in S:
            a, n = s['a'], s['n']
            s['xc'] = s['x'].contiguous()
            s['h'], s['xh1'], s['rs1'] = self.new(n, D), self.new(n, D), self.new(n)
            capi.call('sttode_add_ln_fwd', s['xc'], s['gated'], P[a + 'norm1.weight'], P[a + 'norm1.bias'], s['h'], s['xh1'], s['rs1'], n, D, self.st)
        with self.group():
            for s in S:
                s['f1'] = self.lin(s['h'], P[s['a'] + 'linear1.weight'], P[s['a'] + 'linear1.bias'], act='relu')
        with self.group():
            for s in S:
                s['f2'] = self.lin(s['f1'], P[s['a'] + 'linear2.weight'], P[s['a'] + 'linear2.bias'])
        for s in S:
            a, n = s['a'], s['n']
            s['y'], s['xh2'], s['rs2'] = self.new(n, D), self.new(n, D), self.new(n)
            capi.call('sttode_add_ln_fwd', s['h'], s['f2'], P[a + 'norm2.weight'], P[a + 'norm2.bias'], s['y'], s['xh2'], s['rs2'], n, D, self.st)
        with self.group():
            for s in S:
                s['ode'] = self.new(s['n'], D)
                self.ew(EW_EULER_FWD, s['ode'], s['xc'], s['y'], f0=net.ODE_TIME)
        out = []
        for s in S:
            s['feat'][:, D:2 * D] = s['ode']
            t = s['t']
            t.update(X0=s['X0'], posin=s['posin'], tp=s['tp'], drop=s['drop'], h3in=s['h3in'], xc=s['xc'], qkv=s['qkv'], attn=s['attn'], ao=s['ao'],
                     tt=s['tt'], ss=s['ss'], h=s['h'], xh1=s['xh1'], rs1=s['rs1'], f1=s['f1'], xh2=s['xh2'], rs2=s['rs2'], ode=s['ode'], L=L,
                     Nb=Nb if Nb is not None else s['n'])
            out.append(t)
        return out

    def _trunk_launch(self, src, n, T, feat, phase):
        import ctypes
        tbl = (ctypes.c_void_p * len(capi.TRUNK_PTRS))(*[(src[k].data_ptr() if src.get(k) is not None else None) for k in capi.TRUNK_PTRS])
        capi.call('sttode_ttrunk_fwd', tbl, len(capi.TRUNK_PTRS), n, T, feat.stride(0), float(self.net.ODE_TIME), phase, self.st)

    def _trunk_fwd_fused(self, t, X0, last, feat, drop_mask, phase=0):
        """The same forward and the same tape in ONE launch (csrc/train_trunk.hip, attention length 1): the step is bound by the number of
        launches, and a trunk is 21 of them layer by layer.  ``phase=1``: only up to the in-projection (attention over the forward-call batch:
        the caller runs the attention and then phase 2 through _trunk_launch); returns (tape, pointer sources) then."""
        import ctypes
        P, net = self.P, self.net
        pre, n, T = t['pre'], t['n'], t['T']
        a = pre + _ATT
        sa = a + 'self_attn.temporal_attention_before.'
        out = dict(posin=self.new(n * T, 128), tp=self.new(n * T, 64), h3in=self.new(n, 68), xc=self.new(n, 64), qkv=self.new(n, 192),
                   ao=self.new(n, 64), tt=self.new(n, 64), ss=self.new(n, 64), h=self.new(n, 64), xh1=self.new(n, 64), rs1=self.new(n),
                   f1=self.new(n, 1024), xh2=self.new(n, 64), rs2=self.new(n), ode=self.new(n, 64))
        src = dict(fc1_w=P[pre + 'input_fc.weight'], fc1_b=P[pre + 'input_fc.bias'], pos_w=P[pre + 'pos_encoder.fc.weight'],
                   pos_b=P[pre + 'pos_encoder.fc.bias'], fc2_w=P[pre + 'input_fc2.weight'], fc2_b=P[pre + 'input_fc2.bias'],
                   fc3_w=P[pre + 'input_fc3.weight'], fc3_b=P[pre + 'input_fc3.bias'], inproj_w=P[sa + 'in_proj_weight'],
                   inproj_b=P[sa + 'in_proj_bias'], out_w=P[sa + 'out_proj.weight'], out_b=P[sa + 'out_proj.bias'],
                   info_w=P[a + 'self_attn.temporal_info.weight'], info_b=P[a + 'self_attn.temporal_info.bias'],
                   gate_w=P[a + 'self_attn.temporal_gate.weight'], gate_b=P[a + 'self_attn.temporal_gate.bias'],
                   ln1_w=P[a + 'norm1.weight'], ln1_b=P[a + 'norm1.bias'], l1_w=P[a + 'linear1.weight'], l1_b=P[a + 'linear1.bias'],
                   l2_w=P[a + 'linear2.weight'], l2_b=P[a + 'linear2.bias'], ln2_w=P[a + 'norm2.weight'], ln2_b=P[a + 'norm2.bias'],
                   enc_in=X0, last=last, pe=getattr(net, pre[:-1]).pos_encoder.pe, drop=drop_mask, feat=feat, **out)
        for k, v in src.items():
            assert v is None or v.is_contiguous() or k == 'feat', k
        if phase == 1:
            src['attn'] = self.new(n, 64)
        self._trunk_launch(src, n, T, feat, phase)
        t.update(X0=X0, drop=drop_mask, attn=out['qkv'][:, 128:], L=1, Nb=n, **out)
        return (t, src) if phase == 1 else t

    @contextlib.contextmanager
    def group(self):
        """Independent layers issued inside leave as ONE launch (sttode_tgemm_group: up to four per launch, scene sizes and batch sizes)."""
        capi.call('sttode_tgemm_group', 1)
        try:
            yield
        except BaseException:
            capi.call('sttode_tgemm_group', -1)
            raise
        capi.call('sttode_tgemm_group', 0)

    def trunk_bwd(self, t, dfeat):
        """dfeat [n,128] (row stride free) = grad wrt cat(ftraj_input, ode_out).  Accumulates parameter grads."""
        self.trunk_bwd_multi([(t, dfeat)])

    def trunk_bwd_multi(self, items):
        """Backward of one or several trunks (the past and the future encoder: the same layers, separate weights, independent of each
        other) walked through TOGETHER: layer i of every trunk is issued inside one group, i.e. one launch -- a one-scene step is bound by
        the number of launches, and the two trunks are 2 x 10 linear-layer backward launches otherwise."""
        P, g, net = self.P, self.grad, self.net
        D, HD = self.D, self.HD
        S = []
        for t, dfeat in items:
            pre = t['pre']
            assert dfeat.stride(1) == 1 and dfeat.shape[1] >= 2 * D and dfeat.stride(0) < 65536
            S.append(dict(t=t, pre=pre, a=pre + _ATT, op=pre + _ATT + 'self_attn.temporal_attention_before.', n=t['n'], T=t['T'], dfeat=dfeat))
        with self.group():
            for s in S:                                          # dx = dfeat[:, :64] + d, dy = T d, d = dfeat[:, 64:128] * (ode > 0): one piece,
                n = s['n']                                       # reading the strided rows of dfeat (no .contiguous() copies)
                s['dx'], s['dy'] = self.new(n, D), self.new(n, D)
                self.ew(EW_EULER_BWD_CAT, s['dfeat'], s['t']['ode'], None, s['dx'], s['dy'], i0=s['dfeat'].stride(0) | ((D << 16) if D != 64 else 0),
                        f0=net.ODE_TIME, count=n * D)
        for s in S:
            t, a, n = s['t'], s['a'], s['n']
            s['dsum2'] = self.new(n, D)
            capi.call('sttode_ln_bwd', s['dy'], t['xh2'], t['rs2'], P[a + 'norm2.weight'], s['dsum2'], g(a + 'norm2.weight'), g(a + 'norm2.bias'), n, D,
                      self.scratch, self.scratch.numel(), self.st)
        with self.group():
            for s in S:
                t, a = s['t'], s['a']
                s['df1'] = self.lin_bwd(s['dsum2'], P[a + 'linear2.weight'], t['f1'], g(a + 'linear2.weight'), g(a + 'linear2.bias'), mask=t['f1'])
        with self.group():
            for s in S:                                          # dh = dsum2 (residual branch of LN2(h + f)) + W1^T df1
                t, a = s['t'], s['a']
                self.lin_bwd(s['df1'], P[a + 'linear1.weight'], t['h'], g(a + 'linear1.weight'), g(a + 'linear1.bias'), out=s['dsum2'], accumulate=True)
        for s in S:
            t, a, n = s['t'], s['a'], s['n']
            s['dsum1'] = self.new(n, D)
            capi.call('sttode_ln_bwd', s['dsum2'], t['xh1'], t['rs1'], P[a + 'norm1.weight'], s['dsum1'], g(a + 'norm1.weight'), g(a + 'norm1.bias'), n, D,
                      self.scratch, self.scratch.numel(), self.st)
        with self.group():
            for s in S:
                self.ew(EW_AXPY, s['dx'], s['dsum1'], f0=1.0)    # residual branch of LN1(x + gated)
                s['du'], s['dv'] = self.new(s['n'], D), self.new(s['n'], D)
                self.ew(EW_GATE_BWD, s['dsum1'], s['t']['tt'], s['t']['ss'], s['du'], s['dv'])
        with self.group():
            for s in S:
                t, a = s['t'], s['a']
                s['dao'] = self.lin_bwd(s['du'], P[a + 'self_attn.temporal_info.weight'], t['ao'], g(a + 'self_attn.temporal_info.weight'),
                                        g(a + 'self_attn.temporal_info.bias'))
        with self.group():
            for s in S:
                t, a = s['t'], s['a']
                self.lin_bwd(s['dv'], P[a + 'self_attn.temporal_gate.weight'], t['ao'], g(a + 'self_attn.temporal_gate.weight'),
                             g(a + 'self_attn.temporal_gate.bias'), out=s['dao'], accumulate=True)
        with self.group():
            for s in S:
                t, op = s['t'], s['op']
                s['dattn'] = self.lin_bwd(s['dao'], P[op + 'out_proj.weight'], t['attn'], g(op + 'out_proj.weight'), g(op + 'out_proj.bias'))
        for s in S:
            t = s['t']
            if t['L'] > 1:
                s['dqkv'] = self.new(s['n'], 3 * D)
                capi.call('sttode_mhgsa_attn_bwd', t['qkv'], s['dattn'], s['dqkv'], t['L'], t['Nb'], HD, self.st)
        with self.group():
            for s in S:
                t, op = s['t'], s['op']
                W, gW, gb = P[op + 'in_proj_weight'], g(op + 'in_proj_weight'), g(op + 'in_proj_bias')
                if t['L'] > 1:
                    self.lin_bwd(s['dqkv'], W, t['xc'], gW, gb, out=s['dx'], accumulate=True)
                else:
                    # attention length 1 (scene batches): the softmax over one key is 1, the output is v -- dv = dattn, and the gradients of
                    # q and k are exactly zero (their rows of the in-projection's gradient stay the zeros of the flat buffer)
                    self.lin_bwd(s['dattn'], W[2 * D:], t['xc'], gW[2 * D:], gb[2 * D:], out=s['dx'], accumulate=True)
        # dx is now the gradient wrt ftraj_input
        with self.group():
            for s in S:
                t, pre = s['t'], s['pre']
                s['dh2'] = self.lin_bwd(s['dx'], P[pre + 'input_fc3.weight'], t['h3in'][:, :D + 3], g(pre + 'input_fc3.weight'), g(pre + 'input_fc3.bias'),
                                        in_features=D)
        with self.group():
            for s in S:
                t, pre, n, T = s['t'], s['pre'], s['n'], s['T']
                s['dtp'] = self.lin_bwd(s['dh2'], P[pre + 'input_fc2.weight'], t['tp'].view(n, T * D), g(pre + 'input_fc2.weight'),
                                        g(pre + 'input_fc2.bias')).view(n * T, D)
        with self.group():
            for s in S:
                if s['t']['drop'] is not None:
                    self.ew(EW_MUL, s['dtp'], s['dtp'], s['t']['drop'])
        with self.group():
            for s in S:
                t, pre = s['t'], s['pre']
                s['dtf'] = self.lin_bwd(s['dtp'], P[pre + 'pos_encoder.fc.weight'], t['posin'], g(pre + 'pos_encoder.fc.weight'),
                                        g(pre + 'pos_encoder.fc.bias'), in_features=D)
        with self.group():
            for s in S:
                self.wgrad(s['dtf'], s['t']['X0'], g(s['pre'] + 'input_fc.weight'), g(s['pre'] + 'input_fc.bias'))

    # ---------------------------------------------------------------- decoder (Decoder.forward, model/STTODE.py:320-347)
    def l1_fwd(self, pre, inp, tab, K):
        """Layer 1 of a decoder MLP: plain, or -- layer-1 split -- from the agent's table and the trajectory's own [z | state] columns."""
        P = self.P
        if tab is None:
            return self.lin(inp, P[pre + 'layers.0.weight'], P[pre + 'layers.0.bias'], act='relu')
        return self.lin_tab(inp[:, self.PFW:], P[pre + 'layers.0.weight'][:, self.PFW:], tab, K, act='relu')

    def l1_tables(self, pres, pf):
        """The per-agent tables W1[:, pf] pf + b1 of the given MLPs (one grouped launch)."""
        P = self.P
        with self.group():
            return [self.lin(pf, P[pre + 'layers.0.weight'][:, :self.PFW], P[pre + 'layers.0.bias']) for pre in pres]

    def l1_bwd(self, pre, da, inp, din, accumulate, pf, dpf, K):
        """Backward of layer 1: plain (din [m, IN]), or -- layer-1 split -- the trajectory part into din [m, ZS] and the agent part from the
        K-summed gradient rows into dpf [n, 2 D] (accumulated)."""
        P, g = self.P, self.grad
        W, gW, gb = P[pre + 'layers.0.weight'], g(pre + 'layers.0.weight'), g(pre + 'layers.0.bias')
        if pf is None:
            self.lin_bwd(da, W, inp, gW, gb, out=din, accumulate=accumulate)
            return
        PFW = self.PFW
        self.lin_bwd(da, W[:, PFW:], inp[:, PFW:], gW[:, PFW:], gb, out=din, accumulate=accumulate)   # (gb: the whole bias gradient = sum over all columns)
        n = pf.shape[0]
        dA = self.new(n, da.shape[1])
        capi.call('sttode_rows_reduce', dA, _ld(dA), da, _ld(da), n, da.shape[1], K, 0, self.st)
        self.lin_bwd(dA, W[:, :PFW], pf, gW[:, :PFW], None, out=dpf, accumulate=True)

    def mlp_fwd(self, pre, inp, tab=None, K=1):
        P = self.P
        a1 = self.l1_fwd(pre, inp, tab, K)
        a2 = self.lin(a1, P[pre + 'layers.1.weight'], P[pre + 'layers.1.bias'], act='relu')
        out = self.lin(a2, P[pre + 'layers.2.weight'], P[pre + 'layers.2.bias'])
        return out, (a1, a2)

    def mlp_fwd_pair(self, pre_a, pre_b, inp, tabs=(None, None), K=1):
        """decoder_y and decoder_x of a block (same input, separate weights, model/STTODE.py:71-77) layer by layer: at batch sizes the two
        products of a layer leave as one launch (sttode_tgemm_group)."""
        P = self.P
        acts = []
        xa = xb = inp
        for li, act in ((0, 'relu'), (1, 'relu'), (2, None)):
            with self.group():
                if li == 0:
                    xa, xb = self.l1_fwd(pre_a, inp, tabs[0], K), self.l1_fwd(pre_b, inp, tabs[1], K)
                else:
                    xa = self.lin(xa, P[f'{pre_a}layers.{li}.weight'], P[f'{pre_a}layers.{li}.bias'], act=act)
                    xb = self.lin(xb, P[f'{pre_b}layers.{li}.weight'], P[f'{pre_b}layers.{li}.bias'], act=act)
            acts.append((xa, xb))
        return (acts[2][0], (acts[0][0], acts[1][0])), (acts[2][1], (acts[0][1], acts[1][1]))

    def mlp_bwd_pair(self, pre_a, pre_b, inp, saved_a, saved_b, dout_a, dout_b, din, pf=None, dpf=None, K=1):
        """Backward of the pair: layers 2 and 1 of both MLPs side by side (four products per launch at batch sizes); their layer-0 input
        gradients add into the same ``din``, so those two stay launches of their own, in order."""
        P, g = self.P, self.grad
        da, db_ = dout_a, dout_b
        for li, (sa, sb) in ((2, (saved_a[1], saved_b[1])), (1, (saved_a[0], saved_b[0]))):
            capi.call('sttode_tgemm_group', 1)
            try:
                da = self.lin_bwd(da, P[f'{pre_a}layers.{li}.weight'], sa, g(f'{pre_a}layers.{li}.weight'), g(f'{pre_a}layers.{li}.bias'), mask=sa)
                db_ = self.lin_bwd(db_, P[f'{pre_b}layers.{li}.weight'], sb, g(f'{pre_b}layers.{li}.weight'), g(f'{pre_b}layers.{li}.bias'), mask=sb)
            except BaseException:
                capi.call('sttode_tgemm_group', -1)
                raise
            capi.call('sttode_tgemm_group', 0)
        self.l1_bwd(pre_a, da, inp, din, False, pf, dpf, K)
        self.l1_bwd(pre_b, db_, inp, din, True, pf, dpf, K)

    def mlp_bwd(self, pre, inp, saved, dout, din, accumulate, pf=None, dpf=None, K=1):
        P, g = self.P, self.grad
        a1, a2 = saved
        da2 = self.lin_bwd(dout, P[pre + 'layers.2.weight'], a2, g(pre + 'layers.2.weight'), g(pre + 'layers.2.bias'), mask=a2)
        da1 = self.lin_bwd(da2, P[pre + 'layers.1.weight'], a1, g(pre + 'layers.1.weight'), g(pre + 'layers.1.bias'), mask=a1)
        self.l1_bwd(pre, da1, inp, din, accumulate, pf, dpf, K)

    def block_fwd(self, i, past, K, xhat_prev, pf, z, want_x, inp=None):
        P = self.P
        pre = f'decoder.decompose.{i}.'
        n, Tp = past.shape[0], past.shape[1]
        m = n * K
        # The first block reads x_true - 0: its conv + GRU see the same track for every sample of an agent (model/STTODE.py:329-335, x_hat = 0).
        # Run them ONCE per agent and hand the state to the agent's K columns (the inference path has done so since round 1) -- K = 21 in
        # forward(): a 21 x smaller input projection and GRU sequence; the backward sums the K columns' state gradients first (block_bwd).
        agent = xhat_prev is None and K > 1 and _AGENT_GRU
        mg, Kg = (n, 1) if agent else (m, K)
        x, e = self.new(mg, Tp, 2), self.new(mg * Tp, 32)
        capi.call('sttode_conv_fwd', past, Kg, xhat_prev, P[pre + 'conv_past.weight'], P[pre + 'conv_past.bias'], x, e, mg, Tp, self.st)
        gi = self.lin(e, P[pre + 'encoder_past.weight_ih_l0'], P[pre + 'encoder_past.bias_ih_l0'])       # [mg*Tp, 288], row c*Tp + t
        H = self.new(Tp + 1, mg, 96)                                                                     # H[0] = 0 (written by the launch), H[t+1] = h_t
        tapes = self.new(Tp, mg, 384)
        prefix = inp is None                                                                             # cat(pf_rep, z, state): the prefix may come filled (decoder_fwd)
        IN, ST, PFW, ZD = self.IN, self.ST, self.PFW, self.ZD
        if prefix:
            inp = self.new(m, IN)
        if agent:
            state = self.new(n, 96)
            capi.call('sttode_gru_seq_fwd', gi, P[pre + 'encoder_past.weight_hh_l0'], P[pre + 'encoder_past.bias_hh_l0'], H, tapes,
                      state, 96, n, Tp, self.st)
            capi.call('sttode_rows_copy', inp[:, ST:], IN, state, 96, m, 96, K, n, self.st)              # row c of inp <- state of agent c / K
        else:
            capi.call('sttode_gru_seq_fwd', gi, P[pre + 'encoder_past.weight_hh_l0'], P[pre + 'encoder_past.bias_hh_l0'], H, tapes,
                      inp[:, ST:], IN, m, Tp, self.st)                                                   # all Tp steps, one launch
        if prefix:
            if not self.split:                                                                           # (layer-1 split: the pf columns of inp are never read)
                capi.call('sttode_rows_copy', inp, IN, pf, _ld(pf), m, PFW, K, n, self.st)
            capi.call('sttode_rows_copy', inp[:, PFW:], IN, z, _ld(z), m, ZD, 1, m, self.st)
        tabs = (None, None)
        if self.split:
            tabs = self.l1_tables([pre + 'decoder_y.'] + ([pre + 'decoder_x.'] if want_x else []), pf) + [None]
        if want_x and _PAIRED:
            (yh, sy), (xh, sx) = self.mlp_fwd_pair(pre + 'decoder_y.', pre + 'decoder_x.', inp, tabs, K)
        else:
            yh, sy = self.mlp_fwd(pre + 'decoder_y.', inp, tabs[0], K)
            xh, sx = self.mlp_fwd(pre + 'decoder_x.', inp, tabs[1], K) if want_x else (None, None)
        return dict(pre=pre, m=m, Tp=Tp, K=K, x=x, e=e, H=H, tapes=tapes, inp=inp, yh=yh, sy=sy, xh=xh, sx=sx, pf=pf if self.split else None, agent=agent)

    def block_bwd(self, b, dyh, dxh, need_dx, dpf=None):
        """Returns (din [m, IN] -- layer-1 split: [m, ZS], the trajectory's own columns [z | state]; the pf part went into dpf -- , dx [m,Tp,2] | None)."""
        P, g = self.P, self.grad
        pre, m, Tp = b['pre'], b['m'], b['Tp']
        pf, K = b.get('pf'), b['K']
        split = pf is not None
        LD, SO = (self.ZS, self.ZD) if split else (self.IN, self.ST)          # row length of din, column of the GRU state in it
        din = self.new(m, LD)
        if dxh is not None and _PAIRED:
            self.mlp_bwd_pair(pre + 'decoder_y.', pre + 'decoder_x.', b['inp'], b['sy'], b['sx'], dyh, dxh, din, pf, dpf, K)
        else:
            self.mlp_bwd(pre + 'decoder_y.', b['inp'], b['sy'], dyh, din, False, pf, dpf, K)
            if dxh is not None:
                self.mlp_bwd(pre + 'decoder_x.', b['inp'], b['sx'], dxh, din, True, pf, dpf, K)
        dstate, lds, mg = din[:, SO:], LD, m
        if b.get('agent'):                                          # conv + GRU ran once per agent: its state gradient is the sum over the agent's columns
            assert not need_dx
            mg = m // K
            dstate, lds = self.new(mg, 96), 96
            capi.call('sttode_rows_reduce', dstate, 96, din[:, SO:], LD, mg, 96, K, 0, self.st)
        dgi = self.new(mg * Tp, 288)
        dgh = self.new(Tp, mg, 288)
        capi.call('sttode_gru_seq_bwd', dstate, lds, b['tapes'], b['H'], P[pre + 'encoder_past.weight_hh_l0'], dgi, dgh, mg, Tp, self.st)
        self.wgrad(dgh.view(Tp * mg, 288), b['H'][:Tp].view(Tp * mg, 96), g(pre + 'encoder_past.weight_hh_l0'), g(pre + 'encoder_past.bias_hh_l0'))
        de = self.lin_bwd(dgi, P[pre + 'encoder_past.weight_ih_l0'], b['e'], g(pre + 'encoder_past.weight_ih_l0'),
                          g(pre + 'encoder_past.bias_ih_l0'), mask=b['e'])
        dx = self.new(mg, Tp, 2) if need_dx else None
        capi.call('sttode_conv_bwd', de, b['x'], P[pre + 'conv_past.weight'], dx, g(pre + 'conv_past.weight'), g(pre + 'conv_past.bias'),
                  mg, Tp, self.scratch, self.scratch.numel(), self.st)
        return din, dx

    def decoder_fwd(self, pf, z, K, past, cur, want_recover, qz_eps=None):
        """Decoder.forward (model/STTODE.py:320-347) over ``num_decompose`` blocks: block i reads x_true - x_hat_{i-1} (x_hat_{-1} = 0) and the
        same cat(pf, z); prediction = sum y_hat_i + cur_location, reconstruction = sum x_hat_i (the last block's decoder_x is needed only for it).
        ``qz_eps`` = (qz [n, zd], eps [n (K - 1), zd]) instead of ``z`` [n K, zd]: the blocks' input prefix cat(pf, z) is written for the
        first TWO blocks by one launch (sttode_decoder_inputs; further blocks copy it) and z is never assembled."""
        n, Tp = past.shape[0], past.shape[1]
        Tf = self.net.args.future_length
        m = n * K
        nb = self.NBLK
        self.split = m >= _L1SPLIT_MIN_COLS
        inps = [None] * nb
        if qz_eps is not None:
            inps = [self.new(m, self.IN) for _ in range(nb)]
            if self.split:        # only z is written, at its usual columns (the pf prefix is replaced by the per-agent tables)
                capi.call('sttode_decoder_inputs', inps[0][:, self.PFW:], inps[1][:, self.PFW:] if nb > 1 else None, self.IN, pf, _ld(pf), qz_eps[0], qz_eps[1],
                          n, K, 0, self.ZD, self.st)
            else:
                capi.call('sttode_decoder_inputs', inps[0], inps[1] if nb > 1 else None, self.IN, pf, _ld(pf), qz_eps[0], qz_eps[1], n, K, self.PFW, self.ZD, self.st)
            for i in range(2, nb):                                  # (further blocks: the same prefix)
                o, w = (self.PFW, self.ZD) if self.split else (0, self.ST)
                capi.call('sttode_rows_copy', inps[i][:, o:], self.IN, inps[0][:, o:], self.IN, m, w, 1, m, self.st)
        blocks, xprev = [], None
        for i in range(nb):
            b = self.block_fwd(i, past, K, xprev, pf, z, want_recover or i + 1 < nb, inp=inps[i])
            blocks.append(b)
            xprev = b['xh']
        pred = self.new(m, 2 * Tf)
        rec = None
        ysum, xsum = blocks[0]['yh'], blocks[0]['xh']
        for b in blocks[1:-1]:                                      # (num_decompose > 2: partial sums of the middle blocks)
            t = self.new(m, 2 * Tf)
            self.ew(EW_SUM_CUR, t, ysum, b['yh'], None, i0=2 * Tf, f0=K)
            ysum = t
            if want_recover:
                t = self.new(m, 2 * Tp)
                self.ew(EW_SUM_CUR, t, xsum, b['xh'], None, i0=2 * Tp, f0=K)
                xsum = t
        with self.group():                                          # two independent pieces: one launch
            if nb > 1:
                self.ew(EW_SUM_CUR, pred, ysum, blocks[-1]['yh'], cur, i0=2 * Tf, f0=K)
            else:
                self.ew(EW_SUM_CUR, pred, ysum, self.zeros(m, 2 * Tf), cur, i0=2 * Tf, f0=K)
            if want_recover:
                rec = self.new(m, 2 * Tp)
                self.ew(EW_SUM_CUR, rec, xsum, blocks[-1]['xh'] if nb > 1 else self.zeros(m, 2 * Tp), None, i0=2 * Tp, f0=K)
        return dict(blocks=blocks, b0=blocks[0], b1=blocks[-1], n=n, K=K, m=m, pred=pred, rec=rec, split=self.split)

    def decoder_live(self, d, best):
        """The decoder's tape reduced to the columns that carry a gradient.  forward() decodes 1 + 20 samples per agent, but the objective
        (model/STTODE.py:553-568) touches sample 0 (mse + recover terms) and, through the min over K of loss_diverse (:390-395), ONE of
        the prior samples per agent; the decoder treats trajectory columns independently (conv, GRU and MLPs per column,
        model/STTODE.py:50-77), so the backward pass of the other 19 columns is exactly zero in every layer and contributes nothing to
        any parameter gradient.  The reference's autograd multiplies through those zeros; here the tape rows of the two live columns per
        agent are gathered (ONE launch, csrc/train.hip live_rows_gather_kernel) and the backward pass runs over 2 n columns instead of
        21 n -- the same gradient (the reference's own backward() digests: tests/test_gpu_parity.py::test_training_step_vs_reference_*)."""
        n, K1, m2 = d['n'], d['K'], 2 * d['n']
        items, blocks = [], []

        def take(src, row, outer=1):
            m = n * K1
            dst = self.new(outer * m2 * row)
            assert src.is_contiguous() and src.numel() == outer * m * row
            items.append(_GatherItem(src.data_ptr(), dst.data_ptr(), m * row, m2 * row, row, outer))
            return dst
        for b in d['blocks']:
            Tp = b['Tp']
            nb = dict(pre=b['pre'], m=m2, Tp=Tp, K=2, pf=b['pf'], agent=b.get('agent', False))
            if nb['agent']:                                          # conv + GRU ran per agent: their tape has no column dimension
                nb.update(x=b['x'], e=b['e'], H=b['H'], tapes=b['tapes'])
            else:
                nb['x'] = take(b['x'], 2 * Tp).view(m2, Tp, 2)
                nb['e'] = take(b['e'], Tp * 32).view(m2 * Tp, 32)
                nb['H'] = take(b['H'], 96, Tp + 1).view(Tp + 1, m2, 96)
                nb['tapes'] = take(b['tapes'], 384, Tp).view(Tp, m2, 384)
            nb['inp'] = take(b['inp'], self.IN).view(m2, self.IN)
            for key in ('sy', 'sx'):
                nb[key] = tuple(take(t, t.shape[1]).view(m2, t.shape[1]) for t in b[key]) if b[key] is not None else None
            blocks.append(nb)
        for i in range(0, len(items), _GATHER_MAX):
            chunk = items[i:i + _GATHER_MAX]
            table = (_GatherItem * len(chunk))(*chunk)              # (a named object: it must outlive the call that reads it)
            capi.call('sttode_live_rows_gather', table, len(chunk), best, n, K1, self.st)
        return dict(blocks=blocks, b0=blocks[0], b1=blocks[-1], n=n, K=2, m=m2, split=d.get('split', False))

    def decoder_bwd(self, d, dpred, drec, dpf, dz, dpf_accumulate=True):
        """Accumulates dpf [n, 2 D] (+=; ``dpf_accumulate=False``: writes it); writes dz [m, zd] if not None.  Returns the gradient of the
        blocks' summed layer-1 input [m, IN] = cat(d pf_rep | d z | d state) (its columns 2 D .. 2 D + zd - 1 are dz)."""
        n, K, m = d['n'], d['K'], d['m']
        blocks = d['blocks']
        split = d.get('split', False)
        if split and not dpf_accumulate:
            self.ew(EW_FILL, dpf, f0=0.0)                          # (layer-1 split: every MLP adds its agent part into dpf)
        # x_{i+1} = x_true - x_hat_i  =>  d x_hat_i = (d recover) - d x_{i+1}; the last block's x_hat only feeds the reconstruction
        dxh, din_sum = drec, None
        for i in range(len(blocks) - 1, -1, -1):
            din, dx = self.block_bwd(blocks[i], dpred, dxh, i > 0, dpf)
            if din_sum is None:
                din_sum = din
            else:
                self.ew(EW_AXPY, din, din_sum, f0=1.0)              # (the running sum ends in block 0's buffer, as in rounds 3-4)
                din_sum = din
            if i > 0:
                dxh = dx.view(m, -1)
                self.ew(EW_SCALE_ADD, dxh, drec, f0=-1.0)           # d x_hat_{i-1} = -dx_i (+ drec)
        if split:                                                  # din_sum [m, ZS] = cat(d z | d state); dpf is complete
            if dz is not None:
                dz.copy_(din_sum[:, :self.ZD])
            return din_sum
        capi.call('sttode_rows_reduce', dpf, _ld(dpf), din_sum, self.IN, n, self.PFW, K, int(dpf_accumulate), self.st)
        if dz is not None:
            dz.copy_(din_sum[:, self.PFW:self.ST])
        return din_sum

    # ---------------------------------------------------------------- the objective (model/STTODE.py:553-568)
    def forward_segments(self, eps_q, eps20, drop_past=None, drop_future=None, streams=None):
        """The forward pass as segments (see run_segments): future trunk on side 1 beside the past trunk; then the q-net and ONE decoder
        pass over 1 + 20 samples per agent -- sample 0 decoded from the posterior draw (pred_traj / recover_traj, model/STTODE.py:
        553-560), samples 1..20 from the prior draws (diverse_pred_traj, :562-566): the two passes of the reference share every
        weight, so one pass over 21 columns per agent halves the launches of the decoder's forward and backward."""
        net, a = self.net, self.net.args
        self.P = {k: v for k, v in net.named_parameters()}
        P = self.P
        self._hold = []
        self.multi = self._use_streams(net._past.shape[0]) if streams is None else bool(streams)
        for t_in in (eps_q, eps20, drop_past, drop_future):       # the caller may drop them while a side stream still reads them
            if t_in is not None:
                self.hold(t_in)
        B = net.batch_size if net._mode == 'nba' else 1
        N = net.agent_num
        n, Tp, Tf, zd = net._past.shape[0], a.past_length, a.future_length, a.zdim
        mode = 0 if net._mode == 'scenes' else 1
        K1 = 21
        if n * K1 * Tp > int(os.environ.get('STTODE_TGEMM_MIN_COLS_BWD', '600')) and self.red_scratch is None:         # batch sizes: room for a backward pass's deferred split sums
            self.red_scratch = torch.empty(_SCRATCH_BATCH, dtype=torch.float32, device=self.dev)
        V = self.V = {}

        PFW = self.PFW

        def f_front():
            V['ws'] = ws = net._frontend(vel_from_norm=0)
            V['past'] = ws['xpad'][:, :2 * Tp].reshape(n, Tp, 2).contiguous()
            V['fut'] = (net._future - ws['orig'][:, None, :]).contiguous()
            V['hcat'] = self.new(n, 2 * PFW)
            V['lastpos'] = self.hold(net._past[:, -1].contiguous())

        def f_future():
            ws, hcat = V['ws'], V['hcat']
            enc_f = self.new(n, Tf, 4)
            capi.call('sttode_frontend_future', net._future, V['lastpos'], n, Tf, mode, net._N or 1, ws.get('scene_orig'),
                      ws.get('agent_scene'), net._scene_ptr if mode == 0 else None, enc_f, self.st)
            if _PAIRED and not self.multi:
                V['enc_f'] = enc_f                                   # one stream: the two trunks' forward as ONE launch (f_past)
            else:
                V['tf'] = self.trunk_fwd('future_encoder.', enc_f, ws['last'], hcat[:, PFW:], drop_future)

        def f_past():
            ws, hcat = V['ws'], V['hcat']
            if _PAIRED and not self.multi:                           # one stream: both trunks together (grouped launches)
                V['tf'], V['tp'] = self.trunk_fwd_multi([('future_encoder.', V['enc_f'], ws['last'], hcat[:, PFW:], drop_future),
                                                         ('past_encoder.', ws['enc_in'], ws['last'], hcat[:, :PFW], drop_past)])
            else:
                V['tp'] = self.trunk_fwd('past_encoder.', ws['enc_in'], ws['last'], hcat[:, :PFW], drop_past)

        def f_dec():
            ws, hcat, fut, past = V['ws'], V['hcat'], V['fut'], V['past']
            hq = self.lin(hcat, P['future_encoder.out_mlp.affine_layers.0.weight'], P['future_encoder.out_mlp.affine_layers.0.bias'], act='relu')
            qzp = self.lin(hq, P['future_encoder.qz_layer.weight'], P['future_encoder.qz_layer.bias'])
            qz = self.new(n, zd)
            self.ew(EW_RSAMPLE, qz, qzp, eps_q, i0=zd)
            # z per (agent, sample): sample 0 = the posterior draw, samples 1..20 = the prior draws -- read where they are by the launch that
            # fills both blocks' input prefix
            assert zd % 4 == 0 and eps20.is_contiguous() and qz.is_contiguous()
            d = self.decoder_fwd(hcat[:, :PFW], None, K1, past, ws['cur'], True, qz_eps=(qz, eps20))
            losses = self.new(5)                                    # the four terms and their sum
            live = _LIVE_COLUMNS
            KG = 2 if live else K1                                  # columns per agent that carry a gradient (see decoder_live)
            dpred, drec, dqzp = self.new(n * KG, 2 * Tf), self.new(n * KG, 2 * Tp), self.new(n, 2 * zd)
            best = self.hold(torch.empty(n, dtype=torch.int32, device=self.dev)) if live else None
            # several independent scenes in one step (set_scene_batch): the objective is the SUM of the per-scene objectives, i.e. the
            # gradient equals what S reference steps would accumulate (per-scene KL clamp and per-scene mean of the best-of-20 term)
            seg = net._mode == 'scenes' and net._S > 1
            sp, ags, S = (net._scene_ptr, ws['agent_scene'], net._S) if seg else (None, None, 0)
            if live:
                capi.call('sttode_loss_objective_live', d['pred'], d['rec'], fut, past, qzp, sp, ags, S, n, K1, 2 * Tf, 2 * Tp, zd, 1.0 / (B * Tf),
                          1.0 / (B * Tp), float(B * N), float(a.min_clip), losses, dpred, drec, dqzp, best, self.scratch, self.scratch.numel(), self.st)
            else:
                capi.call('sttode_loss_objective', d['pred'], d['rec'], fut, past, qzp, sp, ags, S, n, K1, 2 * Tf, 2 * Tp, zd, 1.0 / (B * Tf),
                          1.0 / (B * Tp), float(B * N), float(a.min_clip), losses, dpred, drec, dqzp, self.scratch, self.scratch.numel(), self.st)
            if getattr(self, 'publish', None) is not None:         # a step being captured: the values reach the host from HERE (see _GraphedStep)
                capi.call('sttode_publish_values', losses, 5, *self.publish, self.st)
            self.step_id = getattr(self, 'step_id', 0) + 1
            self.tape = dict(step_id=self.step_id, tp=V['tp'], tf=V['tf'], hcat=hcat, hq=hq, qzp=qzp, eps_q=eps_q, d=d, dpred=dpred,
                             drec=drec, dqzp=dqzp, n=n, zd=zd, K1=K1, best=best)
            V['losses'] = losses
            # attributes the reference sets (read by callers)
            pr = d['pred'].view(n, K1, Tf, 2)
            net.past_feature = hcat[:, :PFW]
            net.qz_param = qzp
            net.qz_sampled = qz
            net.pred_traj = pr[:, 0]
            net.recover_traj = d['rec'].view(n, K1, Tp, 2)[:, 0]
            net.diverse_pred_traj = pr[:, 1:]
            net.past_traj, net.future_traj, net.cur_location = past, fut, past[:, -1:]

        return [(-1, (), f_front), (1, (), f_future), (-1, (), f_past), (-1, (1,), f_dec)]

    def run_forward(self, eps_q, eps20, drop_past=None, drop_future=None):
        self.run_segments(self.forward_segments(eps_q, eps20, drop_past, drop_future))
        return self.V['losses']

    def backward_segments(self):
        """The backward pass as segments: decoder pass + q-net on the caller's stream, then the future trunk on side 1 beside the past
        trunk (disjoint parameters)."""
        T = self.tape
        # The tape (and the flat gradient buffer it fills) belongs to the MOST RECENT eager forward().  backward() of an older loss, or
        # a second backward() of the same loss, would silently differentiate the wrong step: refuse instead.
        if T is None:
            raise SttodeError('training backward: the tape of this forward() was already consumed (backward() called twice, or '
                              'retain_graph reuse); run forward() again')
        n, zd, K1 = T['n'], T['zd'], T['K1']
        P, g = self.P, self.grad
        W = {}

        def b_dec():
            self._grad_views()
            if self.red_scratch is not None:                            # batch sizes: the split weight gradients' reductions as one launch per 16
                capi.call('sttode_twgrad_defer', 1, self.red_scratch, self.red_scratch.numel())
            W['dpf'] = dpf = self.new(n, self.PFW)
            d, KG = T['d'], K1
            if T.get('best') is not None:                               # backward over the two columns per agent that carry a gradient
                d, KG = self.decoder_live(T['d'], T['best']), 2
            din = self.decoder_bwd(d, T['dpred'], T['drec'], dpf, None, dpf_accumulate=False)
            dqz = self.new(n, zd)                                       # gradient of the posterior draw = sample 0 of every agent: dz of row a KG
            if d.get('split'):
                capi.call('sttode_rows_copy', dqz, zd, din, KG * self.ZS, n, zd, 1, n, self.st)
            else:
                capi.call('sttode_rows_copy', dqz, zd, din[:, self.PFW:], KG * self.IN, n, zd, 1, n, self.st)
            dqzp = T['dqzp']                                            # starts as the KL gradient
            self.ew(EW_RSAMPLE_BWD, dqz, T['qzp'], T['eps_q'], dqzp, i0=zd)
            dhq = self.lin_bwd(dqzp, P['future_encoder.qz_layer.weight'], T['hq'], g('future_encoder.qz_layer.weight'),
                               g('future_encoder.qz_layer.bias'), mask=T['hq'])
            W['dhcat'] = self.hold(self.lin_bwd(dhq, P['future_encoder.out_mlp.affine_layers.0.weight'], T['hcat'],
                                                g('future_encoder.out_mlp.affine_layers.0.weight'),
                                                g('future_encoder.out_mlp.affine_layers.0.bias')))

        def b_future():
            if not (_PAIRED and not self.multi):
                self.trunk_bwd(T['tf'], W['dhcat'][:, self.PFW:])

        def b_past():
            dpf = W['dpf']
            dh = W['dhcat']
            assert dh.stride(1) == 1 and dh.stride(0) < 65536
            self.ew(EW_AXPY_ROWS, dpf, dh, i0=(dh.stride(0) << 16) | self.PFW, f0=1.0, count=n * self.PFW)   # dpf += dhcat[:, :2 D] (read where it is)
            if _PAIRED and not self.multi:                              # one stream: both trunks layer by layer, grouped launches
                self.trunk_bwd_multi([(T['tf'], W['dhcat'][:, self.PFW:]), (T['tp'], dpf)])
            else:
                self.trunk_bwd(T['tp'], dpf)
            capi.call('sttode_twgrad_defer', 0, None, 0)                # the pending reductions run here, behind the last gradient
            self.tape = None

        return [(-1, (), b_dec), (1, (), b_future), (-1, (), b_past)]

    def run_backward(self, gout=None, step_id=None):
        T = self.tape
        if T is not None and step_id is not None and T.get('step_id') != step_id:
            raise SttodeError('training backward: this loss belongs to an earlier forward(); only the most recent forward() of a model '
                              'can be differentiated (its tape was overwritten by the newer forward())')
        try:
            self.run_segments(self.backward_segments())
        except BaseException:
            capi.call('sttode_twgrad_defer', -1, None, 0)
            raise
        if gout is not None:
            self.Gflat.mul_(gout)
        return {k: self.G[k] for k in self.touched}


class _LossFn(torch.autograd.Function):
    """Hands the HIP-computed parameter gradients to autograd: ``total_loss.backward()`` fills ``.grad`` (train.py:83-87).
    The only autograd input is a private scalar anchor (one graph edge instead of one per parameter: the step is host-bound); the
    gradients are written to the parameters' ``.grad`` directly, so ``torch.autograd.grad(total, params)`` is not supported."""

    @staticmethod
    def forward(ctx, total, engine, names, ready, params, anchor):
        ctx.engine, ctx.names, ctx.ready, ctx.params = engine, names, ready, params
        ctx.step_id = engine.tape['step_id'] if (ready is None and engine.tape is not None) else None
        return total.clone()

    @staticmethod
    def backward(ctx, gout):
        if ctx.ready is not None:                                  # graph replay already produced the gradients
            flat, views, state = ctx.ready
            _consume(state)
            flat.mul_(gout)
            G = views
        else:
            G = ctx.engine.run_backward(gout, ctx.step_id)
        _hand_over(G, ctx.names, ctx.params, rotating=ctx.ready is not None)
        return None, None, None, None, None, None


def _consume(state):
    """A graph-replay step hands its gradients over ONCE, through whichever path comes first (the direct one of _LossTensor.backward or
    the autograd node): a second backward() raises like the eager step's consumed tape does, instead of silently doubling ``.grad``."""
    if state['consumed']:
        raise SttodeError('training backward: the tape of this forward() was already consumed (backward() called twice on a '
                          'graph-replayed step); call forward() again')
    state['consumed'] = True


def _hand_over(G, names, params, rotating=False):
    """The gradients of a step become the parameters' ``.grad``: autograd's AccumulateGrad would copy each of the 88 views of the flat
    buffer into a fresh tensor (88 extra kernels per step).  The flat buffer is private to this step, so the views can BE the .grad
    tensors; an existing .grad (gradient accumulation) is added to -- in place when the step's buffer is its own (eager steps), OUT of
    place when it is one of a replayed step's two rotating buffers (the existing .grad may itself be a view of one of them, which a later
    replay overwrites)."""
    for nm, p in zip(names, params):
        g = G.get(nm)
        if g is None or not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = g
        elif rotating:
            p.grad = p.grad + g
        else:
            p.grad.add_(g)


class _LossTensor(torch.Tensor):
    """``total_loss`` as forward() returns it.  ``total_loss.backward()`` (train.py:84) with no arguments hands the gradients over directly
    -- no autograd-engine round trip (thread hand-off, graph traversal: ~0.1 ms of a 1.5 ms step); any other use (a scaled loss, explicit
    ``gradient=``, ``inputs=``, retain_graph) goes through the autograd node the tensor also carries (_LossFn)."""

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        fast = getattr(self, '_sttode_step', None)
        if fast is None or gradient is not None or inputs is not None or retain_graph or create_graph:
            return super().backward(gradient, retain_graph, create_graph, inputs)
        self._sttode_step = None
        engine, names, ready, params, step_id = fast
        if ready is not None:
            _consume(ready[2])
            G = ready[1]                                           # graph replay already produced the gradients
        else:
            G = engine.run_backward(None, step_id)
        _hand_over(G, names, params, rotating=ready is not None)


_PARKED_GRAPHS = []


class _GraphedStep:
    """One hipGraph per step shape: the ~165 launches of forward-with-tape + backward replayed as a single graph launch
    (the training step is launch-latency-bound at the reference's scene sizes).  Inputs are copied into static buffers,
    the loss values and the flat gradient buffer are static outputs."""

    def __init__(self, eng, net, inputs, drawn=None):
        self.eng, self.net = eng, net
        self.static = {k: (v.clone() if v is not None else None) for k, v in inputs.items()}
        # drawn: name -> (rows, cols, kind) of the random inputs the GRAPH draws itself (torch's generator is graph-safe: a captured
        # normal_() / bernoulli_() reads seed and offset at replay, and replay() advances the offset by the graph's total) -- in the order
        # and with the calls of the eager step, so a seeded run draws the same numbers either way
        self.drawn = drawn or {}
        for k, (rows, cols, kind) in self.drawn.items():
            if kind != 'unused':
                self.static[k] = torch.empty(rows, cols, device=eng.dev)
        self.graph = None

    def __del__(self):
        # torch's CUDAGraph destructor synchronises the device on ROCm; while a stream of the process is being captured that call is refused
        # and the failed check inside a destructor aborts the process (profiles/exp_r05_capture_gc_stress.py: a model collected inside a
        # user's own torch.cuda.graph block).  A graph that dies during a capture is parked instead and released by the next replay.
        try:
            g = getattr(self, 'graph_obj', None)
            if g is not None and torch.cuda.is_current_stream_capturing():
                _PARKED_GRAPHS.append(g)
        except Exception:                                           # (interpreter shutdown: the module's globals may be gone)
            pass

    def _bind(self):
        net, st = self.net, self.static
        net._past, net._future = st['past'], st['future']
        if st['scene_ptr'] is not None:
            net._scene_ptr = st['scene_ptr']

    def _capture(self):
        """The whole step (forward with tape + backward) as ONE hipGraph on one stream.  Measured alternatives: the future trunk's
        segments as branches of the same graph, or every segment as its own graph replayed on two streams -- hipGraphLaunch feeds a
        graph's kernels at about the rate a queue executes small dependent kernels (~5 us each), so neither ran the trunks
        concurrently, and every graph boundary cost 20-60 us of idle queue (7 graphs: ~0.15 ms per step)."""
        eng, st = self.eng, self.static
        segs = eng.forward_segments(st['eps_q'], st['eps20'], st['drop_past'], st['drop_future'], streams=False)
        torch.cuda.synchronize()
        # The loss values leave for the host in the MIDDLE of the graph (after the forward half): forward() returns them as Python floats
        # (model/STTODE.py:568), and reading them off the end of the queue (`.tolist()`) left the GPU idle for the whole host side of
        # train.py:61-67 -- zero_grad, backward's hand-over, optimizer.step, the next set_data and draws: 0.21 ms of a 0.81 ms one-scene step
        # (profiles/r05/train_host_window.txt).  One launch publishes them to pinned memory with a sequence count; run() spins on that word.
        self.host_vals = torch.zeros(8, dtype=torch.float32).pin_memory()
        self.host_seq = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.dev_seq = torch.zeros(2, dtype=torch.int32, device=eng.dev)
        self.host_np, self.replays = self.host_vals.numpy(), 0
        self.graph_obj = torch.cuda.CUDAGraph()
        eng.publish = (self.host_vals, self.dev_seq, self.host_seq)
        # No cyclic-collector run inside the capture: the body allocates thousands of Python objects, and a collection it trips may destroy
        # whatever garbage the process holds -- other models' native handles (streams, events), pinned buffers, older graphs -- whose
        # destructors make HIP calls that are not permitted while a stream captures (the runtime aborts the process: seen once a
        # collection happened to fall into _grad_views under capture).  torch.cuda.graph collects on entry; from there to the end: off.
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph_obj):
                for k, (rows, cols, kind) in self.drawn.items():
                    if kind == 'unused':
                        torch.randn(rows, cols, device=eng.dev)     # pz_distribution.rsample(): drawn, never used (model/STTODE.py:525)
                    elif kind == 'bern':
                        st[k].bernoulli_(_DROP_KEEP).div_(_DROP_KEEP)
                    else:
                        st[k].normal_()
                eng.run_segments(segs)
                try:
                    eng.run_segments(eng.backward_segments())
                except BaseException:
                    capi.call('sttode_twgrad_defer', -1, None, 0)
                    raise
        finally:
            eng.publish = None
            if gc_was_on:
                gc.enable()
        self.keep = (eng._hold, eng.V, eng.G, eng.main_scratch, eng.red_scratch)        # static buffers of the graph
        eng._hold = []
        self.out = (eng.V['losses'], eng.Gflat, {k: eng.G[k] for k in eng.touched})

    def run(self, inputs):
        one_scene = self.graph is not None and self.net._mode == 'scenes' and self.net._S == 1
        for k, v in inputs.items():
            if k == 'scene_ptr' and one_scene:                      # [0, n] for ONE scene of n agents (n is part of the graph's key): copied when captured
                continue
            if v is not None and v is not self.static[k]:           # (random inputs nobody passed in are drawn by the graph itself)
                self.static[k].copy_(v)
        self._bind()
        if _PARKED_GRAPHS:                                          # graphs that died during somebody's capture (see __del__): nothing captures now
            _PARKED_GRAPHS.clear()
        if self.graph is None:
            self._capture()
            self.graph = True
            self.attrs = {k: getattr(self.net, k) for k in ('past_feature', 'qz_param', 'qz_sampled', 'pred_traj', 'recover_traj',
                                                            'diverse_pred_traj', 'past_traj', 'future_traj', 'cur_location')}
        self.graph_obj.replay()
        for k, v in self.attrs.items():
            setattr(self.net, k, v)
        losses, flat, G = self.out
        # .grad must not alias the graph's static output: the step's gradients are copied (ONE D2D copy) into one of TWO persistent flat
        # buffers used in turn, whose per-parameter views are built once -- a fresh clone per step needed 88 slice + view operations, ~0.15 ms
        # of host time on the host-bound one-scene step.  A step's gradients therefore stay valid until the step after the next one runs;
        # gradients that are ACCUMULATED across steps never live in these buffers (_hand_over adds out of place).
        if getattr(self, 'rot', None) is None:
            self.rot, self.rot_i = [], 0
            for _ in range(2):
                buf = torch.empty_like(flat)
                views, off = {}, 0
                for k, v in self.eng.P.items():
                    if k in G:
                        views[k] = buf[off: off + v.numel()].view(v.shape)
                    off += ((v.numel() + 3) // 4) * 4
                self.rot.append((buf, views))
        buf, views = self.rot[self.rot_i]
        self.rot_i ^= 1
        buf.copy_(flat)
        total = losses[4].clone()
        self.replays = (self.replays + 1) & 0xFFFFFFFF
        if capi.lib().sttode_wait_value(self.host_seq.data_ptr(), self.replays, 60.0):
            msg = capi.lib().sttode_last_error().decode()
            torch.cuda.synchronize()
            self.replays = int(self.dev_seq[0].item()) & 0xFFFFFFFF     # whatever did run: the next replay is counted from the device's own count
            raise capi.SttodeError('the replayed step did not publish its loss values: ' + msg)
        return (total, self.host_np[:4].tolist()), (buf, views, {'consumed': False, 'rotating': True})


def _names_params(eng, net):
    """(names, parameters, pointer token) of ``net``, cached on the engine and VALIDATED on every call: each cached Parameter must still
    be the object registered under its name in its module, and each module the one registered in its parent (~130 dict look-ups; walking
    named_parameters() every step costs ~0.1 ms of a 1.3 ms step).  A replaced Parameter or sub-module rebuilds the cache, and the token
    -- a hash of every parameter's storage pointer, part of the hipGraph key -- changes with it, so a captured graph is never replayed
    against parameters it does not hold and ``.grad`` lands on the live objects."""
    cache = getattr(eng, '_names_params', None)
    if cache is not None:
        names, params, token, pslots, mslots = cache
        if all(m._parameters.get(k) is p for m, k, p in pslots) and all(pm._modules.get(k) is m for pm, k, m in mslots):
            return names, params, token
    names, params, pslots, mslots = [], [], [], []
    for mname, mod in net.named_modules():
        if mname:
            parent, _, leaf = mname.rpartition('.')
            mslots.append((net.get_submodule(parent) if parent else net, leaf, mod))
    for name, p in net.named_parameters():
        mname, _, leaf = name.rpartition('.')
        names.append(name)
        params.append(p)
        pslots.append((net.get_submodule(mname) if mname else net, leaf, p))
    token = hash(tuple(p.data_ptr() for p in params))
    eng._names_params = (names, params, token, pslots, mslots)
    return names, params, token


_GRAPH_MAX_AGENTS = int(os.environ.get('STTODE_TRAIN_GRAPH_MAX', '512'))
_DROP_KEEP = 0.9

def training_forward(net, eps_q=None, eps_p=None, eps20=None, drop_past=None, drop_future=None):
    """STTODENet.forward() with autograd support (see module docstring).  Returns the reference's 5-tuple.
    ``net.train_graphs`` (default True): after one eager step per shape the whole step is captured into a hipGraph."""
    a, dev = net.args, net.device
    n = net._past.shape[0]
    eng = getattr(net, '_engine', None)
    if eng is None or eng.dev != dev:
        eng = net._engine = Engine(net)
        net._graphs, net._graph_seen = {}, set()
    names, params, ptr_token = _names_params(eng, net)
    ready = None
    keep = _DROP_KEEP                                               # nn.Dropout(0.1) after the positional fc, both encoders (train mode)
    # random inputs the caller did not pass in, in the order the eager step draws them: (name, rows, cols, kind)
    want = [('eps_q', n, a.zdim, 'normal' if eps_q is None else None), ('eps_p', n, a.zdim, 'unused' if eps_p is None else None),
            ('eps20', n * 20, a.zdim, 'normal' if eps20 is None else None),
            ('drop_past', n * a.past_length, a.hidden_dim, 'bern' if (drop_past is None and net.training) else None),
            ('drop_future', n * a.future_length, a.hidden_dim, 'bern' if (drop_future is None and net.training) else None)]
    drawn = {nm: (r, c, kind) for nm, r, c, kind in want if kind is not None}
    key = (net._mode, n, net._S if net._mode == 'scenes' else net.batch_size, drop_past is not None or net.training,
           drop_future is not None or net.training, tuple(drawn), _LIVE_COLUMNS, _AGENT_GRU,
           ptr_token, params[0].data_ptr(), params[-1].data_ptr(),     # graphs hold raw parameter pointers ...
           float(a.min_clip), float(net.ODE_TIME))            # ... and bake scalar kernel arguments in
    # a step is ~170 launches of 5-30 us each and the host needs ~15 us to enqueue one: replay wins as long as the launches are short
    # (one scene: launch-bound; an NBA batch of 32 x 11 agents: 3.5 ms eager for 2.5 ms of kernels)
    graphs = getattr(net, 'train_graphs', os.environ.get('STTODE_TRAIN_GRAPHS', '1') != '0') and net._future is not None and n <= _GRAPH_MAX_AGENTS
    replay = graphs and (key in net._graphs or key in net._graph_seen)

    def given(t):
        return None if t is None else t.to(dev, torch.float32).contiguous()
    if replay:                                                      # the graph draws what is missing (see _GraphedStep)
        eps_q, eps20, drop_past, drop_future = given(eps_q), given(eps20), given(drop_past), given(drop_future)
        inputs = dict(past=net._past, future=net._future, scene_ptr=net._scene_ptr if net._mode == 'scenes' else None,
                      eps_q=eps_q, eps20=eps20, drop_past=drop_past, drop_future=drop_future)
        if key not in net._graphs:
            if len(net._graphs) >= 48:
                net._graphs.clear()
            net._graphs[key] = _GraphedStep(eng, net, inputs, drawn)
        losses, ready = net._graphs[key].run(inputs)
    else:
        def draw(t, name):
            if t is not None or name not in drawn:
                return given(t)
            rows, cols, kind = drawn[name]
            out = torch.empty(rows, cols, device=dev)
            return out.bernoulli_(keep).div_(keep) if kind == 'bern' else out.normal_()
        eps_q = draw(eps_q, 'eps_q')
        if 'eps_p' in drawn:
            torch.randn(n, a.zdim, device=dev)                      # pz_distribution.rsample(): drawn, never used (model/STTODE.py:525)
        eps20 = draw(eps20, 'eps20')
        drop_past, drop_future = draw(drop_past, 'drop_past'), draw(drop_future, 'drop_future')
        if graphs:
            net._graph_seen.add(key)                                # first time: eager (also warms one-time kernel attributes)
    if ready is None:
        losses = eng.run_forward(eps_q, eps20, drop_past, drop_future)
        tot_dev, lv = losses[4], None
    else:
        tot_dev, lv = losses                                        # a replayed step: the four values are on the host already
    if getattr(eng, 'anchor', None) is None:
        eng.anchor = torch.zeros((), device=dev, requires_grad=True)
    total = _LossFn.apply(tot_dev, eng, names, ready, params, eng.anchor).as_subclass(_LossTensor)
    total._sttode_step = (eng, names, ready, params, eng.tape['step_id'] if (ready is None and eng.tape is not None) else None)
    if lv is None:
        lv = losses.tolist()
    return total, lv[0], lv[1], lv[2], lv[3]
