"""``sttode_amd.optim.Adam``: torch.optim.Adam (what train.py:122 constructs) with the step of ALL parameters as ONE HIP launch.

Drop-in: same constructor, same ``state`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so ``state_dict()`` /
``load_state_dict()`` round-trip with torch's class and the reference's checkpoints (train.py:188-213) load.  torch's own implementations
walk the model's 88 small tensors -- foreach: ~0.3 ms of host time and a dozen launches per step, fused: three multi_tensor_apply launches
of 41-44 us -- which is 6 % of an NBA-size step and 13 % of a one-scene step of this model (profiles/r05/); the update itself moves 26 MB.
``csrc/train.hip adam_step_kernel`` does it in one launch from a device table of the tensors.  Options the kernel does not implement
(amsgrad, maximize, capturable, differentiable, sparse or non-fp32 / non-contiguous / CPU tensors) take torch's own step.
"""
import torch

from . import capi


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop('fused', None)
        kw.pop('foreach', None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._plans = None        # per group: (group, params, device table, chunks, rows, gradient offsets from the first one) -- None: (re)build
        self._t = {}              # group index -> steps taken (the per-parameter ``step`` tensors of torch's state are written on demand)

    @staticmethod
    def _hip_ok(group, ps):
        if group.get('amsgrad') or group.get('maximize') or group.get('capturable') or group.get('differentiable') or not ps:
            return False
        if isinstance(group['lr'], torch.Tensor):
            return False
        dev = ps[0].device
        return dev.type == 'cuda' and all(p.device == dev and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse
                                          and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == dev for p in ps)

    def _build(self):
        """Validate every group once and upload its tensor table; None if any group needs an option the kernel does not implement."""
        plans = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            if not self._hip_ok(group, ps):
                return None
            steps = set()
            for p in ps:                                          # torch's lazy state initialisation (same keys, same dtypes)
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                steps.add(int(st['step']))
            if len(steps) != 1:
                return None                                       # parameters at different step counts (a hand-made state): torch's own step
            self._t.setdefault(gi, steps.pop())
            g0 = ps[0].grad.data_ptr()
            rows, chunk, offs = [], 0, []
            for p in ps:
                st, g = self.state[p], p.grad.data_ptr()
                if (g - g0) % 4:
                    return None
                offs.append(g - g0)
                rows.append((p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), (g - g0) // 4, p.numel(), chunk))
                chunk += (p.numel() + 1023) // 1024
            plans.append((gi, group, ps, torch.tensor(rows, dtype=torch.int64).to(ps[0].device), chunk, len(rows), offs,
                          [p.data_ptr() for p in ps]))
        return plans

    def _sync_steps(self):
        for gi, group, ps, *_ in (self._plans or ()):
            t = float(self._t.get(gi, 0))
            for p in ps:
                self.state[p]['step'] = torch.tensor(t, dtype=torch.float32)

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self._plans, self._t = None, {}

    def add_param_group(self, group):
        super().add_param_group(group)
        self._plans = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plans = self._plans
        if plans is not None:
            # the hot path: the gradients must sit where the uploaded table expects them relative to the first one (the training engine
            # hands every step's gradients out as views of ONE flat buffer with a fixed layout) and the parameters where they were
            try:
                for gi, group, ps, table, chunks, nrows, offs, pptrs in plans:
                    g0 = ps[0].grad.data_ptr()
                    if [p.grad.data_ptr() - g0 for p in ps] != offs or ps[-1].data_ptr() != pptrs[-1] or ps[0].data_ptr() != pptrs[0]:
                        plans = None
                        break
            except AttributeError:                                # a gradient is None this step
                plans = None
        if plans is None:
            self._sync_steps()
            plans = self._plans = self._build()
            if plans is None:                                     # an option the kernel does not implement: torch's own step throughout
                return super().step()
        L, st = capi.lib(), capi.stream_ptr()
        for gi, group, ps, table, chunks, nrows, offs, pptrs in plans:
            t = self._t[gi] = self._t.get(gi, 0) + 1
            b1, b2 = group['betas']
            if L.sttode_adam_step(table.data_ptr(), nrows, chunks, ps[0].grad.data_ptr(), group['lr'], b1, b2, group['eps'], group['weight_decay'], t, st):
                raise capi.SttodeError('sttode_adam_step failed: ' + L.sttode_last_error().decode())
        return loss

    @property
    def steps_taken(self):
        return dict(self._t)
