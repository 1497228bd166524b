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
        self._tables = {}

    def _hip_ok(self, group, ps):
        if group.get('amsgrad') or group.get('maximize') or group.get('capturable') or group.get('differentiable') or not ps:
            return False
        if isinstance(group['lr'], torch.Tensor):
            return False
        dev = ps[0].device
        return dev.type == 'cuda' and all(p.device == dev and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse
                                          and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == dev for p in ps)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(gi, g, [p for p in g['params'] if p.grad is not None]) for gi, g in enumerate(self.param_groups)]
        if not all(self._hip_ok(g, ps) for _, g, ps in groups if ps):
            return super().step()                                 # an option the kernel does not implement anywhere: torch's own step throughout
        for gi, group, ps in groups:
            if not ps:
                continue
            for p in ps:                                          # torch's lazy state initialisation (same keys, same dtypes)
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # gradients as offsets from the lowest gradient address: the engine's per-step flat buffer keeps its layout, so the table is
            # uploaded once and only the base moves
            gp = [p.grad.data_ptr() for p in ps]
            base = min(gp)
            key = (tuple(p.data_ptr() for p in ps), tuple(self.state[p]['exp_avg'].data_ptr() for p in ps), tuple(g - base for g in gp))
            tab = self._tables.get(gi)
            if tab is None or tab[0] != key:
                rows, chunk = [], 0
                for p, g in zip(ps, gp):
                    st = self.state[p]
                    if (g - base) % 4:
                        raise capi.SttodeError('sttode_amd.optim.Adam: gradient not 4-byte aligned')
                    rows.append((p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), (g - base) // 4, p.numel(), chunk))
                    chunk += (p.numel() + 1023) // 1024
                tab = (key, torch.tensor(rows, dtype=torch.int64).to(ps[0].device), chunk, len(rows))
                self._tables[gi] = tab
            t = int(self.state[ps[0]]['step']) + 1
            b1, b2 = group['betas']
            with torch.cuda.device(ps[0].device):
                capi.call('sttode_adam_step', tab[1], tab[3], tab[2], base, float(group['lr']), float(b1), float(b2), float(group['eps']),
                          float(group['weight_decay']), t, capi.stream_ptr())
            for p in ps:
                self.state[p]['step'] += 1
        return loss
