"""Collector runs against stream captures (round 5).  A Python process drops models -- native pipeline handles with streams and events, graphs,
pinned buffers -- as cyclic garbage, and the collector destroys them whenever its counters say so, possibly while a stream of the process is
being captured; HIP refuses (or aborts on) some calls in that state.  Forty rounds with the collector's thresholds turned down to (50, 2, 2):
build a model, run inference (native handle) and a few training steps of changing scene size (graph captures); every fourth round capture a
one-scene inference in a user-level hipGraph and DESTROY the round's model -- native handle, captured training graphs, pinned buffers -- by a
collection inside that capture; replay; drop everything else without collecting.  Must end with the line 'ok'."""
import gc, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.optim import Adam
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
gc.set_threshold(50, 2, 2)
sd = to_torch_state_dict(make_weights(1234))
keep = []
for rnd in range(40):
    m = STTODENet(make_args('eth', 8, 12), dev)
    m.load_state_dict(sd)
    m.__dict__['cycle'] = m                                # make sure the model is CYCLIC garbage when dropped
    m.eval()
    o, p = scenes.eth_scene(5000 + rnd, n_min=3 + rnd % 5, n_max=3 + rnd % 5)
    o, p = torch.from_numpy(o), torch.from_numpy(p)
    m.set_data(None, o, p)
    ref = m.inference(None, z=torch.randn(o.shape[0] * 20, 32, device=dev)).clone()      # native handle: streams, events, workspaces
    m.train()
    opt = Adam(m.parameters(), lr=1e-4)
    for it in range(4):                                    # eager -> capture + replay -> replay (twice per size)
        m.set_data(None, o, p)
        tot = m.forward()[0]
        opt.zero_grad(); tot.backward(); opt.step()
    if rnd % 4 == 3:                                       # a user-level capture of the one-scene call with garbage pending and the collector eager
        e = STTODENet(make_args('eth', 8, 12), dev).eval()
        e.load_state_dict(sd)
        e.set_data(None, o, p)
        z = torch.randn(o.shape[0] * 20, 32, device=dev)
        want = e.inference(None, z=z).clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        doomed = [m, opt]                                  # this round's model (native handle, three captured training graphs, pinned buffers): it
        del m, opt, tot                                    # dies INSIDE the capture below, by an explicit collection there
        with torch.cuda.graph(g):
            junk = [[i] for i in range(400)]               # allocations that trip the collector inside the capture as well
            doomed.clear()
            freed = gc.collect()
            out = e.inference(None, z=z)
        assert freed > 0
        m = opt = tot = None
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), rnd
        keep = [g, out, e]
    assert bool(torch.isfinite(ref).all())
    del m, opt, tot
torch.cuda.synchronize()
print('ok: 40 rounds, collector thresholds (50, 2, 2), 10 user-level captures with the round\'s model destroyed inside the capture')
