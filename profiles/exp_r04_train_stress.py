"""Training loop stress: 300 steps (one ETH scene per step, train() mode with rotation and dropout, torch fused Adam) with the grouped /
paired launch forms against the same loop with STTODE_TRAIN_PAIRED=0 (run as two processes with the same seeds): prints a digest of the
losses and of the final parameters; the two digests must agree to ~1e-4 (same math, other summation order in the split weight gradients)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
torch.manual_seed(11); np.random.seed(11)
m = STTODENet(make_args('eth', 8, 12), dev); m.load_state_dict(to_torch_state_dict(make_weights(1234))); m.train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
data = [scenes.eth_scene(200000 + i) for i in range(24)]
data = [(torch.from_numpy(o), torch.from_numpy(p)) for o, p in data]
losses = []
for i in range(300):
    o, p = data[i % len(data)]
    m.set_data(None, o, p, None, None)
    out = m.forward()
    opt.zero_grad(); out[0].backward(); opt.step()
    losses.append(float(out[0].detach()))
torch.cuda.synchronize()
ps = torch.cat([q.detach().flatten().double() for q in m.parameters()])
print(f"paired={os.environ.get('STTODE_TRAIN_PAIRED', '1')}: loss[0] {losses[0]:.6f} loss[-1] {losses[-1]:.6f} mean {np.mean(losses):.6f} | params sum {float(ps.sum()):.8f} norm {float(ps.norm()):.8f} finite {bool(torch.isfinite(ps).all())}")
