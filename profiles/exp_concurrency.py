import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from helpers import make_args
from sttode_amd import STTODENet, capi, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
m = STTODENet(make_args(), dev).eval(); m.load_state_dict(to_torch_state_dict(make_weights(1234)))
sb = scenes.make_scene_batch(range(512), 'eth'); n = sb.n_agents; K = 20; mm = n * K
m.set_scene_batch(sb.past, sb.future, sb.scene_ptr); m.inference(None); torch.cuda.synchronize()
P = m.packed(); b0, b1 = P['blk0'], P['blk1']
f = lambda *s: torch.randn(*s, device=dev) * 0.1
A0x, A0y, A1y, z, xpad = f(n, 512), f(n, 512), f(n, 512), f(mm, 32), f(n, 16)
dbuf, ybuf, state1, cur, orig, pred = f(mm, 16), f(mm, 32), f(mm, 96), f(n, 2), f(n, 2), f(mm, 24)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def k0(st): capi.call('sttode_mlp_block0', A0x, A0y, b0['stream'], b0['n_chunks'], z, xpad, dbuf, ybuf, mm, K, 1, 2, st.cuda_stream)
def k1(st): capi.call('sttode_mlp_block1', A1y, b1['stream'], b1['n_chunks'], z, state1, ybuf, cur, orig, pred, mm, K, 12, 2, st.cuda_stream)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e6
print('nonpersistent =', os.environ.get('STTODE_NONPERSISTENT'))
a = timeit(lambda: k0(s1)); b = timeit(lambda: k1(s1))
c = timeit(lambda: (k0(s1), k1(s1)))
d = timeit(lambda: (k0(s1), k1(s2)))
print('mlp0 alone %.0f us, mlp1 alone %.0f us, back-to-back %.0f us, two streams %.0f us' % (a, b, c, d))
