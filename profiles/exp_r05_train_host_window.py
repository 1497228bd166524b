"""Where a replayed training step's wall time goes on the HOST (train.py:59-95 loop: set_data -> forward() [returns four Python floats: a
sync] -> zero_grad -> backward -> optimizer.step): per-phase perf_counter means over 200 steps, with the graph's own GPU time from events
around the replay (second loop).  Round 5, first half: forward() waited for the END of the queue and the GPU idled through every other
phase (train_host_window_before.txt); now the loss values leave the graph after its forward half and the host phases overlap the backward half."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes, training
from sttode_amd.optim import Adam
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')


def run(tag, m, set_data):
    opt = Adam(m.parameters(), lr=1e-4)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    orig = training._GraphedStep.run

    def timed_run(self, inputs):
        ev[0].record(); r = orig(self, inputs); ev[1].record()
        return r
    def step(acc=None):
        t0 = time.perf_counter(); set_data(); t1 = time.perf_counter(); tot = m.forward()[0]; t2 = time.perf_counter()
        opt.zero_grad(); t3 = time.perf_counter(); tot.backward(); t4 = time.perf_counter(); opt.step(); t5 = time.perf_counter()
        if acc is not None:
            for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                acc[i] += d
    for _ in range(5): step()
    acc, N = [0.0] * 5, 200
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N):
        step(acc)                                       # free-running: nothing in the loop waits for the end of the queue
    torch.cuda.synchronize(); wall = (time.perf_counter() - t) / N
    training._GraphedStep.run = timed_run               # the replay's own GPU time: a second loop that synchronises after every step
    g = 0.0
    for _ in range(50):
        step(); torch.cuda.synchronize(); g += ev[0].elapsed_time(ev[1])
    training._GraphedStep.run = orig
    names = ('set_data', 'forward (draws, replay, wait for the four loss values)', 'zero_grad', 'backward (hand-over)', 'optimizer.step')
    print(f'{tag}: {wall * 1e3:.3f} ms/step; replay + gradient copy on the GPU {g / 50:.3f} ms; host time per step {sum(acc) / N * 1e3:.3f} ms, of which')
    for nme, v in zip(names, acc):
        print(f'    {nme:58s} {v / N * 1e6:7.1f} us')


Tp, Tf = 5, 10
m = STTODENet(make_args('nba', Tp, Tf), dev); m.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=Tp, future_length=Tf))); m.train()
d = scenes.nba_batch(1, 32)
data = {k: (torch.from_numpy(v) if hasattr(v, 'shape') else v) for k, v in d.items()}
run('NBA-size step (32 x 11)', m, lambda: m.set_data_nba(data))
m2 = STTODENet(make_args('eth', 8, 12), dev); m2.load_state_dict(to_torch_state_dict(make_weights(1234))); m2.train()
obs, pred = scenes.eth_scene(3, n_min=6, n_max=6)
o, p = torch.from_numpy(obs), torch.from_numpy(pred)
run('one-scene step (6 agents)', m2, lambda: m2.set_data(None, o, p))
