#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04ag
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "evaluation_loops or futures_to_host or dataset" > $O/gputests.log 2>&1 || { tail -30 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/eval_scenes_rate.txt
import sys, time, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.evaluate import eval_scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
m = STTODENet(make_args('eth', 8, 12), torch.device('cuda')).eval(); m.load_state_dict(to_torch_state_dict(make_weights(1234)))
class DS:
    def __len__(self): return 512 * 24
    def scene_batch(self, idx): return scenes.make_scene_batch([int(i) % 4096 for i in idx], 'eth')
ds = DS()
for pipe in (True, False, True, False):
    torch.manual_seed(0); torch.cuda.synchronize(); t = time.perf_counter()
    a, f, n = eval_scenes(m, ds, pipelined=pipe)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f'eval_scenes over {len(ds)} scenes ({n} agents), pipelined={pipe}: {dt * 1e3:.0f} ms, ADE {a:.4f} FDE {f:.4f}')
PY
