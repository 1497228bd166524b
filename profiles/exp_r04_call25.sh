#!/bin/bash
# Round 4, call 25: the futures to pinned host memory by a few persistent workgroups (sttode_copy_to_host) against hipMemcpyAsync on the call's stream.
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04y
mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/copy_check.txt
import torch, sys
sys.path.insert(0, '.')
from sttode_amd import capi
src = torch.randn(1 << 22, device='cuda'); dst = torch.empty(1 << 22).pin_memory()
for wgs in (4, 8, 16, 32):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    capi.call('sttode_copy_to_host', dst, src, src.numel() * 4, wgs, capi.stream_ptr()); torch.cuda.synchronize()
    e0.record(); capi.call('sttode_copy_to_host', dst, src, src.numel() * 4, wgs, capi.stream_ptr()); e1.record(); torch.cuda.synchronize()
    print(f'copy_to_host {wgs} workgroups, 16 MB alone: {e0.elapsed_time(e1):.3f} ms = {16.8 / e0.elapsed_time(e1):.1f} GB/s, equal {bool(torch.equal(dst, src.cpu()))}')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
e0.record(); dst.copy_(src, non_blocking=True); e1.record(); torch.cuda.synchronize()
print(f'hipMemcpyAsync 16 MB alone: {e0.elapsed_time(e1):.3f} ms')
PY
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-serial-check --no-sustained --warmup 5 --steps 20"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'incl d2h', round(d['value_incl_d2h']/1e6,2))"; }
for i in 1 2; do
echo "hipMemcpyAsync on the call's stream: $($B 2>/dev/null | line)" | tee -a $O/d2h_kernel_ab.txt
for w in 4 8 16 32; do echo "copy kernel, $w workgroups         : $(STTODE_BENCH_D2H=kernel STTODE_BENCH_D2H_WGS=$w $B 2>/dev/null | line)" | tee -a $O/d2h_kernel_ab.txt; done
done
