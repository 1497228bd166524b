line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(r.get('frac',0),3), round(r.get('frac_serial_equivalent',0),3), round(d['host_enqueue_ms_per_step'],3))"; }
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms\" for k,v in d['configs'].items()))"; }
timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 > /dev/null 2>&1
for i in 1 2; do
echo "default (lead 160, fe in role, depth 3): $(timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
echo "depth 4: $(timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 --depth 4 2>/dev/null | line)"
echo "fe launch: $(STTODE_FE_IN_ROLE=0 timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
echo "roles first: $(STTODE_ROLE_LEAD=-1 timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
echo "lead 64: $(STTODE_ROLE_LEAD=64 timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
echo "lead 400: $(STTODE_ROLE_LEAD=400 timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
echo "unfused: $(STTODE_FUSED=0 timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
done
echo "legs depth3: $(timeout -k 10 300 python bench.py --no-cpu --no-train --steps 10 2>/dev/null | legs)"
echo "legs depth4: $(timeout -k 10 300 python bench.py --no-cpu --no-train --steps 10 --depth 4 2>/dev/null | legs)"
echo "legs depth3 fe launch: $(STTODE_FE_IN_ROLE=0 timeout -k 10 300 python bench.py --no-cpu --no-train --steps 10 2>/dev/null | legs)"
