line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(r.get('frac',0),3))"; }
timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 --no-exploratory > /dev/null 2>&1
for i in 1 2 3; do
echo "f32: $(timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 --no-exploratory 2>/dev/null | line)"
echo "bf16x3: $(STTODE_BF16X3=1 timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 --no-exploratory 2>/dev/null | line)"
done
