timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 > /dev/null 2>&1
for S in 512 1024 256; do for i in 1 2 3; do
  for CFG in "0 1 0 0" "2 0 1 1" "2 0 1 0" "0 1 0 1"; do
    set -- $CFG
    echo "scenes=$S chain_wgs=$1 agents_fused=$2 gru0_stream=$3 a_priority=$4: $(STTODE_CHAIN_WGS=$1 STTODE_AGENTS_FUSED=$2 STTODE_GRU0_STREAM=$3 STTODE_A_PRIORITY=$4 timeout -k 10 200 python bench.py --legs none --no-cpu --scenes $S --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))")"
  done
done; done
