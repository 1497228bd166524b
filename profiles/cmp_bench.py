import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open("gpurun_out/r02/bench_%s.json"%f)); print(f, round(d["value"]/1e6,2), round(d["ms_per_step"],3), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), {k:round(v["mean_us"]) for k,v in d["kernels"].items()})
    except Exception as e: print(f, 'ERR', e)
