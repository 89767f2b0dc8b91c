python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bo.json 2>gpurun_out/bo.err; python - <<XEOF
import json
d=json.loads([l for l in open("gpurun_out/bo.json") if l.startswith("{")][0])
print("cfg2", d["roofline"]["kernel_avg_ms"], d["value"])
for k,v in d["config"].get("others",{}).items(): print(k, v.get("kernel"), round(v.get("ms_per_step",0),4), "%.3g frames/s" % v.get("frames_per_s",0), "%.1f GB/s" % v.get("achieved_GBps",0)) if isinstance(v,dict) else print(k,v)
XEOF
