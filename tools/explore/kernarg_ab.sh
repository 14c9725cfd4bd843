#!/bin/bash
for r in 1 2 3; do for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 300 python bench.py --steps 40 --warmup 5 --repeats 5 --no-cpu-baseline --no-host-pipeline --no-real-crops > gpurun_out/ka_${v}_$r.json 2> gpurun_out/ka_${v}_$r.err || { echo fail; tail -3 gpurun_out/ka_${v}_$r.err; exit 1; }
done; done
python - <<'PY'
import json,statistics
for v in (0,1):
    two=[];one=[];k={}
    for r in (1,2,3):
        j=json.loads(open("gpurun_out/ka_%d_%d.json"%(v,r)).read().strip().splitlines()[-1])
        two.append(j["two_lanes"]["ms_per_step"]); one.append(j["single_lane"]["ms_per_step"])
        for n,x in j["kernels_avg_ms"].items(): k.setdefault(n,[]).append(x["avg_ms"])
    print("HIP_FORCE_DEV_KERNARG=%d two lanes %s median %.4f  one lane %s median %.4f"%(v,two,statistics.median(two),one,statistics.median(one)))
    print("   "+"  ".join("%s=%.4f"%(n.replace("conv_","").replace("_kernel",""),statistics.median(x)) for n,x in k.items()))
PY
