#!/bin/bash
# On the GPU box: bench the shipped library and each named variant (variants_so/libglomseg_<name>.so), one JSON per run
# under gpurun_out/.  tools/ab.sh tag name1 name2 ...
tag=$1; shift
mkdir -p gpurun_out
run() {
    name=$1; lib=$2
    GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=$lib timeout -k 10 300 python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-host-pipeline --no-real-crops > gpurun_out/ab_${tag}_$name.json 2> gpurun_out/ab_${tag}_$name.err || { echo "variant $name failed"; tail -5 gpurun_out/ab_${tag}_$name.err; return 1; }
    python - "$name" gpurun_out/ab_${tag}_$name.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=j["kernels_avg_ms"]
print("%-10s %8.1f patches/s  %.3f ms/step  (one lane %8.1f)  miou %.6f  agree %.7f  frac %.4f" % (sys.argv[1], j["value"], j["ms_per_step"], j["single_lane"]["value"], j["parity"]["miou_vs_reference"], j["parity"]["pixel_agreement"], j["roofline"]["whole_net_frac"]))
print("   " + "  ".join("%s=%.4f" % (n.replace("conv_","").replace("_kernel",""), v["avg_ms"]) for n,v in k.items()))
PY
}
run shipped glomeruli_segmentation_amd/libglomseg.so || exit 1
for v in "$@"; do run $v variants_so/libglomseg_$v.so || exit 1; done
