#!/bin/bash
# Interleaved same-box A/B on the GPU box: the shipped library and each named variant (variants_so/libglomseg_<name>.so) are
# benched ROUNDS times in turn (A B A B ...), so that drift of the box shows in both; prints per-variant medians of the two-lane
# and one-lane step and of every kernel.      tools/ab2.sh <tag> [rounds] name1 [name2 ...]
tag=$1; shift
rounds=3
if [[ $1 =~ ^[0-9]+$ ]]; then rounds=$1; shift; fi
mkdir -p gpurun_out
names=(shipped "$@")
for r in $(seq 1 $rounds); do
    for name in "${names[@]}"; do
        lib=variants_so/libglomseg_$name.so
        [ $name = shipped ] && lib=glomeruli_segmentation_amd/libglomseg.so
        GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=$lib timeout -k 10 300 python bench.py --steps 40 --warmup 5 --repeats 5 --no-cpu-baseline --no-host-pipeline --no-real-crops > gpurun_out/ab2_${tag}_${name}_$r.json 2> gpurun_out/ab2_${tag}_${name}_$r.err || { echo "variant $name failed"; tail -5 gpurun_out/ab2_${tag}_${name}_$r.err; exit 1; }
    done
done
python - "$tag" "$rounds" "${names[@]}" <<'PY'
import json, statistics, sys
tag, rounds, names = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for name in names:
    two, one, kern, agree = [], [], {}, []
    for r in range(1, rounds + 1):
        j = json.loads(open("gpurun_out/ab2_%s_%s_%d.json" % (tag, name, r)).read().strip().splitlines()[-1])
        two.append(j["two_lanes"]["ms_per_step"]); one.append(j["single_lane"]["ms_per_step"]); agree.append(j["parity"]["pixel_agreement"])
        for k, v in j["kernels_avg_ms"].items():
            kern.setdefault(k, []).append(v["avg_ms"])
    print("%-12s two lanes %s median %.4f ms   one lane %s median %.4f ms   agree %.7f" % (
        name, " ".join("%.3f" % v for v in two), statistics.median(two), " ".join("%.3f" % v for v in one), statistics.median(one), min(agree)))
    print("   " + "  ".join("%s=%.4f" % (k.replace("conv_", "").replace("_kernel", ""), statistics.median(v)) for k, v in kern.items()))
PY
