for v in 0 151 0 151; do GS_VARIANT=$v GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=variants_so/libglomseg_diag.so python bench.py --steps 30 --warmup 4 --repeats 5 --no-cpu-baseline --no-host-pipeline --no-real-crops 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels_avg_ms']; print('variant', sys.argv[1], d['value'], d['ms_per_step'], 'one lane', d['single_lane']['ms_per_step'], 'l2_down', k['conv_l2_down_branches']['avg_ms'])" $v; done
