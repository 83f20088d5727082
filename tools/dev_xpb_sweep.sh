# dev: knobs of the backward X-panel kernel (name, ms/step, episodes/s, kernel us)
run() { name=$1; shift
  env "$@" python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-as-worded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['roofline']['avg_us'])"
}
for i in 1 2 3; do
run slabs16 X=1
run slabs8 FUMI_XPB_WG=256
run slabs12 FUMI_XPB_WG=384
done
