# dev: knobs of the backward X-panel kernel (name, ms/step, episodes/s, kernel us)
run() { name=$1; shift
  env "$@" python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-as-worded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['roofline']['avg_us'])"
}
run sb_nst2 X=1
run sb_nst3 FUMI_XPB_NST=3
run sb_nst2_wg256 FUMI_XPB_WG=256
run sb_nst2_wg384 FUMI_XPB_WG=384
run sb_nst2_wg1024 FUMI_XPB_WG=1024
run sb_nst3_wg256 FUMI_XPB_NST=3 FUMI_XPB_WG=256
run fp32 FUMI_XPB_SB=0
