set -e
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-as-worded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['roofline']['avg_us'])"
}
run base X=1
run t64_wg512 FUMI_XPB_64=1 FUMI_XPB_WG=512
run t64_wg1024 FUMI_XPB_64=1 FUMI_XPB_WG=1024
run t64_wg2048 FUMI_XPB_64=1 FUMI_XPB_WG=2048
run w256_wg256 FUMI_XPB_WG=256
run w256_wg384 FUMI_XPB_WG=384
run w256_wg768 FUMI_XPB_WG=768
