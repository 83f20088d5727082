for cfg in X=1 FUMI_XPB_NB=2 FUMI_XPB_NB=2,FUMI_XPB_WG=384 X=2 FUMI_XPB_NB=2; do
  env_args=$(echo $cfg | tr ',' ' ')
  for kv in $env_args; do export $kv; done
  echo "== $cfg $(timeout -k 10 100 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-as-worded --no-phase-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  for kv in $env_args; do unset ${kv%%=*}; done
done
