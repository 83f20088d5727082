# dev: kernel times of the X-panel kernels under rocprofv3 for a list of environment settings, e.g.
#   bash tools/dev_run_dbg.sh X=1 FUMI_XP_PS=0 FUMI_XPB_NB=2,FUMI_XPB_WG=384
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  env_args=$(echo $cfg | tr ',' ' ')
  for kv in $env_args; do export $kv; done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/psd_$cfg -o ps -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-as-worded > /dev/null 2>&1
  echo "== $cfg"
  grep -E "xpanel_" $GRAFT_REPO_ROOT/gpurun_out/psd_$cfg/ps_kernel_stats.csv | sed -E 's/^"[^"]*(xpanel_[a-z0-9_]+_kernel)[^"]*"/\1/' | cut -d, -f1-4
  for kv in $env_args; do unset ${kv%%=*}; done
done
