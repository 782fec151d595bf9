cd $GRAFT_REPO_ROOT
for i in 1 2; do for L in mid_r5 pre_heavy base; do
  if [ "$L" == "base" ]; then LP=gs-2m_amd/csrc/libgs2m_raster.so; else LP=gs-2m_amd/csrc/variants/lib$L.so; fi
  GS2M_LIB=$LP python3 bench.py --config c1 --no-cpu-baseline --no-caller-levels 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['ms_per_step'], d['median_ms_per_step'], d['clock_ramp']['ms_per_step_at_start'], d.get('reference_binning_ms_per_step'))"
done; done
