# whole-batch solve under layout variants (the summation order is a free parameter; each variant re-rolls the slowest instance)
run() { echo -n "$1: "; env $1 python bench.py --cpu-sample 0 --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f M it/s  %.1f ms  max_iters %d  us/iter %.2f' % (d['value']/1e6, d['ms_per_step'], d['detail']['max_outer_iters'], d['detail']['us_per_outer_iteration_slowest_instance']))"; }
run X=1; run LPBOX_LP_SPLITBIAS=0; run LPBOX_LP_SPLITBIAS=1; run LPBOX_LP_SPLITBIAS=3; run LPBOX_LP_BANKAWARE=1; run "LPBOX_LP_BANKAWARE=1 LPBOX_LP_SPLITBIAS=1"; run "LPBOX_LP_BANKAWARE=1 LPBOX_LP_SPLITBIAS=0"
