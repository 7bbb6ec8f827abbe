# development: the bench step with 1 / 2 / 3 batches in flight (one context + host thread each), and with the kNN kernel held to
# two workgroups per CU (extra dynamic LDS) so that another stream's waves fit beside it
set -e
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --streams $S 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f frames/s  ms_per_step %.2f  knn launch %.2f ms  frac %.3f  single %.3f' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['single_frame']['ms_per_frame']))"; }
S=1 run A=1
S=2 run A=1
S=3 run A=1
S=2 run IMATCH_KNN_DYNLDS=14336
S=2 run IMATCH_KNN_DYNLDS=40960
