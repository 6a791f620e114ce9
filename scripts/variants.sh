cd ${GRAFT_REPO_ROOT:-.}
run() {
  echo "== $*"
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d['roofline_families']
print('steps/s %.1f ms %.3f | solve %.3f factor %.3f taylor %.3f io %.3f tail %.3f | levels %d'%(d['value'],d['ms_per_step'],f['solve']['ms_per_step'],f['factor']['ms_per_step'],f['taylor']['ms_per_step'],f['io']['ms_per_step'],f['tail']['ms_per_step'],d['config']['solver_stats']['nr_level']))"
}
for v in "$@"; do run $v; done
