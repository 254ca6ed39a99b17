# usage: ab_levels.sh VAR "levels" [bench flags]: alternates VAR over the levels three times on one box
V=$1; L=$2; shift; shift
for rep in 1 2 3; do for f in $L; do env $V=$f python bench.py --no-cpu-baseline --no-info-leg --no-sweep "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); cl=d.get('closed_loop') or {}; print('$V',$f, round(d['value']), round(1e3*d['ms_per_step'],1),'us  closed loop', round(cl.get('device_env', cl.get('value', 0))))"; done; done
