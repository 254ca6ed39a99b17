# usage: ab_env.sh VAR [bench flags]: alternates VAR=1 / VAR=0 three times on one box
V=$1; shift
for rep in 1 2 3; do for f in 1 0; do env $V=$f python bench.py --no-cpu-baseline --no-info-leg --no-sweep "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); cl=d.get('closed_loop') or {}; print('$V',$f, round(d['value']), round(1e3*d['ms_per_step'],1),'us  closed loop', round(cl.get('device_env', cl.get('value', 0))))"; done; done
