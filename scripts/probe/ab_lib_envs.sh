for rep in 1 2 3; do for lib in covo_mpc_amd/csrc/libcovo_hip.so covo_mpc_amd/csrc/libcovo_hip_chainwise.so; do COVO_HIP_LIB=$lib python bench.py --config envs --no-cpu-baseline --no-closed-loop --no-info-leg --no-sweep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], round(d['value']), round(1e3*d['ms_per_step'],1),'us')"; done; done
