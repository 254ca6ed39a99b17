"""bench.py with another split of the Sigma chain's phases between separate launches and the two persistent tail launches
(covo_debug_set_ns_tail) and with / without the deflation (covo_debug_set_ns_deflate).
usage: tail_bench.py <tail squarings> <tail iterations> <deflate 0|1> [bench.py flags]"""
import sys, os, json, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from covo_mpc_amd import _lib
sq, it, defl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
# (per-handle switches since round 6: every handle bench.py creates takes these defaults from the environment, csrc/step.hip)
os.environ["COVO_NS_TAIL"] = f"{sq},{it}"
os.environ["COVO_NS_DEFLATE"] = str(defl)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-info-leg"] + sys.argv[4:]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
cl = d.get("closed_loop") or {}
print(f"tail sq={sq} it={it} deflate={defl}: value {d['value']:.0f} {d['unit'].split()[0]} ({1e3*d['ms_per_step']:.1f} us)  closed loop {cl.get('device_env', cl.get('value', 0)):.0f}"
      + (f"  sigma {d['phases_us_per_batched_step']['sigma_us']:.1f} us" if "phases_us_per_batched_step" in d else ""))
