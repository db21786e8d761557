# fixed cost of one `gortt` process: a one-line input, wall time against the time inside main(), fast and orderly exit
printf '1 4 450 600 800 1000\n10 0 30 20\n' > /tmp/c1.txt
for mode in fast fast fast orderly orderly orderly; do
python3 - "$mode" <<'PY'
import subprocess, time, os, sys
env = dict(os.environ, GORTT_VERBOSE="1")
if sys.argv[1] == "orderly": env["GORTT_ORDERLY_EXIT"] = "1"
t0=time.perf_counter()
r=subprocess.run(["gort_amd/bin/gortt","-LAI","4.0"],stdin=open("/tmp/c1.txt"),capture_output=True,env=env)
dt=time.perf_counter()-t0
print(sys.argv[1], "wall %.3f s rc %d |"%(dt, r.returncode), " | ".join(l[:110] for l in r.stderr.decode().strip().split("\n")))
PY
done
