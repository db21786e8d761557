printf '1 4 450 600 800 1000\n10 0 30 20\n' > /tmp/c1.txt
for i in 1 2 3 4 5; do
python3 - <<'PY'
import subprocess, time, os
t0=time.perf_counter()
r=subprocess.run(["gort_amd/bin/gortt","-LAI","4.0"],stdin=open("/tmp/c1.txt"),capture_output=True,env=dict(os.environ,GORTT_VERBOSE="1"))
dt=time.perf_counter()-t0
print("wall %.3f s |"%dt, r.stderr.decode().strip()[:200])
PY
done
# how long does a process that only loads the HIP runtime and creates a context take
cat > /tmp/hipinit.cpp <<'CPP'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main(){ auto t0=std::chrono::steady_clock::now(); int n=0; hipGetDeviceCount(&n); auto t1=std::chrono::steady_clock::now(); hipFree(0); auto t2=std::chrono::steady_clock::now();
 void*p; hipHostMalloc(&p, 150u<<20, 0); auto t3=std::chrono::steady_clock::now();
 printf("count %.3f ctx %.3f pin150MB %.3f\n", std::chrono::duration<double>(t1-t0).count(), std::chrono::duration<double>(t2-t1).count(), std::chrono::duration<double>(t3-t2).count()); return 0; }
CPP
/opt/rocm/bin/hipcc -O2 /tmp/hipinit.cpp -o /tmp/hipinit 2>/dev/null
for i in 1 2 3; do python3 - <<'PY'
import subprocess, time
t0=time.perf_counter(); r=subprocess.run(["/tmp/hipinit"],capture_output=True); dt=time.perf_counter()-t0
print("hipinit wall %.3f s |"%dt, r.stdout.decode().strip())
PY
done
