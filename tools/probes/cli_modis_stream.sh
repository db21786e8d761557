python3 - <<'PY'
import numpy as np
rng=np.random.default_rng(3); n=1000000
a=np.stack([rng.uniform(-70,70,n),rng.uniform(0,360,n),rng.uniform(0,75,n),rng.uniform(0,360,n)],1)
with open('/tmp/modis_in.txt','w') as f:
    f.write("%d 7 450 555 645 858.5 1240 1640 2130\n"%n)
    np.savetxt(f,a,fmt="%.4f")
PY
for i in 1 2 3; do GORTT_VERBOSE=1 ./gort_amd/bin/gortt -HB 2 -BR 2 -PCC 0.6 -LAI 3.3 -prnprop < /tmp/modis_in.txt > /tmp/modis_out.txt 2> /tmp/modis_err.txt; tail -3 /tmp/modis_err.txt | cut -c1-260; done; wc -l /tmp/modis_out.txt; head -2 /tmp/modis_out.txt | cut -c1-200
