import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gort_amd import api
eng = api.Engine()
eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
wl = np.arange(400.0, 2501.0)
eng.set_spectra(*api.spectra(wl))
n = 3 * (1 << 27)          # 3 GiB of doubles... 402M doubles = 3.2 GB
t0 = time.perf_counter()
b = eng.lut_alloc(n, window=(1 << 20, 2 * (1 << 27)), max_draws=3)
print("lut_alloc %.1f ms" % ((time.perf_counter() - t0) * 1e3), b.placement, "weights", eng.xcd_weights())
t = b.tensor()
print("tensor", t.shape, t.dtype, t.device, hex(t.data_ptr()), hex(b.ptr))
t[:8] = torch.arange(8, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
print("readback", b.to_numpy(8))
b2 = eng.lut_alloc(n, window=(1 << 20, 2 * (1 << 27)), max_draws=3)
print("second call", b2.placement)
del t
b.free(); b2.free()
small = eng.lut_alloc(1000)
print("small", small.placement)
