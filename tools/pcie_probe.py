import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from pfb_imaging_amd._lib import DeviceArray, lib, check
a = np.random.default_rng(0).standard_normal((8192, 8192))
d = DeviceArray(a.shape)
for _ in range(2): d.upload(a)
t=time.perf_counter(); 
for _ in range(5): d.upload(a)
t1=(time.perf_counter()-t)/5
out=np.empty_like(a)
d.download(out)
t=time.perf_counter()
for _ in range(5): d.download(out)
t2=(time.perf_counter()-t)/5
print("H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)" % (t1*1e3, a.nbytes/t1/1e9, t2*1e3, a.nbytes/t2/1e9))
