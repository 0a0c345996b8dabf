"""Raw page-cache write bandwidth of the box (16 threads x 256 MB pwrite): the ceiling the native CSV writer is held against."""
import os, time, threading, numpy as np
N=16; SZ=256*1024*1024
buf=np.random.default_rng(0).integers(0,255,SZ,dtype=np.uint8).tobytes()
for rep in range(2):
    fd=os.open("/tmp/pw_test.bin", os.O_CREAT|os.O_WRONLY|os.O_TRUNC, 0o644)
    t0=time.perf_counter()
    th=[threading.Thread(target=lambda i=i: os.pwrite(fd, buf, i*SZ)) for i in range(N)]
    [t.start() for t in th]; [t.join() for t in th]
    t1=time.perf_counter(); os.close(fd)
    print("parallel pwrite", N*SZ/1e9, "GB in", round(t1-t0,3), "s ->", round(N*SZ/1e9/(t1-t0),2), "GB/s")
    os.remove("/tmp/pw_test.bin")
fd=os.open("/tmp/pw_test.bin", os.O_CREAT|os.O_WRONLY|os.O_TRUNC, 0o644)
t0=time.perf_counter()
for i in range(4): os.pwrite(fd, buf, i*SZ)
print("serial pwrite", round(4*SZ/1e9/(time.perf_counter()-t0),2), "GB/s"); os.close(fd); os.remove("/tmp/pw_test.bin")
