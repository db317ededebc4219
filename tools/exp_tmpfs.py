"""Experiment: what does the box's /dev/shm take from N writer threads (5.5 MB files, like LNA outputs)?"""
import os, sys, threading, time, shutil
d = "/dev/shm/aasr_wtest"
shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
buf = bytes(5_500_000)
def work(t, n):
    for i in range(n):
        p = "%s/t%d_%d.lna" % (d, t, i)
        fd = os.open(p + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.write(fd, buf); os.close(fd); os.rename(p + ".tmp", p)
for nt in (1, 2, 4, 8, 12, 16, 32):
    per = 1600 // nt
    th = [threading.Thread(target=work, args=(t, per)) for t in range(nt)]
    t0 = time.time(); [x.start() for x in th]; [x.join() for x in th]; dt = time.time() - t0
    print("threads %2d: %.2f GB/s (fresh files)" % (nt, nt * per * len(buf) / dt / 1e9), flush=True)
    th = [threading.Thread(target=work, args=(t, per)) for t in range(nt)]
    t0 = time.time(); [x.start() for x in th]; [x.join() for x in th]; dt = time.time() - t0
    print("threads %2d: %.2f GB/s (rewriting)" % (nt, nt * per * len(buf) / dt / 1e9), flush=True)
    shutil.rmtree(d); os.makedirs(d)
shutil.rmtree(d, ignore_errors=True)
