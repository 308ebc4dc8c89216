#!/usr/bin/env python3
"""Cold-disk read rates on a fresh box (first thing of a gpurun call): one sequential reader against N parallel readers, on files
nobody has touched yet.  probe_readpar.py"""
import os, sys, time, threading, glob
def seq(path, nbytes):
    fd = os.open(path, os.O_RDONLY); t0 = time.perf_counter(); off = 0
    while off < nbytes:
        b = os.pread(fd, 1 << 20, off)
        if not b: break
        off += len(b)
    os.close(fd); return off, time.perf_counter() - t0
def par(path, nbytes, nthreads, chunk=256 << 10):
    fd = os.open(path, os.O_RDONLY); t0 = time.perf_counter()
    nchunks = (nbytes + chunk - 1) // chunk
    def work(k):
        for c in range(k, nchunks, nthreads):
            os.pread(fd, chunk, c * chunk)
    th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    [t.start() for t in th]; [t.join() for t in th]
    os.close(fd); return nbytes, time.perf_counter() - t0
def fadv(path, nbytes):
    fd = os.open(path, os.O_RDONLY); t0 = time.perf_counter()
    for o in range(0, nbytes, 2 << 20): os.posix_fadvise(fd, o, 2 << 20, os.POSIX_FADV_WILLNEED)
    t1 = time.perf_counter() - t0
    # wait until resident: read it
    off = 0
    while off < nbytes:
        b = os.pread(fd, 1 << 20, off); off += len(b)
        if not b: break
    os.close(fd); return t1, time.perf_counter() - t0
big = sorted((os.path.getsize(p), p) for p in glob.glob("/opt/rocm/lib/lib*.so*") if os.path.isfile(p) and not os.path.islink(p) and os.path.getsize(p) > (60 << 20))
print("candidates:", [(s >> 20, os.path.basename(p)) for s, p in big][:8])
N = 32 << 20
tests = [("sequential 1 MB reads", lambda p: seq(p, N)), ("16 parallel readers", lambda p: par(p, N, 16)), ("64 parallel readers", lambda p: par(p, N, 64)), ("fadvise WILLNEED then read", lambda p: fadv(p, N))]
for (name, fn), (sz, path) in zip(tests, big):
    r = fn(path)
    print(f"{name:28s} on {os.path.basename(path):40s}: {r}")
hip = "/opt/rocm/lib/libamdhip64.so.7"
hip = os.path.realpath(hip)
print("libamdhip64", os.path.getsize(hip) >> 20, "MB: 16 parallel readers:", par(hip, os.path.getsize(hip), 16))
