import sys, time, os
sys.path[:0] = ['/root/repo/global-motion-estimation_amd', '/root/repo']
import numpy as np
import _gme_native as native
ctx = native.Context(0)
n, H, W = 2049, 480, 720
seq = native.Sequence(ctx, n, H, W); seq.synth(1234, 0)
host = native.pinned_empty((n, H, W))
for i in range(n): host[i] = seq.read_frame(i)
seq2 = native.Sequence(ctx, n, H, W)
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); f(); ctx.sync(); best = min(best, time.perf_counter() - t0)
    return best
GB = n * H * W / 1e9
seq2.upload(0, host); 
print("upload (main stream, 64MB chunks + repack): %.1f GB/s" % (GB / t(lambda: seq2.upload(0, host))))
for ch in (2049, 1024, 512, 128, 32):
    seq2.bbme_streamed(host[:min(n, 2*ch+1)], 1, 16, 16, 0, 0, ch)
    dt = t(lambda: seq2.bbme_streamed(host, 1, 16, 16, 0, 0, ch))
    print("streamed chunk %4d: %.2f ms  %.1f GB/s  %.0f pairs/s" % (ch, dt*1e3, GB/dt, (n-1)/dt))
dt = t(lambda: seq2.bbme(1, 16, 16, 0, 0)); print("search only: %.2f ms" % (dt*1e3))
dt = t(lambda: seq2.read_mv()); print("read_mv: %.2f ms" % (dt*1e3))
