"""How long does the allocation of the edge tables take (C3: 58.5 GB of fat slots; round-2 preprocess phases:
alloc 1.63 s of 2.06 s)?  torch.empty of several sizes, fresh vs cached, and two halves from two threads."""
import threading
import time

import torch

torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()


def t_alloc(gb):
    t0 = time.perf_counter()
    x = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x.zero_()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return x, t1 - t0, t2 - t1


for gb in (4, 16, 29, 58):
    x, a, z = t_alloc(gb)
    print("fresh %5.1f GB: alloc %.3fs (%.1f ms/GB)  first zero_ %.3fs" % (gb, a, a / gb * 1e3, z), flush=True)
    del x
    torch.cuda.empty_cache()
x, a, z = t_alloc(58)
del x
x, a2, z2 = t_alloc(58)
print("cached 58 GB: alloc %.4fs" % a2, flush=True)
del x
torch.cuda.empty_cache()
out = [None, None]


def half(i):
    out[i] = torch.empty(29 * (1 << 30), dtype=torch.uint8, device="cuda")


t0 = time.perf_counter()
th = [threading.Thread(target=half, args=(i,)) for i in range(2)]
[t.start() for t in th]
[t.join() for t in th]
torch.cuda.synchronize()
print("two threads x 29 GB: %.3fs" % (time.perf_counter() - t0), flush=True)
