"""How long does the allocation of the edge tables take (C3: 58.5 GB of fat slots; preprocess phases of round 2:
alloc 1.45-1.63 s of 1.8-2.1 s)?  torch.empty of several sizes and shapes, before and after host-side work."""
import sys
import time

import numpy as np
import torch

torch.zeros(1, device="cuda")
torch.cuda.synchronize()


def t_alloc(shape, dtype, label):
    t0 = time.perf_counter()
    x = torch.empty(shape, dtype=dtype, device="cuda")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    x.view(torch.uint8).reshape(-1)[:: 1 << 21].zero_()       # touch one byte per 2 MiB
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    gb = x.numel() * x.element_size() / 2**30
    print("%-34s %6.1f GiB: empty %.3fs  sync %.3fs  touch-per-2MiB %.3fs   reserved %.1f GB" % (
        label, gb, t1 - t0, t2 - t1, t3 - t2, torch.cuda.memory_reserved() / 1e9), flush=True)
    return x


mode = sys.argv[1] if len(sys.argv) > 1 else "all"
total = 1829297628
if mode in ("all", "plain"):
    x = t_alloc((total, 4), torch.int64, "first thing: int64 (total, 4)")
    del x
    torch.cuda.empty_cache()
    x = t_alloc((total, 4), torch.int64, "again after empty_cache")
    del x
    torch.cuda.empty_cache()
if mode in ("all", "host"):
    a = [np.random.randint(0, 1 << 30, size=1 << 27) for _ in range(8)]      # 8 GiB of host arrays, touched
    small = [torch.empty(1 << 20, device="cuda") for _ in range(200)]
    x = t_alloc((total, 4), torch.int64, "after 8 GiB host arrays + 200 small")
    del x
    torch.cuda.empty_cache()
if mode in ("all", "graph"):
    sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/node2vec-by-ecc_amd")
    from n2v_hip import synth
    cg, info = synth.make_config_graph("C3")
    x = t_alloc((total, 4), torch.int64, "after building the C3 graph on the host")
    del x
    torch.cuda.empty_cache()
    d = torch.from_numpy(cg.col).cuda()
    k = torch.argsort(d, descending=True)
    torch.cuda.synchronize()
    x = t_alloc((total, 4), torch.int64, "after a device argsort of 2e7 keys")
