"""What makes the first 58.5 GB allocation of preprocess() take 1.5 s (BENCH_r02: preprocess 1.53 s, 0.19 s of it kernels)
when the same torch.empty is instant as the first thing a process does (tools/alloc_probe.py)?  Every scenario runs in
a fresh child process: a prelude (what a process has done before), then timed allocations."""
import subprocess
import sys
import time

SCENARIOS = ["first", "kernel", "h2d_pageable", "h2d_pinned", "sort_only", "h2d_then_chunks", "h2d_then_sizes",
             "h2d_then_twice", "h2d_then_threads", "hipmalloc_direct"]


def child(name):
    import numpy as np
    import torch
    GB = 1 << 30
    total = 1829297628 * 32

    def alloc(nbytes, label):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("  [%s] %-28s %6.1f GiB: %.3f s (%.1f ms/GiB)" % (name, label, nbytes / GB, dt, dt * 1e3 / (nbytes / GB)), flush=True)
        return x

    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    if name == "first":
        alloc(total, "first thing")
    elif name == "kernel":
        a = torch.arange(1 << 24, device="cuda")
        (a * 2).sum().item()
        alloc(total, "after elementwise+reduce")
    elif name == "h2d_pageable":
        h = np.random.randint(0, 1 << 30, size=2 * 10**7).astype(np.int32)
        d = torch.from_numpy(h).cuda()
        torch.cuda.synchronize()
        alloc(total, "after 80 MB pageable H2D")
    elif name == "h2d_pinned":
        h = torch.randint(0, 1 << 30, (2 * 10**7,), dtype=torch.int32).pin_memory()
        d = h.cuda(non_blocking=True)
        torch.cuda.synchronize()
        alloc(total, "after 80 MB pinned H2D")
    elif name == "sort_only":
        d = torch.randint(0, 1 << 30, (2 * 10**7,), dtype=torch.int32, device="cuda")
        k = torch.argsort(d)
        torch.cuda.synchronize()
        alloc(total, "after device argsort")
    else:
        h = np.random.randint(0, 1 << 30, size=2 * 10**7).astype(np.int32)
        d = torch.from_numpy(h).cuda()
        torch.cuda.synchronize()
        if name == "h2d_then_chunks":
            t0 = time.perf_counter()
            xs = [torch.empty(total // 16, dtype=torch.uint8, device="cuda") for _ in range(16)]
            torch.cuda.synchronize()
            print("  [%s] 16 chunks of %.1f GiB: %.3f s" % (name, total / 16 / GB, time.perf_counter() - t0), flush=True)
        elif name == "h2d_then_sizes":
            for gb in (1, 4, 16, 32):
                x = alloc(gb * GB, "size scan")
                del x
                torch.cuda.empty_cache()
            alloc(total, "then the full size")
        elif name == "h2d_then_twice":
            x = alloc(total, "first")
            del x
            torch.cuda.empty_cache()
            x = alloc(total, "again after hipFree")
            y = alloc(total, "second buffer beside it")
        elif name == "h2d_then_threads":
            import threading
            out = [None] * 8
            t0 = time.perf_counter()

            def work(i):
                out[i] = torch.empty(total // 8, dtype=torch.uint8, device="cuda")
            th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
            [t.start() for t in th]
            [t.join() for t in th]
            torch.cuda.synchronize()
            print("  [%s] 8 threads x %.1f GiB: %.3f s" % (name, total / 8 / GB, time.perf_counter() - t0), flush=True)
        elif name == "hipmalloc_direct":
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            p = ctypes.c_void_p()
            t0 = time.perf_counter()
            rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(total))
            print("  [%s] hipMalloc direct rc=%d: %.3f s" % (name, rc, time.perf_counter() - t0), flush=True)
            t0 = time.perf_counter()
            rc = hip.hipFree(p)
            print("  [%s] hipFree rc=%d: %.3f s" % (name, rc, time.perf_counter() - t0), flush=True)
            t0 = time.perf_counter()
            rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(total))
            print("  [%s] hipMalloc again rc=%d: %.3f s" % (name, rc, time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for s in (sys.argv[1:] or SCENARIOS):
            r = subprocess.run([sys.executable, __file__, "--child", s], capture_output=True, text=True, timeout=120)
            print((r.stdout or "") + ("".join(l for l in (r.stderr or "").splitlines(True) if "amdgpu.ids" not in l)), end="", flush=True)
