"""Run the calibration kernels once each (under rocprofv3 --pmc FETCH_SIZE or WRITE_SIZE) and
print the known useful bytes of every kernel as JSON (stdout)."""
import ctypes as C
import json
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libcalib.so"))
GB = 1 << 30
buf = torch.zeros(8 * GB // 4, dtype=torch.float32, device="cuda")   # 8 GiB >> 256 MiB Infinity Cache
out = torch.zeros(16, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
p, o = C.c_void_p(buf.data_ptr()), C.c_void_p(out.data_ptr())
nbytes = buf.numel() * 4
known = {}
lib.run_stream_f4(p, C.c_int64(nbytes), o)
known["calib_stream_f4"] = {"read": nbytes, "write": 0}
threads, iters = 1 << 22, 64
lib.run_gather16(p, C.c_int64(nbytes), C.c_int64(threads), C.c_int(iters), o)
known["calib_gather16"] = {"read": threads * iters * 16, "write": 0, "accesses": threads * iters}
waves, it2 = 1 << 16, 256
lib.run_rows(p, C.c_int64(nbytes), C.c_int64(waves), C.c_int(it2), C.c_int(1), o)
known["calib_rows<true>"] = {"read": waves * it2 * 512, "write": 0}
lib.run_rows(p, C.c_int64(nbytes), C.c_int64(waves), C.c_int(it2), C.c_int(0), o)
known["calib_rows<false>"] = {"read": waves * it2 * 512, "write": 0}
lib.run_atomic_rows(p, C.c_int64(nbytes), C.c_int64(waves), C.c_int(it2))
known["calib_atomic_rows"] = {"read": 0, "write": waves * it2 * 512}
n_rows = 1 << 24
lib.run_store16(p, C.c_int64(n_rows))
known["calib_store16"] = {"read": 0, "write": n_rows * 320}
torch.cuda.synchronize()
print(json.dumps(known))
