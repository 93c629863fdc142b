// Shared host-side helpers for the C-ABI implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "n2v_hip.h"

namespace n2v {

inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(N2V_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return N2V_OK;
}

inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

}  // namespace n2v
