// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the
// walk and SGNS kernels (MI355X_MICROARCH.md, HBM section: only 16-B/lane streaming reads are
// calibrated there; "calibrate on a known byte count in your own access pattern").
// Every kernel moves a known number of useful bytes through a buffer far larger than the
// 256 MiB Infinity Cache; tools/calib/run_calib.py pairs them with the counter values.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
    return x;
}

// 1. coalesced 16 B/lane streaming read
__global__ void calib_stream_f4(const float4* __restrict__ buf, int64_t n_vec, float* out) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = buf[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// 2. one random aligned 16-B load per lane per iteration (the walk's slot / record gathers)
__global__ void calib_gather16(const uint4* __restrict__ buf, uint64_t n_slots_mask, int iters, uint32_t* out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint4 v = buf[mix64(tid * 1000003ULL + it) & n_slots_mask];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// 3./4. one random 512-B row per wave per iteration: 2 x dword sc1 loads per lane (SGNS atomic
// mode) or one float2 plain load per lane (SGNS plain mode)
template <bool SC1>
__global__ void calib_rows(const float* __restrict__ tab, uint64_t n_rows_mask, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const uint64_t row = mix64(wave * 1000003ULL + it) & n_rows_mask;
        const float* p = tab + row * 128;
        if (SC1) {
            acc += __hip_atomic_load(p + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += __hip_atomic_load(p + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const float2 v = *reinterpret_cast<const float2*>(p + 2 * lane);
            acc += v.x + v.y;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
template __global__ void calib_rows<true>(const float*, uint64_t, int, float*);
template __global__ void calib_rows<false>(const float*, uint64_t, int, float*);

// 5. one random 512-B row per wave per iteration updated by float atomic adds (256 contiguous
// bytes per wave-instruction)
__global__ void calib_atomic_rows(float* tab, uint64_t n_rows_mask, int iters) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int it = 0; it < iters; ++it) {
        const uint64_t row = mix64(wave * 7919ULL + it) & n_rows_mask;
        float* p = tab + row * 128;
        __hip_atomic_fetch_add(p + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(p + 64 + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// 6. every lane writes its own 320-B row in 16-B pieces (the walk's output pattern)
__global__ void calib_store16(int4* buf, int64_t n_rows) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    for (int g = 0; g < 20; ++g) buf[r * 20 + g] = make_int4((int)r, g, 0, 0);
}

extern "C" {
void run_stream_f4(const void* buf, int64_t bytes, void* out) {
    hipLaunchKernelGGL(calib_stream_f4, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, (float*)out);
}
void run_gather16(const void* buf, int64_t bytes, int64_t threads, int iters, void* out) {
    hipLaunchKernelGGL(calib_gather16, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, (const uint4*)buf,
                       (uint64_t)(bytes / 16 - 1), iters, (uint32_t*)out);
}
void run_rows(const void* tab, int64_t bytes, int64_t waves, int iters, int sc1, void* out) {
    if (sc1) hipLaunchKernelGGL(calib_rows<true>, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, (const float*)tab,
                                (uint64_t)(bytes / 512 - 1), iters, (float*)out);
    else hipLaunchKernelGGL(calib_rows<false>, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, (const float*)tab,
                            (uint64_t)(bytes / 512 - 1), iters, (float*)out);
}
void run_atomic_rows(void* tab, int64_t bytes, int64_t waves, int iters) {
    hipLaunchKernelGGL(calib_atomic_rows, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, (float*)tab,
                       (uint64_t)(bytes / 512 - 1), iters);
}
void run_store16(void* buf, int64_t n_rows) {
    hipLaunchKernelGGL(calib_store16, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, 0, (int4*)buf, n_rows);
}
}
