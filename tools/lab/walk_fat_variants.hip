// LAB COPY (round 2) of csrc/n2v_walk_fat.hip with the A/B variants selected by N2V_WALK_VARIANT — not built into the
// product library; see tools/lab/README.md.
// "Fat slot" walk — gfx950 (MI355X): one 32-byte gather per step.
//
// Same walk as n2v_walk (src/node2vec.py:55-95, :271-281) over a table layout that spends
// HBM capacity to halve the number of memory requests: alias slot k of a table carries, next
// to q, the walk records of BOTH outcomes of the draw (neighbour k and neighbour J[k]):
//
//     fat = fat_slots[table + floor(u1*K)];          one aligned 32-B load
//     rec = (u2 < fat.q) ? fat.keep : fat.alias;     (:278-281)
//     table = rec.slot, K = rec.deg, node = rec.dst
//
// The thin layout costs two dependent 16-B gathers per step and is bound by the chip's
// random-request rate (each gather is one 64-B request); a 32-B aligned slot
// never straddles a 64-B line, so a step is one request.  Memory: 32 B per alias slot, i.e.
// 2x the thin tables (C3: 58.5 GB) — the reason this layout only makes sense on a 288 GB part.
// Tables are expanded from the thin, reference-exact tables, so the walks are bit-identical.
#include "n2v_common.h"

#include <cstdlib>

namespace {

struct FatArgs {
    const int64_t* row_ptr;
    const n2v_fat_slot* node_fat;
    const n2v_fat_slot* fat;
    const int32_t* starts;
    int64_t n_starts, pos_begin, pos_count, round_begin, n_local;
    int32_t L;
    const double* uniforms;
    const int64_t* walk_uoff;
    uint64_t seed;
    int32_t* walks;
    int32_t* lens;
};

template <int RNG, bool VEC4>
__global__ void __launch_bounds__(256) walk_fat_kernel(FatArgs a) {
    const int64_t lw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (lw >= a.n_local) return;
    const int64_t rl = lw / a.pos_count, pl = lw - rl * a.pos_count;
    const uint64_t gw = (uint64_t)((a.round_begin + rl) * a.n_starts + a.pos_begin + pl);
    const int32_t L = a.L;
    const int32_t cur0 = a.starts[a.pos_begin + pl];
    const int64_t b0 = a.row_ptr[cur0], b1 = a.row_ptr[cur0 + 1];
    const n2v_fat_slot* tab = a.node_fat + b0;  // first step: node table (:69-70)
    uint32_t K = (uint32_t)(b1 - b0);
    int32_t len = 1;
    uint32_t t = 0;
    const double* up = nullptr;
    if (RNG == N2V_RNG_UNIFORMS) up = a.uniforms + (a.walk_uoff ? a.walk_uoff[lw] : (int64_t)2 * (L - 1) * lw);

    auto step = [&]() -> int32_t {
        if (K == 0) return -1;  // dead end: stop, consume nothing (:76-77)
        double u1, u2;
        if (RNG == N2V_RNG_UNIFORMS) {
            const double2 u = *reinterpret_cast<const double2*>(up + 2 * (int64_t)t);
            u1 = u.x; u2 = u.y;
        } else {
            n2v::philox_uniforms(a.seed, gw, t, u1, u2);
        }
        ++t;
        const uint32_t kk = (uint32_t)(u1 * (double)K);  // :277
        const uint4* p = reinterpret_cast<const uint4*>(tab + kk);
        const uint4 lo = p[0], hi = p[1];  // {q.lo, q.hi, keep.slot_lo, keep.deg_hi | keep.dst, alias.slot_lo, alias.deg_hi, alias.dst}
        const double q = __hiloint2double((int)lo.y, (int)lo.x);
        const bool keep = u2 < q;  // :278
        const uint32_t slot_lo = keep ? lo.z : hi.y;
        const uint32_t deg_hi = keep ? lo.w : hi.z;
        const uint32_t dst = keep ? hi.x : hi.w;
        tab = a.fat + (((uint64_t)(deg_hi >> 24) << 32) | slot_lo);
        K = deg_hi & 0xFFFFFFu;
        ++len;
        return (int32_t)dst;
    };

    int32_t* out = a.walks + lw * (int64_t)L;
    if (VEC4) {
        int4 o;
        o.x = cur0; o.y = step(); o.z = step(); o.w = step();
        *reinterpret_cast<int4*>(out) = o;
        for (int32_t g = 4; g < L; g += 4) {
            o.x = step(); o.y = step(); o.z = step(); o.w = step();
            *reinterpret_cast<int4*>(out + g) = o;
        }
    } else {
        out[0] = cur0;
        for (int32_t i = 1; i < L; ++i) out[i] = step();
    }
    a.lens[lw] = len;
}

// Round-2 variant.  tools/lab/gather_lab*.hip: the chip serves ~4.9e10 random 16-B gathers/s from a 56 GB table but
// only 3.8e10 32-B slots/s when a lane fetches its slot with two dwordx4 loads (each load instruction is its own
// request to the same 64-B line), and 16-B pieces of output rows cost a partial-line write each.  So here
//   PAIR : lanes 2i and 2i+1 fetch the two halves of ONE slot with ONE load instruction — 32 contiguous bytes per
//          lane pair, i.e. one request per walk step — first for the even lane's walk, then for the odd lane's, and
//          swap the halves they fetched for each other through DPP (quad_perm [1,0,3,2]);
//   BURST: node ids leave the lane as whole 64-B lines (16 steps buffered in registers).
// Same draws, same order of operations per walk: the walks are bit-identical to walk_fat_kernel's.
__device__ __forceinline__ uint32_t dpp_swap1(uint32_t v) {   // value of lane ^ 1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint4 dpp_swap1(uint4 v) {
    return make_uint4(dpp_swap1(v.x), dpp_swap1(v.y), dpp_swap1(v.z), dpp_swap1(v.w));
}

template <int RNG, bool PAIR, int BURST, bool NT = false>
__global__ void __launch_bounds__(256) walk_fat2_kernel(FatArgs a) {
    const int64_t lw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool mine = lw < a.n_local;          // the last pair may have one lane without a walk: it still helps its partner
    const int32_t L = a.L;
    int64_t rl = 0, pl = 0;
    if (mine) { rl = lw / a.pos_count; pl = lw - rl * a.pos_count; }
    const uint64_t gw = (uint64_t)((a.round_begin + rl) * a.n_starts + a.pos_begin + pl);
    const int32_t cur0 = mine ? a.starts[a.pos_begin + pl] : 0;
    int64_t b0 = 0, b1 = 0;
    if (mine) { b0 = a.row_ptr[cur0]; b1 = a.row_ptr[cur0 + 1]; }
    const n2v_fat_slot* tab = a.node_fat + b0;  // first step: node table (:69-70)
    uint32_t K = (uint32_t)(b1 - b0);
    int32_t len = 1;
    uint32_t t = 0;
    const double* up = nullptr;
    if (RNG == N2V_RNG_UNIFORMS && mine) up = a.uniforms + (a.walk_uoff ? a.walk_uoff[lw] : (int64_t)2 * (L - 1) * lw);
    const int odd = threadIdx.x & 1;

    auto step = [&]() -> int32_t {
        const bool live = mine && K != 0;       // dead end: stop, consume nothing (:76-77)
        double u1 = 0.0, u2 = 0.0;
        if (live) {
            if (RNG == N2V_RNG_UNIFORMS) {
                const double2 u = *reinterpret_cast<const double2*>(up + 2 * (int64_t)t);
                u1 = u.x; u2 = u.y;
            } else {
                n2v::philox_uniforms(a.seed, gw, t, u1, u2);
            }
            ++t;
        }
        const uint32_t kk = (uint32_t)(u1 * (double)K);  // :277
        const uint64_t addr = live ? (uint64_t)(uintptr_t)(tab + kk) : 0ull;
        uint4 lo, hi;
        if (PAIR) {
            const uint64_t other = ((uint64_t)dpp_swap1((uint32_t)(addr >> 32)) << 32) | dpp_swap1((uint32_t)addr);
            const uint64_t even_addr = odd ? other : addr, odd_addr = odd ? addr : other;
            uint4 x = make_uint4(0, 0, 0, 0), y = x;
            if (even_addr) x = *reinterpret_cast<const uint4*>(even_addr + (odd ? 16 : 0));   // the pair reads 32 contiguous bytes
            if (odd_addr) y = *reinterpret_cast<const uint4*>(odd_addr + (odd ? 16 : 0));
            const uint4 got = dpp_swap1(odd ? x : y);
            lo = odd ? got : x;
            hi = odd ? y : got;
        } else {
            lo = hi = make_uint4(0, 0, 0, 0);
            if (live) { const uint4* p = reinterpret_cast<const uint4*>(addr); lo = p[0]; hi = p[1]; }
        }
        if (!live) return -1;
        const double q = __hiloint2double((int)lo.y, (int)lo.x);
        const bool keep = u2 < q;  // :278
        const uint32_t slot_lo = keep ? lo.z : hi.y;
        const uint32_t deg_hi = keep ? lo.w : hi.z;
        const uint32_t dst = keep ? hi.x : hi.w;
        tab = a.fat + (((uint64_t)(deg_hi >> 24) << 32) | slot_lo);
        K = deg_hi & 0xFFFFFFu;
        ++len;
        return (int32_t)dst;
    };

    int32_t* out = a.walks + lw * (int64_t)L;
    int32_t buf[BURST];
    buf[0] = cur0;
#pragma unroll
    for (int i = 1; i < BURST; ++i) buf[i] = step();
    for (int32_t g = 0;;) {
        if (mine) {
#pragma unroll
            for (int i = 0; i < BURST; i += 4) {
                typedef int v4i __attribute__((ext_vector_type(4)));
                v4i v = {buf[i], buf[i + 1], buf[i + 2], buf[i + 3]};
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(out + g + i));   // streamed: never re-read by this kernel
                else *reinterpret_cast<v4i*>(out + g + i) = v;
            }
        }
        g += BURST;
        if (g >= L) break;
#pragma unroll
        for (int i = 0; i < BURST; ++i) buf[i] = step();
    }
    if (mine) a.lens[lw] = len;
}

// One wavefront expands one table: fat[t+k] = {thin[t+k].q, rec(base+k), rec(base+J[k])} where
// base = row_ptr[node the table draws from] and rec(e) is the thin walk record of CSR entry e.
__global__ void __launch_bounds__(256)
fat_expand_kernel(int64_t n_tables, const int64_t* __restrict__ tab_off, const int32_t* __restrict__ tab_node,
                  const int64_t* __restrict__ row_ptr, const n2v_alias_slot* __restrict__ thin,
                  const n2v_edge_rec* __restrict__ recs, n2v_fat_slot* __restrict__ fat) {
    const int lane = threadIdx.x & 63;
    const int64_t tb = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (tb >= n_tables) return;
    const int64_t t0 = tab_off[tb];
    const int32_t node = tab_node ? tab_node[tb] : (int32_t)tb;
    const int64_t base = row_ptr[node];
    const int K = (int)(row_ptr[node + 1] - base);
    for (int k = lane; k < K; k += 64) {
        const n2v_alias_slot s = thin[t0 + k];
        const uint4 ra = *reinterpret_cast<const uint4*>(recs + base + k);
        const uint4 rb = *reinterpret_cast<const uint4*>(recs + base + s.J);
        uint4 lo, hi;  // rec = {slot_lo, base, dst, deg_hi}; the row base is not needed any more
        lo.x = (uint32_t)__double2loint(s.q);
        lo.y = (uint32_t)__double2hiint(s.q);
        lo.z = ra.x; lo.w = ra.w;
        hi.x = ra.z;
        hi.y = rb.x; hi.z = rb.w; hi.w = rb.z;
        uint4* o = reinterpret_cast<uint4*>(fat + t0 + k);
        o[0] = lo;
        o[1] = hi;
    }
}

}  // namespace

extern "C" int n2v_build_fat_slots(int64_t n_tables, const int64_t* tab_off, const int32_t* tab_node,
                                   const int64_t* row_ptr, const n2v_alias_slot* thin, const n2v_edge_rec* recs,
                                   n2v_fat_slot* fat, void* stream) {
    if (n_tables < 0) return n2v::fail(N2V_ERR_INVALID, "n2v_build_fat_slots: negative count");
    if (n_tables == 0) return N2V_OK;
    if (!tab_off || !row_ptr || !thin || !recs || !fat)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_fat_slots: null pointer");
    if (((uintptr_t)fat & 31) != 0) return n2v::fail(N2V_ERR_INVALID, "n2v_build_fat_slots: fat slots not 32-byte aligned");
    const int64_t blocks = (n_tables + 3) / 4;
    if (blocks > 0x7fffffff) return n2v::fail(N2V_ERR_INVALID, "n2v_build_fat_slots: too many tables in one call");
    hipLaunchKernelGGL(fat_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_tables, tab_off,
                       tab_node, row_ptr, thin, recs, fat);
    return n2v::check_launch("n2v_build_fat_slots");
}

extern "C" int n2v_walk_fat(const int64_t* row_ptr, const n2v_fat_slot* node_fat, const n2v_fat_slot* fat,
                            const int32_t* starts, int64_t n_starts, int64_t pos_begin, int64_t pos_count,
                            int64_t round_begin, int64_t round_count, int32_t walk_length, int32_t rng_mode,
                            const double* uniforms, const int64_t* walk_uoff, int64_t /*uoff_round_stride: product only*/,
                            uint64_t seed, int32_t* walks, int32_t* lens, void* stream) {
    if (pos_count < 0 || round_count < 0 || pos_begin < 0 || round_begin < 0 || walk_length < 1 ||
        pos_begin + pos_count > n_starts)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_fat: bad shard or length");
    const int64_t n_local = pos_count * round_count;
    if (n_local == 0) return N2V_OK;
    if (!row_ptr || !node_fat || !starts || !lens || !walks || (walk_length > 1 && !fat))
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_fat: null pointer");
    if (rng_mode != N2V_RNG_UNIFORMS && rng_mode != N2V_RNG_PHILOX)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_fat: rng_mode %d", (int)rng_mode);
    if (rng_mode == N2V_RNG_UNIFORMS && walk_length > 1 && !uniforms)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_fat: parity mode needs a uniform buffer");
    if (((uintptr_t)uniforms & 15) != 0 || ((uintptr_t)node_fat & 31) != 0 || ((uintptr_t)fat & 31) != 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_fat: misaligned buffer");
    if (n_local > (int64_t)0x7fffffff * 256) return n2v::fail(N2V_ERR_INVALID, "n2v_walk_fat: too many walks in one call");
    FatArgs a{row_ptr, node_fat, fat, starts, n_starts, pos_begin, pos_count, round_begin, n_local, walk_length,
              uniforms, walk_uoff, seed, walks, lens};
    const bool vec4 = walk_length >= 4 && (walk_length % 4) == 0 && ((uintptr_t)walks & 15) == 0;
    const dim3 grid(n2v::grid_for(n_local, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    // default: pair-cooperative slot fetch + 64-B output lines whenever the row length allows it (L % 16 == 0);
    // N2V_WALK_VARIANT = 0 (round 1's kernel) / 1 (pair, 16-B pieces) / 2 (pair, 64-B lines) / 3 (two loads, 64-B lines) /
    // 4 (pair, 64-B lines, nontemporal stores)
    // is a tuning switch for tools/walk_probe.py
    const char* env = getenv("N2V_WALK_VARIANT");
    int variant = env ? atoi(env) : 2;
    if ((variant == 2 || variant == 3 || variant == 4) && !(walk_length % 16 == 0 && ((uintptr_t)walks & 63) == 0)) variant = 1;
    if (variant == 1 && !vec4) variant = 0;
    const bool par = rng_mode == N2V_RNG_UNIFORMS;
    if (variant == 1) {
        if (par) hipLaunchKernelGGL((walk_fat2_kernel<N2V_RNG_UNIFORMS, true, 4>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((walk_fat2_kernel<N2V_RNG_PHILOX, true, 4>), grid, block, 0, st, a);
    } else if (variant == 2) {
        if (par) hipLaunchKernelGGL((walk_fat2_kernel<N2V_RNG_UNIFORMS, true, 16>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((walk_fat2_kernel<N2V_RNG_PHILOX, true, 16>), grid, block, 0, st, a);
    } else if (variant == 4) {
        if (par) hipLaunchKernelGGL((walk_fat2_kernel<N2V_RNG_UNIFORMS, true, 16, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((walk_fat2_kernel<N2V_RNG_PHILOX, true, 16, true>), grid, block, 0, st, a);
    } else if (variant == 3) {
        if (par) hipLaunchKernelGGL((walk_fat2_kernel<N2V_RNG_UNIFORMS, false, 16>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((walk_fat2_kernel<N2V_RNG_PHILOX, false, 16>), grid, block, 0, st, a);
    } else if (par) {
        if (vec4) hipLaunchKernelGGL((walk_fat_kernel<N2V_RNG_UNIFORMS, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((walk_fat_kernel<N2V_RNG_UNIFORMS, false>), grid, block, 0, st, a);
    } else {
        if (vec4) hipLaunchKernelGGL((walk_fat_kernel<N2V_RNG_PHILOX, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((walk_fat_kernel<N2V_RNG_PHILOX, false>), grid, block, 0, st, a);
    }
    return n2v::check_launch("n2v_walk_fat");
}
