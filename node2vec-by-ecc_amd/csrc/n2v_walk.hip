// 2nd-order p,q-biased random walk over CSR + alias tables — gfx950 (MI355X) kernels.
//
// Replaces Graph.node2vec_walk / simulate_walks / alias_draw of the reference
// (src/node2vec.py:55-95, :271-281).  One lane owns one walk: walks are independent
// given read-only tables, consecutive lanes take consecutive start positions, and a
// step is two dependent 16-byte gathers:
//
//     slot = slots[table(prev->cur) + floor(u1 * deg(cur))]     alias draw  (:277-281)
//     rec  = recs[row_ptr[cur] + (u2 < slot.q ? kk : slot.J)]   move to the chosen neighbour
//
// `rec` carries everything the next step needs (next table, row base, degree, node id),
// so no other memory is touched.  The walk is gather-bound (no MFMA); latency is hidden
// by having every lane of the chip own a walk (>= 2048 lanes per CU).  Node ids leave the
// lane as 16-byte stores (4 steps buffered in registers) so rows are written in whole
// 16-B pieces.
#include "n2v_common.h"

namespace {

struct WalkArgs {
    const int64_t* row_ptr;
    const n2v_alias_slot* node_slots;
    const n2v_edge_rec* recs;
    const n2v_alias_slot* slots;
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
__global__ void __launch_bounds__(256) walk_kernel(WalkArgs a) {
    const int64_t lw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (lw >= a.n_local) return;
    const int64_t rl = lw / a.pos_count, pl = lw - rl * a.pos_count;
    const uint64_t gw = (uint64_t)((a.round_begin + rl) * a.n_starts + a.pos_begin + pl);
    const int32_t L = a.L;

    int32_t cur = a.starts[a.pos_begin + pl];
    const int64_t b0 = a.row_ptr[cur], b1 = a.row_ptr[cur + 1];
    // state of the walk: where the current node's table is, its row base and degree
    const n2v_alias_slot* tab = a.node_slots + b0;  // first step: node table (:69-70)
    uint32_t base = (uint32_t)b0;
    uint32_t K = (uint32_t)(b1 - b0);
    int32_t len = 1;
    uint32_t t = 0;  // 0-based step counter (philox counter / uniform offset)

    const double* up = nullptr;
    if (RNG == N2V_RNG_UNIFORMS)
        up = a.uniforms + (a.walk_uoff ? a.walk_uoff[lw] : (int64_t)2 * (L - 1) * lw);

    auto step = [&]() -> int32_t {
        if (K == 0) return -1;  // dead end: the walk stops, no draw is consumed (:76-77)
        double u1, u2;
        if (RNG == N2V_RNG_UNIFORMS) {
            const double2 u = *reinterpret_cast<const double2*>(up + 2 * (int64_t)t);
            u1 = u.x;
            u2 = u.y;
        } else {
            n2v::philox_uniforms(a.seed, gw, t, u1, u2);
        }
        ++t;
        const uint32_t kk = (uint32_t)(u1 * (double)K);  // int(floor(rand()*K)), :277
        const n2v_alias_slot s = tab[kk];
        const uint32_t pick = (u2 < s.q) ? kk : (uint32_t)s.J;  // :278-281
        const uint4 r = *reinterpret_cast<const uint4*>(a.recs + (base + pick));
        // r = {slot_lo, base, dst, deg_hi}
        tab = a.slots + (((uint64_t)(r.w >> 24) << 32) | r.x);
        base = r.y;
        K = r.w & 0xFFFFFFu;
        ++len;
        return (int32_t)r.z;
    };

    int32_t* out = a.walks + lw * (int64_t)L;
    if (VEC4) {
        int4 o;
        o.x = cur;
        o.y = step();
        o.z = step();
        o.w = step();
        *reinterpret_cast<int4*>(out) = o;
        for (int32_t g = 4; g < L; g += 4) {
            o.x = step();
            o.y = step();
            o.z = step();
            o.w = step();
            *reinterpret_cast<int4*>(out + g) = o;
        }
    } else {
        if (L > 0) out[0] = cur;
        for (int32_t i = 1; i < L; ++i) out[i] = step();
    }
    a.lens[lw] = (L > 0) ? len : 0;
}

__global__ void __launch_bounds__(256)
edge_recs_kernel(int64_t nnz, const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                 const int64_t* __restrict__ edge_off, int64_t slot_base, n2v_edge_rec* __restrict__ recs) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const int32_t dst = col[e];
    const int64_t b = row_ptr[dst], deg = row_ptr[dst + 1] - b;
    const int64_t off = edge_off ? edge_off[e] : b;
    // a negative offset: this entry's table is not stored (tables under a memory budget, n2v_walk_hybrid rebuilds it)
    const uint64_t slot = off < 0 ? (uint64_t)N2V_NO_TABLE : (uint64_t)(slot_base + off);
    uint4 r;
    r.x = (uint32_t)slot;
    r.y = (uint32_t)b;
    r.z = (uint32_t)dst;
    r.w = (uint32_t)deg | ((uint32_t)(slot >> 32) << 24);
    *reinterpret_cast<uint4*>(recs + e) = r;
}

}  // namespace

extern "C" int n2v_build_edge_recs(int64_t n_nodes, int64_t nnz, const int64_t* row_ptr, const int32_t* col,
                                   const int64_t* edge_off, int64_t slot_base, int64_t max_degree,
                                   int64_t total_slots, n2v_edge_rec* recs, void* stream) {
    if (n_nodes < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col || !recs)))
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_recs: null pointer or negative size");
    if (nnz >= (int64_t)1 << 32) return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_recs: nnz %lld >= 2^32", (long long)nnz);
    if (max_degree >= (int64_t)1 << 24)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_recs: max degree %lld >= 2^24", (long long)max_degree);
    if (total_slots >= (int64_t)N2V_NO_TABLE)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_recs: %lld alias slots >= 2^40", (long long)total_slots);
    if (nnz == 0) return N2V_OK;
    hipLaunchKernelGGL(edge_recs_kernel, dim3(n2v::grid_for(nnz, 256)), dim3(256), 0, (hipStream_t)stream, nnz,
                       row_ptr, col, edge_off, slot_base, recs);
    return n2v::check_launch("n2v_build_edge_recs");
}

extern "C" int n2v_walk(const int64_t* row_ptr, const n2v_alias_slot* node_slots, const n2v_edge_rec* recs,
                        const n2v_alias_slot* slots, const int32_t* starts, int64_t n_starts, int64_t pos_begin,
                        int64_t pos_count, int64_t round_begin, int64_t round_count, int32_t walk_length,
                        int32_t rng_mode, const double* uniforms, const int64_t* walk_uoff, uint64_t seed,
                        int32_t* walks, int32_t* lens, void* stream) {
    if (pos_count < 0 || round_count < 0 || pos_begin < 0 || round_begin < 0 || walk_length < 0 ||
        pos_begin + pos_count > n_starts)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk: bad shard (pos %lld+%lld of %lld, rounds %lld+%lld, L %d)",
                         (long long)pos_begin, (long long)pos_count, (long long)n_starts, (long long)round_begin,
                         (long long)round_count, (int)walk_length);
    const int64_t n_local = pos_count * round_count;
    if (n_local == 0) return N2V_OK;
    if (!row_ptr || !node_slots || !starts || !lens || (walk_length > 0 && !walks))
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk: null pointer");
    if (walk_length > 1 && (!recs || !slots)) return n2v::fail(N2V_ERR_INVALID, "n2v_walk: null tables");
    if (rng_mode != N2V_RNG_UNIFORMS && rng_mode != N2V_RNG_PHILOX)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk: rng_mode %d", (int)rng_mode);
    if (rng_mode == N2V_RNG_UNIFORMS && walk_length > 1 && !uniforms)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk: parity mode needs a uniform buffer");
    if (((uintptr_t)uniforms & 15) != 0) return n2v::fail(N2V_ERR_INVALID, "n2v_walk: uniforms not 16-byte aligned");
    if (n_local > (int64_t)0x7fffffff * 256) return n2v::fail(N2V_ERR_INVALID, "n2v_walk: too many walks in one call");

    WalkArgs a{row_ptr, node_slots, recs, slots, starts, n_starts, pos_begin, pos_count, round_begin, n_local,
               walk_length, uniforms, walk_uoff, seed, walks, lens};
    const bool vec4 = walk_length >= 4 && (walk_length % 4) == 0 && ((uintptr_t)walks & 15) == 0;
    const dim3 grid(n2v::grid_for(n_local, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (rng_mode == N2V_RNG_UNIFORMS) {
        if (vec4) hipLaunchKernelGGL((walk_kernel<N2V_RNG_UNIFORMS, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((walk_kernel<N2V_RNG_UNIFORMS, false>), grid, block, 0, st, a);
    } else {
        if (vec4) hipLaunchKernelGGL((walk_kernel<N2V_RNG_PHILOX, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((walk_kernel<N2V_RNG_PHILOX, false>), grid, block, 0, st, a);
    }
    return n2v::check_launch("n2v_walk");
}
