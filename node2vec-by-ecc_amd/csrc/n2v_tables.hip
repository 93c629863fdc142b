// Edge alias tables, one WAVEFRONT per table — gfx950 (MI355X).
//
// Replaces get_alias_edge + alias_setup for every edge (src/node2vec.py:133-152, 240-269, called from
// preprocess_transition_probs :193-199) like edge_tables_kernel of n2v_alias.hip, with the same bits, but
//   * the table of (src -> dst) is built by the 64 lanes of one wave: the neighbour classification
//     (:142-148), the normalisation (:150, :253) and the initial fill of Vose's two stacks (:252-257) run in
//     parallel with coalesced row reads; the two inherently serial pieces keep the reference's order — the
//     left-to-right fp64 sum (:149) and the stack pairing (:259-268) — but are fed from registers: 64 entries
//     of a stack are loaded by the 64 lanes at once and consumed through v_readlane, and an element pushed back
//     is always the next one popped from its stack, so it never touches memory (n2v_vose.h has the argument);
//   * tables of up to 512 slots live in the wave's slice of LDS while they are built; of larger ones (hubs) only the
//     two stacks are stored, in a per-wave scratch, and finished slots are written straight to the output, 64 at a
//     time (n2v_wave_table.h, wave_build_stream) — building them in place in the output cost 2.65x the output in
//     write traffic (profiles/r02);
//   * the result is written once, coalesced, in the layout the walk reads: 16-B thin slots {q, J} or directly
//     the 32-B "fat" slots {q, record of neighbour k, record of neighbour J[k]} — no thin copy has to exist
//     next to the fat tables any more.
// One lane per table (n2v_alias.hip) touches every 16-B slot ~8 times with uncoalesced accesses and sits on the
// chip's random-request rate (tools/lab: ~5e10 64-B requests/s): 0.33 s for the 1.83e9 slots of C3.
// Compile with -ffp-contract=off (separately rounded divide / multiply / adds as in the reference).
#include "n2v_common.h"
#include "n2v_wave_table.h"

namespace {

using n2v::uni;
using n2v::uni64;

#ifndef N2V_LDS_SLOTS
#define N2V_LDS_SLOTS 512
#endif
constexpr int kLdsSlots = N2V_LDS_SLOTS;   // 8 KiB per wave (+ 0.5 KiB feed + 1.5 KiB source row): 40 KiB per 4-wave workgroup -> 4 workgroups per CU

struct TabArgs {
    n2v::RowCtx g;
    const int32_t* src_of;
    const int64_t* edge_off;
    const int32_t* order;
    int64_t e_begin, e_end;
    const n2v_edge_rec* recs;    // fat output only
    n2v_alias_slot* thin;
    n2v_fat_slot* fat;
    int32_t* status;
    unsigned long long* work;    // dynamic hand-out of tables (NULL: static grid-stride)
    unsigned char* scratch;      // per-wave stacks of the large tables: [n_waves][max_k] x (int32 + double)
    int64_t max_k;               // scratch entries per wave
};

constexpr int kChunk = 16;       // tables a wave takes per visit to the shared counter

// fat[t0 + k] = {q[k], rec(base + k), rec(base + J[k])} (include/n2v_hip.h, n2v_fat_slot)
template <typename Slot>
__device__ __forceinline__ void emit_fat(const TabArgs& a, const Slot* T, int64_t t0, int64_t base, int K, int lane) {
    for (int k = lane; k < K; k += 64) {
        const double q = T[k].q;
        const int J = T[k].J;
        const uint4 ra = *reinterpret_cast<const uint4*>(a.recs + base + k);   // {slot_lo, base, dst, deg_hi}
        const uint4 rb = *reinterpret_cast<const uint4*>(a.recs + base + J);
        uint4 lo, hi;
        lo.x = (uint32_t)__double2loint(q); lo.y = (uint32_t)__double2hiint(q);
        lo.z = ra.x; lo.w = ra.w;
        hi.x = ra.z; hi.y = rb.x; hi.z = rb.w; hi.w = rb.z;
        uint4* o = reinterpret_cast<uint4*>(a.fat + t0 + k);
        o[0] = lo;
        o[1] = hi;
    }
}

#ifdef N2V_TAB_STAMPS
// [0..3] phases of wave_build_table (weights, sum, normalise + stacks, pairing), [4] emit, [5] hand-out / table header,
// for tables in LDS; [8..13] the same for tables built in global memory; [6] / [14] slots
__device__ unsigned long long g_tab_stamps[16];
extern "C" int n2v_debug_tab_stamps(unsigned long long* host_out, int reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tab_stamps), sizeof(g_tab_stamps)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tab_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

// one finished slot -> its output slot (the large tables' emitters; tables in LDS are written by emit_fat / the loop)
struct FatEmit {
    const n2v_edge_rec* recs;
    n2v_fat_slot* fat;       // + t0
    __device__ __forceinline__ void operator()(bool active, int idx, double q, int J) const {
        if (!active) return;
        const uint4 ra = *reinterpret_cast<const uint4*>(recs + idx);   // recs + base
        const uint4 rb = *reinterpret_cast<const uint4*>(recs + J);
        uint4 lo, hi;
        lo.x = (uint32_t)__double2loint(q); lo.y = (uint32_t)__double2hiint(q);
        lo.z = ra.x; lo.w = ra.w;
        hi.x = ra.z; hi.y = rb.x; hi.z = rb.w; hi.w = rb.z;
        uint4* o = reinterpret_cast<uint4*>(fat + idx);
        o[0] = lo;
        o[1] = hi;
    }
};
struct ThinEmit {
    n2v_alias_slot* thin;    // + t0
    __device__ __forceinline__ void operator()(bool active, int idx, double q, int J) const {
        if (!active) return;
        n2v_alias_slot s;
        s.q = q; s.J = J; s.aux = 0;
        thin[idx] = s;
    }
};

template <bool FAT>
__global__ void __launch_bounds__(256) edge_tables_wave_kernel(TabArgs a) {
    __shared__ n2v_alias_slot lds[4 * kLdsSlots];
    __shared__ double feed[4 * n2v::kFeed];
    __shared__ int32_t rows[4 * n2v::kRowCache];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    n2v_alias_slot* Tl = lds + wv * kLdsSlots;
    int32_t* my_row = rows + wv * n2v::kRowCache;
    n2v::WaveScratch ws{feed + wv * n2v::kFeed, my_row, -1};
    int32_t cached_src = -1;               // consecutive CSR entries share their source: its row is staged once
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    bool zero = false;
#ifdef N2V_TAB_STAMPS
    unsigned long long st_l[8] = {0}, st_g[8] = {0};
    unsigned long long t_last_ = __builtin_amdgcn_s_memtime();
#define N2V_KSTAMP(arr, i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); arr[i] += now_ - t_last_; t_last_ = now_; } while (0)
    unsigned long long* stl = st_l; unsigned long long* stg = st_g;
#else
#define N2V_KSTAMP(arr, i) do { } while (0)
    unsigned long long* stl = nullptr; unsigned long long* stg = nullptr;
#endif
    // Tables are taken kChunk at a time: from a shared counter (table sizes span 10 ... 16 614 slots on C3, so a static
    // assignment leaves the waves with the hubs as the tail) or, without a counter, by static striding.  The next
    // chunk's counter value is requested before this chunk is worked on, and the headers of a chunk's tables (source,
    // destination, its row, the table's place: two levels of dependent loads) are fetched by 16 lanes at once —
    // per table they cost a quarter of a small table's time (profiles/r03/logs/tab_stamps_*.log).
    unsigned long long nxt = 0, stat = ((unsigned long long)blockIdx.x * 4 + wv) * kChunk;
    auto request_chunk = [&]() {
        if (a.work) { if (lane == 0) nxt = atomicAdd(a.work, (unsigned long long)kChunk); }
        else { nxt = stat; stat += (unsigned long long)n_waves * kChunk; }
    };
    request_chunk();
    for (;;) {
        const int64_t i0 = a.e_begin + uni64((int64_t)nxt);
        if (i0 >= a.e_end) break;
        request_chunk();
        const int n_in = (int)min((int64_t)kChunk, a.e_end - i0);
        int32_t h_src = 0, h_dst = 0, h_K = 0;
        int64_t h_base = 0, h_t0 = 0;
        if (lane < n_in) {
            const int64_t e = a.order ? (int64_t)(uint32_t)a.order[i0 + lane] : i0 + lane;
            h_src = a.src_of[e];
            h_dst = a.g.col[e];
            h_base = a.g.row_ptr[h_dst];
            h_K = (int32_t)(a.g.row_ptr[h_dst + 1] - h_base);
            h_t0 = a.edge_off[e];
        }
      for (int j = 0; j < n_in; ++j) {
        const int32_t src = __builtin_amdgcn_readlane(h_src, j);
        const int K = __builtin_amdgcn_readlane(h_K, j);
        if (K == 0) continue;
        const int64_t base = ((int64_t)__builtin_amdgcn_readlane((int)(h_base >> 32), j) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)h_base, j);
        const int64_t t0 = ((int64_t)__builtin_amdgcn_readlane((int)(h_t0 >> 32), j) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)h_t0, j);
        if (src != cached_src) {
            ws.row_n = n2v::wave_cache_row(a.g, my_row, src, lane);
            cached_src = src;
        }
        if (K <= kLdsSlots) {
            N2V_KSTAMP(st_l, 5);
            if (!n2v::wave_build_table(a.g, Tl, ws, src, base, K, lane, stl)) { zero = true; continue; }
#ifdef N2V_TAB_STAMPS
            t_last_ = __builtin_amdgcn_s_memtime(); st_l[6] += K;
#endif
            if (FAT) emit_fat(a, Tl, t0, base, K, lane);
            else for (int k = lane; k < K; k += 64) { n2v_alias_slot s = Tl[k]; s.aux = 0; a.thin[t0 + k] = s; }
            __builtin_amdgcn_wave_barrier();   // the LDS slice is reused by the next table
            N2V_KSTAMP(st_l, 4);
        } else {
            N2V_KSTAMP(st_g, 5);
            // stacks in this wave's scratch row; the output queue overlays the (unused) LDS table slice
            unsigned char* row = a.scratch + ((int64_t)blockIdx.x * 4 + wv) * a.max_k * 12;
            const n2v::StreamStacks S{reinterpret_cast<int32_t*>(row + a.max_k * 8), reinterpret_cast<double*>(row)};
            int32_t* qi = reinterpret_cast<int32_t*>(Tl);
            bool ok;
            if (FAT) {
                n2v::QueueSink<FatEmit> sink{qi, qi + 128, reinterpret_cast<double*>(qi + 256), FatEmit{a.recs + base, a.fat + t0}, lane};
                ok = n2v::wave_build_stream(a.g, S, ws, sink, src, base, K, lane, stg);
            } else {
                n2v::QueueSink<ThinEmit> sink{qi, qi + 128, reinterpret_cast<double*>(qi + 256), ThinEmit{a.thin + t0}, lane};
                ok = n2v::wave_build_stream(a.g, S, ws, sink, src, base, K, lane, stg);
            }
            if (!ok) { zero = true; continue; }
#ifdef N2V_TAB_STAMPS
            st_g[6] += K;
#endif
            __builtin_amdgcn_wave_barrier();   // the LDS slice is reused by the next table
        }
      }
    }
    if (zero && lane == 0) atomicOr(a.status, N2V_STATUS_ZERO_NORM);
#ifdef N2V_TAB_STAMPS
    if (lane == 0)
        for (int i = 0; i < 7; ++i) { atomicAdd(&g_tab_stamps[i], st_l[i]); atomicAdd(&g_tab_stamps[8 + i], st_g[i]); }
#endif
}

}  // namespace

namespace {
constexpr int kLdsPerWg = 4 * (kLdsSlots * 16 + n2v::kFeed * 8 + n2v::kRowCache * 4);
constexpr int kWgPerCu = (160 * 1024 / kLdsPerWg) < 8 ? (160 * 1024 / kLdsPerWg) : 8;
constexpr int64_t kMaxBlocks = 256 * kWgPerCu;     // every resident workgroup slot of the chip, once: tables are handed out
static_assert(kLdsSlots * 16 >= 2048, "the output queue of the large tables overlays the LDS table slice");
inline int64_t scratch_k(int64_t max_degree) { return max_degree <= kLdsSlots ? 0 : (max_degree + 15) / 16 * 16; }
}  // namespace

extern "C" int64_t n2v_edge_tables_wave_scratch_bytes(int64_t max_degree) {
    return max_degree < 0 ? -1 : kMaxBlocks * 4 * scratch_k(max_degree) * 12;
}

extern "C" int n2v_build_edge_tables_wave(int64_t n_nodes, const int64_t* row_ptr, const int32_t* col, const double* w,
                                          const int32_t* src_of, double p, double q, int32_t symmetric,
                                          const int64_t* edge_off, const int32_t* order, int64_t e_begin, int64_t e_end,
                                          const n2v_edge_rec* recs, n2v_alias_slot* thin, n2v_fat_slot* fat,
                                          int32_t* status, uint64_t* work_counter, int64_t max_degree, void* scratch,
                                          int64_t scratch_bytes, void* stream) {
    if (n_nodes < 0 || e_begin < 0 || e_end < e_begin || max_degree < 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables_wave: bad range [%lld, %lld)", (long long)e_begin, (long long)e_end);
    if (e_end == e_begin) return N2V_OK;
    if (!row_ptr || !col || !src_of || !edge_off || !status)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables_wave: null pointer");
    if ((thin == nullptr) == (fat == nullptr))
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables_wave: exactly one of thin / fat output");
    if (fat && !recs) return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables_wave: fat output needs the edge records");
    if (fat && ((uintptr_t)fat & 31) != 0) return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables_wave: fat slots not 32-byte aligned");
    if (!(p == p) || !(q == q)) return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables_wave: p or q is NaN");
    const int64_t need = n2v_edge_tables_wave_scratch_bytes(max_degree);
    if (need > 0 && (!scratch || scratch_bytes < need || ((uintptr_t)scratch & 63) != 0))
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables_wave: max degree %lld needs %lld bytes of 64-byte aligned scratch "
                         "(n2v_edge_tables_wave_scratch_bytes), got %lld", (long long)max_degree, (long long)need, (long long)scratch_bytes);
    TabArgs a{n2v::RowCtx{row_ptr, col, w, p, q, symmetric}, src_of, edge_off, order, e_begin, e_end, recs, thin, fat, status,
              reinterpret_cast<unsigned long long*>(work_counter), reinterpret_cast<unsigned char*>(scratch), scratch_k(max_degree)};
    int64_t blocks = (e_end - e_begin + 4 * kChunk - 1) / (4 * kChunk);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    if (fat) hipLaunchKernelGGL((edge_tables_wave_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((edge_tables_wave_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return n2v::check_launch("n2v_build_edge_tables_wave");
}
