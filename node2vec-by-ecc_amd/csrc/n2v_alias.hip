// Alias-table construction — gfx950 (MI355X) kernels.
//
// Replaces alias_setup / get_alias_edge / preprocess_transition_probs of the reference
// (src/node2vec.py:240-269, :133-152, :176-204).  Results are bit-identical to the
// reference: fp64 throughout, the neighbourhood sum is the plain left-to-right sum of
// Python's sum() (:149,:186), probabilities are divide-then-multiply (:150,:253), and
// Vose's pairing pops both stacks from their most recently pushed end (:259-268).  That
// pairing is serial per table, so one lane builds one table; the caller hands tables to
// lanes in size order (`order`) so the 64 lanes of a wave run tables of similar length.
// The two stacks live in the `aux` word of the table's own slots (smaller grows up from
// slot 0, larger grows down from slot K-1; together they never hold more than K indices),
// so construction needs no scratch memory beyond the output.
//
// Compile with -ffp-contract=off: no fused multiply-add may replace the reference's
// separately rounded operations.
#include "n2v_common.h"
#include "n2v_vose.h"

namespace {

// On entry T[k].q holds the normalised probability of slot k (src/node2vec.py:150/187).
// The pairing itself (register-carried, reference-exact) is in n2v_vose.h.
__device__ __forceinline__ void vose_inplace(n2v_alias_slot* __restrict__ T, int64_t K) {
    n2v::vose_pair<n2v::kProb>(T, K);
}

__global__ void __launch_bounds__(256)
setup_tables_kernel(int64_t n_tables, const int64_t* __restrict__ tab_off, n2v_alias_slot* __restrict__ slots) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_tables) return;
    vose_inplace(slots + tab_off[i], tab_off[i + 1] - tab_off[i]);
}

__global__ void __launch_bounds__(256)
node_tables_kernel(int64_t n_nodes, const int64_t* __restrict__ row_ptr, const double* __restrict__ w,
                   n2v_alias_slot* __restrict__ slots, int32_t* __restrict__ status) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_nodes) return;
    const int64_t b = row_ptr[v], K = row_ptr[v + 1] - b;
    if (K == 0) return;
    n2v_alias_slot* T = slots + b;
    double norm = 0.0;
    for (int64_t k = 0; k < K; ++k) norm = norm + (w ? w[b + k] : 1.0);  // sum(), :186
    if (norm == 0.0) {
        atomicOr(status, N2V_STATUS_ZERO_NORM);
        return;
    }
    for (int64_t k = 0; k < K; ++k) T[k].q = (w ? w[b + k] : 1.0);
    n2v::vose_pair<n2v::kWeight>(T, K, norm);  // prob = u / norm (:187), q = K * prob (:253), pairing
}

__global__ void __launch_bounds__(256)
edge_tables_kernel(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                   const double* __restrict__ w, const int32_t* __restrict__ src_of, double p, double q,
                   int32_t symmetric, const int64_t* __restrict__ edge_off, const int32_t* __restrict__ order, int64_t e_begin,
                   int64_t e_end, n2v_alias_slot* __restrict__ slots, int32_t* __restrict__ status) {
    const int64_t i = e_begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= e_end) return;
    const int64_t e = order ? (int64_t)(uint32_t)order[i] : i;
    const int32_t src = src_of[e], dst = col[e];
    const int64_t b = row_ptr[dst], K = row_ptr[dst + 1] - b;
    if (K == 0) return;
    n2v_alias_slot* T = slots + edge_off[e];
    double norm = 0.0;
    for (int64_t k = 0; k < K; ++k) {
        const int32_t nb = col[b + k];
        const double wt = w ? w[b + k] : 1.0;
        double u;
        if (nb == src) u = wt / p;                               // :143-144
        // G.has_edge(dst_nbr, src), :145-146; on an undirected graph that is "dst_nbr in row(src)":
        // the lane then probes ONE row for all of its slots instead of a different row per slot
        else if (symmetric ? n2v::row_contains(row_ptr, col, src, nb) : n2v::row_contains(row_ptr, col, nb, src)) u = wt;
        else u = wt / q;                                         // :147-148
        T[k].q = u;
        norm = norm + u;  // sum(), :149
    }
    if (norm == 0.0) {
        atomicOr(status, N2V_STATUS_ZERO_NORM);
        return;
    }
    n2v::vose_pair<n2v::kWeight>(T, K, norm);  // prob = u / norm (:150), q = K * prob (:253), pairing
}

}  // namespace

extern "C" int n2v_alias_setup_tables(int64_t n_tables, const int64_t* tab_off, n2v_alias_slot* slots, void* stream) {
    if (n_tables < 0) return n2v::fail(N2V_ERR_INVALID, "n2v_alias_setup_tables: negative count");
    if (n_tables == 0) return N2V_OK;
    if (!tab_off || !slots) return n2v::fail(N2V_ERR_INVALID, "n2v_alias_setup_tables: null pointer");
    hipLaunchKernelGGL(setup_tables_kernel, dim3(n2v::grid_for(n_tables, 256)), dim3(256), 0, (hipStream_t)stream,
                       n_tables, tab_off, slots);
    return n2v::check_launch("n2v_alias_setup_tables");
}

extern "C" int n2v_build_node_tables(int64_t n_nodes, const int64_t* row_ptr, const int32_t* col, const double* w,
                                     n2v_alias_slot* slots, int32_t* status, void* stream) {
    (void)col;
    if (n_nodes < 0 || !row_ptr || !status)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_node_tables: null pointer or negative size");
    if (n_nodes == 0) return N2V_OK;
    if (!slots) return n2v::fail(N2V_ERR_INVALID, "n2v_build_node_tables: null slots");
    hipLaunchKernelGGL(node_tables_kernel, dim3(n2v::grid_for(n_nodes, 256)), dim3(256), 0, (hipStream_t)stream,
                       n_nodes, row_ptr, w, slots, status);
    return n2v::check_launch("n2v_build_node_tables");
}

extern "C" int n2v_build_edge_tables(int64_t n_nodes, const int64_t* row_ptr, const int32_t* col, const double* w,
                                     const int32_t* src_of, double p, double q, int32_t symmetric,
                                     const int64_t* edge_off,
                                     const int32_t* order, int64_t e_begin, int64_t e_end, n2v_alias_slot* slots,
                                     int32_t* status, void* stream) {
    if (n_nodes < 0 || e_begin < 0 || e_end < e_begin)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables: bad range [%lld, %lld)", (long long)e_begin,
                         (long long)e_end);
    if (e_end == e_begin) return N2V_OK;
    if (!row_ptr || !col || !src_of || !edge_off || !slots || !status)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables: null pointer");
    if (!(p == p) || !(q == q)) return n2v::fail(N2V_ERR_INVALID, "n2v_build_edge_tables: p or q is NaN");
    hipLaunchKernelGGL(edge_tables_kernel, dim3(n2v::grid_for(e_end - e_begin, 256)), dim3(256), 0,
                       (hipStream_t)stream, row_ptr, col, w, src_of, p, q, symmetric, edge_off, order, e_begin, e_end,
                       slots, status);
    return n2v::check_launch("n2v_build_edge_tables");
}

extern "C" int n2v_abi_version(void) { return N2V_ABI_VERSION; }
extern "C" const char* n2v_last_error(void) { return n2v::err_buf(); }
