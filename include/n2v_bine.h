/* n2v_bine.h — C-ABI of the BiNE path (BASELINE config 5) on MI355X (gfx950).
 *
 * Replaces, in the reference (chan0park/node2vec-by-ecc), the bipartite embedding pipeline of
 * src/bine_train.py / src/bine_graph_utils.py / src/bine_graph.py / src/bine_lsh.py:
 * HITS centrality -> restart walks on the two implicit projections A*A^T, A^T*A -> window
 * contexts + pooled negatives -> per-iteration pass over the rating list (skip-gram blocks for
 * first-seen vertices + the KL term per edge) with the loss-driven learning rate.
 * Same conventions as n2v_hip.h (device pointers, caller-owned memory, async on `stream`,
 * int return + n2v_last_error()).  The Python binding is n2v_hip/_lib.py; the reference-side
 * stub is in INTEGRATION.md.
 *
 * Graph layout: ONE symmetric CSR over the combined vertex set — users 0..n_u-1 (ascending
 * label, as `node_u.sort()`, src/bine_graph_utils.py:53), items n_u..n_u+n_v-1 — rows sorted
 * ascending, fp64 ratings.  A user's row holds item ids and vice versa, so "the vertices two
 * hops away" are the projection's neighbours without A*A^T ever being materialised
 * (the reference materialises it: src/bine_graph_utils.py:114-123).
 *
 * Randomness: the reference draws from an unseeded `random.Random()` default argument
 * (src/bine_graph.py:169,224,334) and the global `random` — not reproducible even by itself.
 * Every draw here is Philox4x32-10 keyed by a caller seed and counted by the item it decides
 * (walk & step, vertex & pool slot, occurrence, vertex & iteration), so results are a pure
 * function of (inputs, seeds) at any launch geometry, and oracle/bine_oracle.py restates the
 * same draws bit for bit.
 */
#ifndef N2V_BINE_H
#define N2V_BINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- calculate_centrality (src/bine_graph_utils.py:60-86 -> networkx 1.11 hits()) ------------ */

/* y[r] = sum_e w[e] * x[col[e]] over row r of the CSR (one wavefront per row).  Two of these
 * are one HITS iteration: a = M h_last, h = M a (M symmetric here).                         */
int n2v_bine_spmv(int64_t n_rows, const int64_t* row_ptr, const int32_t* col, const double* w,
                  const double* x, double* y, void* stream);

/* The rest of one networkx-1.11 hits() iteration: h *= 1/max(h); a *= 1/max(a);
 * state[0] = sum_i |h[i] - h_last[i]| (fixed summation order: deterministic).               */
int n2v_bine_hits_normalise(int64_t n, double* h, double* a, const double* h_last, double* state,
                            void* stream);

/* Per-side min-max scaling of the authority scores and the walk count per vertex:
 * counts[v] = max(ceil(maxT * (a[v]-min)/(max-min)), minT) for v in [lo, hi) (all zero scores
 * when max == min), src/bine_graph_utils.py:62-86 + src/bine_graph.py:338,358.
 * auth_out (may be NULL) receives the scaled scores.                                         */
int n2v_bine_walk_counts(const double* a, int64_t lo, int64_t hi, int32_t maxT, int32_t minT,
                         int32_t* counts, double* auth_out, void* stream);

/* ---- restart walks on a projection (src/bine_graph.py:263-308, 346-363) ------------------------
 * Walk gw (global index within its side) starts at walk_node[gw].  Before every step a stop
 * draw u is taken; the walk goes on while u > percentage (src/bine_graph.py:276).  A step moves
 * to a vertex drawn UNIFORMLY from the DISTINCT vertices two hops from cur, cur excluded
 * (`rand.choice(matrix[cur])` redrawn while == cur, :300-303; matrix rows are the de-duplicated
 * non-zeros of A*A^T).  Without the matrix: propose a two-hop path cur - mid - w uniformly (cum2 =
 * exclusive prefix sum of deg(col[e]), int64[nnz+1]), reject it if w == cur, and keep it only if mid
 * is the smallest common neighbour of cur and w — exactly one of the |N(cur) & N(w)| paths that
 * reach w survives, so the kept w is uniform over the distinct two-hop vertices.
 * A vertex with no other vertex two hops away ends the walk (:298,306-307).
 * n2v_bine_walk_lengths: planned token count per walk (1 + leading stop draws > percentage,
 * capped at max_len; 1 for a dead-end start), so the caller can prefix-sum ragged offsets.
 * n2v_bine_walk: fills tokens[walk_off[i] ...] (walk_off: int64[n_walks+1] from the lengths)
 * for local walks i = 0..n_walks-1 with global index gw_base + i.  One wavefront per walk.     */
int n2v_bine_walk_lengths(const int64_t* row_ptr, const int64_t* cum2, const int32_t* walk_node,
                          int64_t n_walks, int64_t gw_base, double percentage, int32_t max_len,
                          uint64_t seed, int32_t* lens, void* stream);
int n2v_bine_walk(const int64_t* row_ptr, const int32_t* col, const int64_t* cum2,
                  const int32_t* walk_node, const int64_t* walk_off, int64_t n_walks, int64_t gw_base,
                  uint64_t seed, int32_t* tokens, void* stream);

/* ---- negative pools (src/bine_lsh.py:22-51; datasketch MinHash LSH forest, absent offline) ----
 * pool[v][s], s < pool_size (reference sample_num = 200): a vertex of v's own side
 * [side_lo, side_hi), drawn uniformly, redrawn (up to 16 times) while it is v itself or its
 * Jaccard similarity with v (over their rows) exceeds max_jaccard — the stand-in for "not
 * returned by the LSH forest query" (DESIGN.md 4.7).  Rows v in [v_begin, v_end) are filled;
 * pool is int32[(v_end-v_begin)][pool_size] (row 0 = v_begin).                              */
int n2v_bine_neg_pools(const int64_t* row_ptr, const int32_t* col, int64_t side_lo, int64_t side_hi,
                       int64_t v_begin, int64_t v_end, int32_t pool_size, double max_jaccard,
                       uint64_t seed, int32_t* pool, void* stream);

/* ---- init_embedding_vectors (src/bine_train.py:183-206) ---------------------------------------
 * emb/ctx: fp64 [n][row_stride] (row_stride a multiple of 64 >= dim; padding stays 0).  Every
 * row ~ U[0,1)^dim scaled to unit l2 norm (sklearn normalize), Philox keyed by (seed; row, table). */
int n2v_bine_init(double* emb, double* ctx, int64_t n, int32_t dim, int32_t row_stride, uint64_t seed,
                  void* stream);

/* ---- one training iteration (src/bine_train.py:454-504; skip_gram :243-274; KL_divergence :277-309)
 * For every rating e = (edge_u[e], edge_v[e], edge_w[e]) in list order: if first[e] & 1 the
 * skip-gram block of the user, if first[e] & 2 that of the item (the reference's visited_u /
 * visited_v dictionaries: a vertex is handled at its first rating, :462,475), then the KL update.
 * Skip-gram block of vertex c: min(#occurrences, 10) distinct occurrences of c in its side's
 * walks (random.sample, :465); per occurrence the window contexts z != c (window ws, within the
 * walk) and up to ns negatives — distinct slots of pool[c], dropped when inside the window or
 * repeated (src/bine_graph_utils.py:163-187) — then skip_gram(c, z, negs) for every z.
 * Occurrence index: occ_ptr int64[n+1], occ_pos int64[n_tokens] (token positions of each vertex,
 * ascending), tokens int32[n_tokens], tok_walk int32[n_tokens] (walk of each token), walk_off
 * int64[n_walks+1]; both sides in one token array (walks of users, then walks of items).
 * state: double[8] = {lam, loss, last_loss, stop, rows, rows_ref, work counter (uint64 bits), -}; the pass adds its loss
 * to state[1] and the number of embedding rows it read + wrote (algorithmic traffic = rows * dim * 8 B) to
 * state[4], and to state[5] the rows the reference's access pattern moves for the same work (every
 * skip_gram call reads and writes its context row and all target rows; a KL update 4 rows);
 * state[6] hands out chunks of ratings to the wavefronts and must be 0 on entry
 * (n2v_bine_lambda_step resets it).
 * mode N2V_BINE_SEQUENTIAL: one wavefront walks the list in order with plain loads/stores — the
 * reference's exact update order (used for parity tests, small inputs).  N2V_BINE_PARALLEL:
 * wavefronts take ratings e, e+W, ...; rows are read at agent scope and updated with fp64 atomic
 * adds (no update lost; Hogwild ordering).  N2V_BINE_PARALLEL_STORE: the same, except that the context rows
 * of an occurrence (its centre and negatives) are written back whole with agent-scope stores — the fp64
 * atomic rate is what bounds the pass, and these rows are shared only when a vertex happens to be another
 * wavefront's negative at that moment (that racing update is then lost).  Ratings [e_begin, e_end) are
 * processed (a rank's shard); `first` is indexed by the global e.                                               */
#define N2V_BINE_SEQUENTIAL 0
#define N2V_BINE_PARALLEL 1
#define N2V_BINE_PARALLEL_STORE 2
int n2v_bine_train_pass(const int32_t* edge_u, const int32_t* edge_v, const double* edge_w,
                        const uint8_t* first, int64_t e_begin, int64_t e_end, double* emb, double* ctx,
                        int32_t dim, int32_t row_stride, const int64_t* occ_ptr, const int64_t* occ_pos,
                        const int32_t* tokens, const int32_t* tok_walk, const int64_t* walk_off,
                        const int32_t* pool, int32_t pool_size, int32_t ws, int32_t ns, double alpha,
                        double beta, double gamma, double* state, int32_t iteration, uint64_t seed_occ,
                        uint64_t seed_neg, int32_t mode, int32_t max_blocks, void* stream);

/* End of an iteration (src/bine_train.py:495-502): lam *= 1.05 if last_loss > loss else 0.95;
 * stop = |loss - last_loss| < epsilon; last_loss = loss; loss = 0; work counter = 0.         */
int n2v_bine_lambda_step(double* state, double epsilon, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* N2V_BINE_H */
