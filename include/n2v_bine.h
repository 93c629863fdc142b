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

/* ---- negative pools, exact-Jaccard variant (role of src/bine_lsh.py:22-51 without the forest) ----
 * pool[v][s], s < pool_size (reference sample_num = 200): a vertex of v's own side
 * [side_lo, side_hi), drawn uniformly, redrawn (up to 16 times) while it is v itself or its
 * Jaccard similarity with v (over their rows) exceeds max_jaccard — the stand-in for "not
 * returned by the LSH forest query" (DESIGN.md 4.7).  Rows v in [v_begin, v_end) are filled;
 * pool is int32[(v_end-v_begin)][pool_size] (row 0 = v_begin).                              */
int n2v_bine_neg_pools(const int64_t* row_ptr, const int32_t* col, int64_t side_lo, int64_t side_hi,
                       int64_t v_begin, int64_t v_end, int32_t pool_size, double max_jaccard,
                       uint64_t seed, int32_t* pool, void* stream);

/* ---- negative pools, MinHash LSH forest (src/bine_lsh.py:7-51 on datasketch 1.2.5, requirements.txt:11) ----
 * The reference builds MinHash(num_perm=128) of every vertex's neighbour labels, adds them to a
 * MinHashLSHForest(num_perm=128) (l = 8 trees of k = 16 values), and for every not-yet-visited vertex i takes
 * sim = forest.query(ms[i], 200); every unvisited vertex of sim joins i's cluster; the cluster's pool is a
 * random.sample of 200 vertices from the side minus sim(i) and minus sim(j) of every j in sim(i)
 * (src/bine_lsh.py:27-51).  The five steps below are that pipeline; ids are LOCAL to one side (0..n-1) except where
 * said.  datasketch is absent offline: MinHash/LSHForest follow its published 1.2.5 source.
 *
 * n2v_lsh_sha1_labels: hv[i] = first four bytes (little endian) of SHA-1(label i) — datasketch MinHash.update's hash of
 *   `d.encode('utf8')` (src/bine_lsh.py:16).  bytes: uint8[n][width] (label i in its first lens[i] bytes).
 * n2v_lsh_minhash: sig[v - v_begin][j] = min over neighbours c of ((perm_a[j]*hv[c] + perm_b[j]) mod 2^64 mod (2^61-1))
 *   & (2^32-1), 2^32-1 for an empty row; perm_a/perm_b: uint64[128] = datasketch's RandomState(1) parameters
 *   (src/bine_lsh.py:14-16).  hv is indexed by the GLOBAL ids col holds.
 * n2v_lsh_forest_query: forest.query(ms[v], k) for every v of the side (src/bine_lsh.py:38,46).  order: int32[8][n], tree
 *   t's vertices sorted by their 16 values (ties by id = insertion order, src/bine_lsh.py:18); lo/hi: int32[n][8][16],
 *   [v][t][r-1] = range of sorted positions of tree t sharing v's first r values.  sim: int32[n][k] receives the keys in
 *   the order the forest yields new ones (r = 16..1, trees 0..7), sim_n their number (<= k; k <= 256).
 * n2v_lsh_leader_round: one round of the `visted` sweep (src/bine_lsh.py:32-36,41-45).  rev_ptr/rev_src: CSR of
 *   {l < i : i in sim(l)}, ascending l; owner: int32[n], -1 = unresolved on entry of the first round; a resolved
 *   vertex holds the vertex whose turn produced its pool (itself = it has a turn).  unresolved (int32, device) is
 *   incremented once per vertex still open; call until it stays 0.
 * n2v_lsh_pools: pool rows of the vertices lead[0..n_lead) (owners of themselves): pool_size distinct vertices of the
 *   side outside the exclusion set, drawn by Philox(seed; leader, round, lane) rejection in lane order, stored as
 *   id_base + local id; when at most pool_size vertices are left, all of them ascending, then -1
 *   (random.sample(total_list, min(sample_num, len(total_list))), src/bine_lsh.py:48).  pool: int32[n][pool_size], only
 *   leaders' rows are written (followers copy their owner's row).  bitmap: uint32[n_groups][words_per_group], zero on
 *   entry and on return, words_per_group >= ceil(n_side / 32).                                                   */
#define N2V_LSH_NUM_PERM 128
#define N2V_LSH_TREES 8
int n2v_lsh_sha1_labels(const uint8_t* bytes, int32_t width, const int32_t* lens, int64_t n, uint32_t* hv,
                        void* stream);
int n2v_lsh_minhash(const int64_t* row_ptr, const int32_t* col, const uint32_t* hv, const uint64_t* perm_a,
                    const uint64_t* perm_b, int64_t v_begin, int64_t v_end, uint32_t* sig, void* stream);
int n2v_lsh_forest_query(const int32_t* order, const int32_t* lo, const int32_t* hi, int64_t n, int32_t k,
                         int32_t* sim, int32_t* sim_n, void* stream);
int n2v_lsh_leader_round(const int64_t* rev_ptr, const int32_t* rev_src, int64_t n, int32_t* owner,
                         int32_t* unresolved, void* stream);
int n2v_lsh_pools(const int32_t* sim, const int32_t* sim_n, int32_t k, const int32_t* lead, int64_t n_lead,
                  int32_t n_side, int32_t pool_size, uint64_t seed, int32_t id_base, uint32_t* bitmap,
                  int64_t words_per_group, int32_t n_groups, int32_t* pool, void* stream);

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
 * walk) and up to ns negatives — distinct slots of pool[c], dropped when empty (-1: a pool shorter than pool_size),
 * inside the window or repeated (src/bine_graph_utils.py:163-187) — then skip_gram(c, z, negs) for every z.
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
