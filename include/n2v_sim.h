/* n2v_sim.h — C-ABI of the all-pairs similarity + selection kernels (gfx950), SURVEY.md 8(f-1) and 8(f-3).
 *
 * Replaces, in the reference (paths relative to its root):
 *   src/main_link.py:62-170   precision_at_k / make_links_and_score / links_score / link_prediction:
 *                             score every (user, item) — or every unordered node — pair that is not a
 *                             training edge, keep the k best for k in {1,10,50,100,500,1000}
 *   src/main_link.py:351-453  js / get_similarity / build_user_sim_matrx / get_add_edge_by_*:
 *                             N_user x N_user similarity ("cos", "pearson", "jsd") and a per-user
 *                             threshold or top-int(N*ratio) selection
 * Both are O(N^2 d) Python loops over gensim's `similarity` there.  Here a 64x64-tile kernel forms the
 * scores of a row block against all columns (fp32 FMA over rows prepared so that the similarity is a dot
 * product; the Jensen-Shannon form evaluates the reference's rel_entr sum per element) and either
 *   (a) streams candidates above a running threshold into a small buffer (global top-k; nothing of size
 *       N^2 is stored, training edges are dropped by a binary search of their sorted keys — only for the
 *       few candidates that pass the threshold), or
 *   (b) writes the row block [rows x n_cols] once, from which one workgroup per row selects by threshold
 *       (count + ordered fill) or by exact radix select of the k-th largest score (ordered fill), in list
 *       order — the order the reference's per-user loops emit.
 * Conventions as in n2v_hip.h: device pointers, caller-owned memory, asynchronous on `stream`,
 * int return codes + n2v_last_error().
 */
#ifndef N2V_SIM_H
#define N2V_SIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N2V_SIM_COS 0      /* emb.similarity: dot of unit vectors            (src/main_link.py:359-360) */
#define N2V_SIM_PEARSON 1  /* scipy pearsonr(x, y)[0]: dot of centred units   (src/main_link.py:361-362) */
#define N2V_SIM_JSD 2      /* js(p, q): (KL(p|m) + KL(q|m)) / 2, p = x/sum(x) (src/main_link.py:351-356,363-364) */

/* Rows -> the form the tile kernel multiplies.  vec: fp32[n_src][stride]; rows (may be NULL = 0..n_rows-1):
 * int64[n_rows] gather index into vec; out: fp32[n_rows][dpad], dpad a multiple of 32 >= dim, padding 0.
 *   COS     x / sqrt(sum x^2)          (gensim matutils.unitvec)
 *   PEARSON (x - mean) / |x - mean|    (scipy.stats.pearsonr's own normalisation)
 *   JSD     x / sum(x)                 (the reference's p_norm; negative entries are kept — they make the
 *                                       reference's rel_entr infinite and so they do here)                  */
int n2v_sim_prepare(const float* vec, int32_t stride, int32_t dim, const int64_t* rows, int64_t n_rows,
                    int32_t method, float* out, int32_t dpad, void* stream);

/* Scores of rows [row_begin, row_begin + n_rows) of A against all n_cols rows of B (both prepared, row
 * length dpad): out[(r - row_begin) * ld + c].  method: N2V_SIM_JSD evaluates the rel_entr sum, anything
 * else the dot product.  zero_diag_off >= 0: the score of (r, c == r + zero_diag_off) is set to 0
 * (`user_user_sim_list[i] = 0`, src/main_link.py:386,404,421,438,451); < 0: off.                          */
int n2v_sim_block(const float* A, int64_t row_begin, int64_t n_rows, const float* B, int64_t n_cols,
                  int32_t dpad, int32_t method, int64_t zero_diag_off, float* out, int64_t ld, void* stream);

/* Global top-k scan (src/main_link.py:69-105): every score of rows [row_begin, row_end) x [0, n_cols) that
 * is > *tau (device float), whose pair is not in excl_keys (sorted int64 row * n_cols + col; NULL/0 = none)
 * and — if upper_triangle — has col > row (`for j in range(i+1, len(nodes))`, :72), is appended to the
 * candidate arrays at an index taken from *counter (int64, device; keeps counting past `capacity`, entries
 * beyond it are dropped — the caller raises tau and rescans).                                              */
int n2v_sim_topk_scan(const float* A, int64_t row_begin, int64_t row_end, const float* B, int64_t n_cols,
                      int32_t dpad, int32_t method, int32_t upper_triangle, const float* tau,
                      const int64_t* excl_keys, int64_t n_excl, float* cand_score, int32_t* cand_row,
                      int32_t* cand_col, int64_t capacity, int64_t* counter, void* stream);

/* Per-row selection from a score block (scores: fp32[n_rows][ld], n_cols valid columns), one workgroup per
 * row, results in COLUMN order inside a row:
 *   n2v_sim_rows_count: counts[r] = #{c : score > thre}                       (:396-424)
 *   n2v_sim_rows_fill : cols/vals at out_off[r] .. (out_off: int64[n_rows], exclusive prefix of counts)
 *   n2v_sim_rows_topk : the k largest of every row — all scores above the k-th largest value plus the first
 *                       ties of it in column order, i.e. exactly sorted(..., key=-score)[:k] as a SET
 *                       (:379-394,426-440); cols/vals: [n_rows][k].  NaN ranks lowest.  k <= n_cols.       */
int n2v_sim_rows_count(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, float thre,
                       int64_t* counts, void* stream);
int n2v_sim_rows_fill(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, float thre,
                      const int64_t* out_off, int32_t* cols, float* vals, void* stream);
int n2v_sim_rows_topk(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k,
                      int32_t* cols, float* vals, void* stream);

#ifdef __cplusplus
}
#endif
#endif
