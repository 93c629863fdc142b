/* n2v_hip.h — C-ABI of the MI355X (gfx950) node2vec walk-and-embed engine.
 *
 * The reference (chan0park/node2vec-by-ecc) has no FFI of its own; its seam for this
 * path is the Python surface of src/node2vec.py and src/main.py.  Each entry point
 * below names the reference code it replaces (paths relative to the reference root).
 * The Python binding a maintainer adds is in INTEGRATION.md; the in-tree binding is
 * node2vec-by-ecc_amd/n2v_hip/_lib.py (ctypes).
 *
 * Conventions
 *  - plain `extern "C"`, no C++/torch types; every pointer is a DEVICE pointer unless the
 *    parameter name ends in `_host`; the caller allocates and frees everything, the
 *    library keeps no device memory between calls.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Launches are
 *    asynchronous on that stream; no call synchronises the device unless stated.
 *  - return value: N2V_OK (0) or a negative N2V_ERR_*; n2v_last_error() gives the
 *    message for the calling thread.  Data-dependent failures (zero-sum neighbourhood)
 *    are reported through a device `status` word the caller reads back.
 *  - graph layout: dense node ids 0..N-1 assigned by ascending node label, CSR rows
 *    sorted ascending — so "k-th neighbour in sorted(G.neighbors(v))"
 *    (src/node2vec.py:67,142,185) is col[row_ptr[v] + k].
 */
#ifndef N2V_HIP_H
#define N2V_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N2V_ABI_VERSION 2

#define N2V_OK 0
#define N2V_ERR_INVALID (-1)   /* bad argument (null pointer, negative size, limit exceeded) */
#define N2V_ERR_HIP (-2)       /* HIP runtime error */
#define N2V_STATUS_ZERO_NORM 1 /* device status bit: a neighbourhood's weights sum to 0
                                  (the reference raises ZeroDivisionError, src/node2vec.py:150,187) */

/* One alias slot: (J[k], q[k]) of src/node2vec.py:240-269, 16 B so a draw is one
 * aligned 16-B load.  `aux` is scratch during construction (Vose's two stacks), then 0. */
typedef struct n2v_alias_slot {
    double q;
    int32_t J;
    int32_t aux;
} n2v_alias_slot;

/* One record per CSR entry e = (src -> dst): everything the walk needs to leave `dst`.
 *   slot  = 40-bit index of dst's transition table for arrivals through e
 *           (lo 32 bits in slot_lo, hi 8 bits in the top byte of deg_hi)
 *   base  = row_ptr[dst]           (CSR entry of dst's k-th neighbour is base + k)
 *   dst   = col[e]
 *   deg   = low 24 bits of deg_hi  (out-degree of dst; tables have deg slots)
 * Limits (checked by n2v_build_edge_recs): nnz < 2^32, deg < 2^24, slots < 2^40.        */
typedef struct n2v_edge_rec {
    uint32_t slot_lo;
    uint32_t base;
    uint32_t dst;
    uint32_t deg_hi;
} n2v_edge_rec;

/* ---- queries ------------------------------------------------------------------------ */
int n2v_abi_version(void);
const char* n2v_last_error(void);

/* ---- preprocess_transition_probs (src/node2vec.py:176-204) --------------------------- */

/* alias_setup (src/node2vec.py:240-269) for n_tables independent tables: table i occupies
 * slots[tab_off[i] .. tab_off[i+1]) and on entry slots[].q holds its probabilities (the
 * `probs` argument); on return q/J are the alias table.  tab_off: int64[n_tables+1].     */
int n2v_alias_setup_tables(int64_t n_tables, const int64_t* tab_off, n2v_alias_slot* slots, void* stream);

/* Node tables (src/node2vec.py:184-188): slots[row_ptr[v] + k] for every node v.
 * w == NULL means every weight is 1.  status: int32[1], OR-ed with N2V_STATUS_*.        */
int n2v_build_node_tables(int64_t n_nodes, const int64_t* row_ptr, const int32_t* col,
                          const double* w, n2v_alias_slot* slots, int32_t* status, void* stream);

/* Edge tables (src/node2vec.py:133-152,193-199): for CSR entries e in
 * [e_begin, e_end) — or, if `order` != NULL, for order[i], i in [e_begin, e_end) — the
 * table of (src -> dst=col[e]) is written to slots[edge_off[e] ...], deg(dst) slots.
 * edge_off: int64[nnz+1], exclusive prefix sum of deg(col[e]).  `src_of`: int32[nnz], the
 * row of each CSR entry.  `order` lets the caller bin tables by size (one lane builds one
 * table, so lanes of a wave should get similar sizes).  symmetric != 0 declares the CSR
 * symmetric (undirected graph): G.has_edge(nbr, src) is then looked up in src's own row.   */
int n2v_build_edge_tables(int64_t n_nodes, const int64_t* row_ptr, const int32_t* col,
                          const double* w, const int32_t* src_of, double p, double q,
                          int32_t symmetric, const int64_t* edge_off, const int32_t* order, int64_t e_begin,
                          int64_t e_end, n2v_alias_slot* slots, int32_t* status, void* stream);

/* The same tables (same bits) built by ONE WAVEFRONT per table, with coalesced row reads; tables of up to 512 slots
 * are staged in LDS, of larger ones only the two Vose stacks are kept (in `scratch`) and finished slots go straight to the
 * output; every output slot is written once, in the layout the walk reads: exactly one of `thin` (16-B slots at
 * thin[edge_off[e] ...]) and `fat` (32-B n2v_fat_slot at fat[edge_off[e] ...], needs the walk records `recs` of
 * n2v_build_edge_recs, whose table indices address `fat`) is non-NULL.  With `fat` no thin copy of the edge tables has
 * to exist.  work_counter: uint64[1] set to 0 by the caller — tables are then handed to the wavefronts dynamically
 * (sizes differ by three orders of magnitude on a power-law graph), and `order` is not needed; NULL: static assignment.
 * max_degree: largest out-degree of the graph; scratch: n2v_edge_tables_wave_scratch_bytes(max_degree) bytes, 64-B
 * aligned (0 bytes / NULL when max_degree <= 512).  src/node2vec.py:133-152,240-269.                              */
struct n2v_fat_slot;
int64_t n2v_edge_tables_wave_scratch_bytes(int64_t max_degree);
int n2v_build_edge_tables_wave(int64_t n_nodes, const int64_t* row_ptr, const int32_t* col, const double* w,
                               const int32_t* src_of, double p, double q, int32_t symmetric,
                               const int64_t* edge_off, const int32_t* order, int64_t e_begin, int64_t e_end,
                               const n2v_edge_rec* recs, n2v_alias_slot* thin, struct n2v_fat_slot* fat,
                               int32_t* status, uint64_t* work_counter, int64_t max_degree, void* scratch,
                               int64_t scratch_bytes, void* stream);

/* Walk records.  edge_off == NULL: first-order shortcut (p == q == 1), every record
 * points at dst's node table, slot = slot_base + row_ptr[dst].  Otherwise
 * slot = slot_base + edge_off[e]; edge_off[e] < 0 marks an entry whose table is NOT stored
 * (slot = N2V_NO_TABLE: tables under a memory budget, n2v_walk_hybrid).  Host-side limit
 * checks need `max_degree` and `total_slots` (slot_base + last offset).                  */
#define N2V_NO_TABLE 0xFFFFFFFFFFULL /* 40-bit table index of a record whose table is rebuilt on the fly */
int n2v_build_edge_recs(int64_t n_nodes, int64_t nnz, const int64_t* row_ptr, const int32_t* col,
                        const int64_t* edge_off, int64_t slot_base, int64_t max_degree,
                        int64_t total_slots, n2v_edge_rec* recs, void* stream);

/* ---- simulate_walks / node2vec_walk (src/node2vec.py:55-95, :271-281) ---------------- */

#define N2V_RNG_UNIFORMS 0 /* parity mode: two fp64 uniforms per step read from `uniforms` */
#define N2V_RNG_PHILOX 1   /* throughput mode: Philox4x32-10(seed; global walk index, step) */
#define N2V_RNG_UNIFORMS_TILED 2 /* parity mode, `uniforms` in the layout of n2v_mt19937_fill_tiled (n2v_walk_fat only):
                                    a walk whose LINEAR offset is o = s * 2(L-1) (s = its rank among the walks that draw)
                                    reads step t (1-based) at uniforms[2 * (((s >> 6) * (L-1) + (t-1)) * 64 + (s & 63))], [+1]
                                    — the 64 walks of a wavefront read 1 KiB of consecutive bytes per step.  Only for
                                    calls in which no walk can end early (every offset is a multiple of 2(L-1)).      */

/* Walks for start positions [pos_begin, pos_begin+pos_count) of `starts` (dense ids in
 * list(G.nodes()) order) and rounds [round_begin, round_begin+round_count).
 * Local walk lw = round_local * pos_count + pos_local; global walk index
 * gw = (round_begin + round_local) * n_starts + pos_begin + pos_local
 * (= the reference's position of that walk in the returned list, src/node2vec.py:89-93).
 * node_slots: table of node v at node_slots[row_ptr[v]...]; recs/slots from the builders
 * above (slots = the array the recs' 40-bit indices refer to).
 * N2V_RNG_UNIFORMS: step t (1-based) of local walk lw reads uniforms[o + 2(t-1)], [+1]
 * with o = walk_uoff ? walk_uoff[lw] : 2*(walk_length-1)*lw.
 * Output: walks int32[n_local][walk_length] (dense ids, padded with -1), lens int32[n_local]. */
int n2v_walk(const int64_t* row_ptr, const n2v_alias_slot* node_slots, const n2v_edge_rec* recs,
             const n2v_alias_slot* slots, const int32_t* starts, int64_t n_starts, int64_t pos_begin,
             int64_t pos_count, int64_t round_begin, int64_t round_count, int32_t walk_length,
             int32_t rng_mode, const double* uniforms, const int64_t* walk_uoff, uint64_t seed,
             int32_t* walks, int32_t* lens, void* stream);

/* "Fat" alias slot (32 B, 32-B aligned): q plus the walk records of both outcomes of the draw
 * (keep = neighbour k, alias = neighbour J[k]), each {40-bit table index, 24-bit degree, node
 * id} as in n2v_edge_rec without the row base.  One aligned 32-B gather per walk step instead
 * of two dependent 16-B gathers, at 2x the table memory.                                   */
typedef struct n2v_fat_slot {
    double q;
    uint32_t keep_slot_lo, keep_deg_hi, keep_dst;
    uint32_t alias_slot_lo, alias_deg_hi, alias_dst;
} n2v_fat_slot;

/* Expand thin tables into fat ones.  Table i occupies slots [tab_off[i], tab_off[i+1]) of
 * `thin` and of `fat` and draws from the row of node tab_node[i] (tab_node == NULL: node i —
 * the node tables with tab_off = row_ptr; edge tables: tab_off = edge_off, tab_node = col).
 * `recs` are the thin walk records (their 40-bit indices address `fat` identically).        */
int n2v_build_fat_slots(int64_t n_tables, const int64_t* tab_off, const int32_t* tab_node,
                        const int64_t* row_ptr, const n2v_alias_slot* thin, const n2v_edge_rec* recs,
                        n2v_fat_slot* fat, void* stream);

/* n2v_walk over fat tables: identical output.  node_fat: fat node tables (slot k of node v at
 * row_ptr[v]+k); fat: the array the records' table indices address.  rng_mode may also be
 * N2V_RNG_UNIFORMS_TILED (walk_uoff still holds the LINEAR offsets).
 * uoff_round_stride > 0: every round consumes that many uniforms and walk_uoff has pos_count entries, the offsets
 * inside a round: local walk (round_local, pos_local) reads at walk_uoff[pos_local] + round_local * uoff_round_stride
 * (no per-walk array of round_count * pos_count offsets); 0: walk_uoff[lw] as in n2v_walk.                        */
int n2v_walk_fat(const int64_t* row_ptr, const n2v_fat_slot* node_fat, const n2v_fat_slot* fat,
                 const int32_t* starts, int64_t n_starts, int64_t pos_begin, int64_t pos_count,
                 int64_t round_begin, int64_t round_count, int32_t walk_length, int32_t rng_mode,
                 const double* uniforms, const int64_t* walk_uoff, int64_t uoff_round_stride, uint64_t seed,
                 int32_t* walks, int32_t* lens, void* stream);

/* numpy's legacy MT19937 stream (np.random.rand(), src/node2vec.py:277-278) on the device.
 * n2v_mt19937_jump_host: pure host arithmetic.  key_host: the 624 state words of
 * np.random.get_state(); states_host[k] (k < n_streams, 624 words each) receives the state
 * window advanced by k * stride_words outputs (polynomial jump-ahead), states_host[0] = key.
 * n2v_mt19937_fill: stream k (one wavefront) writes the doubles random_sample() would return
 * for outputs [k*words_per_stream, (k+1)*words_per_stream) of the sequence that starts at
 * position `pos` (0..624) of the caller's state, n_doubles in total, to out[] in order.
 * states: DEVICE copy of states_host; words_per_stream even, n_streams*words_per_stream >=
 * 2*n_doubles.  final_state (device uint32[625], may be NULL): key words + pos after the last
 * draw, i.e. what np.random.get_state() would hold after n_doubles random_sample() calls.   */
int n2v_mt19937_jump_host(const uint32_t* key_host, int64_t stride_words, int32_t n_streams,
                          uint32_t* states_host);
/* The same start states computed on the device, by doubling: with the host-made polynomials
 * x^(stride_words * 2^r) mod x*phi(x), r < n_rounds (n2v_mt19937_jump_polys_host: uint32[n_rounds][19968] on the
 * host — per polynomial the number of set bits, then their positions — copied to the device by the caller), round r derives streams [2^r, 2^(r+1)) from streams [0, 2^r).
 * states: DEVICE uint32[n_streams][624], states[0] = the caller's key on entry.  A thousand streams cost about a
 * millisecond instead of the host's half a millisecond each, which lets n2v_mt19937_fill use the whole chip. */
int n2v_mt19937_jump_polys_host(int64_t stride_words, int32_t n_rounds, uint32_t* polys_host);
int n2v_mt19937_jump_device(uint32_t* states, int32_t n_streams, const uint32_t* polys, int32_t n_rounds,
                            void* stream);
int n2v_mt19937_fill(const uint32_t* states, int32_t n_streams, int32_t pos, int64_t words_per_stream,
                     int64_t n_doubles, double* out, uint32_t* final_state, void* stream);
/* The same doubles, regrouped for the walk kernel's coalesced reads (N2V_RNG_UNIFORMS_TILED): the stream is cut into
 * segments of 2*pairs_per_walk doubles (one walk of pairs_per_walk = L-1 steps each: the two np.random.rand() of
 * src/node2vec.py:277-278 per step); segment s, pair t, component c (double 2*pairs_per_walk*s + 2t + c of the
 * stream) is written to out[2 * (((s >> 6) * pairs_per_walk + t) * 64 + (s & 63)) + c].  out must hold
 * ceil(S / 64) * 64 * 2 * pairs_per_walk doubles for S = ceil(n_doubles / (2*pairs_per_walk)) segments (a last partial
 * group of 64 leaves gaps that are never read).  Same stream, same final_state as n2v_mt19937_fill.             */
int n2v_mt19937_fill_tiled(const uint32_t* states, int32_t n_streams, int32_t pos, int64_t words_per_stream,
                           int64_t n_doubles, int32_t pairs_per_walk, double* out, uint32_t* final_state, void* stream);

/* simulate_walks_on_the_fly / node2vec_walk_on_the_fly (src/node2vec.py:13-53,97-111): the
 * same walk with the (prev, cur) table rebuilt at every step instead of read from the edge
 * tables (for graphs whose sum of deg^2 slots does not fit in HBM).  Output is identical to
 * n2v_walk under the same uniforms.  One wavefront per walk.  A step first asks whether it needs
 * the table at all: a slot that alias_setup classifies as `smaller` keeps q = K * prob (:253-255;
 * only q[large] is rewritten, :264), so when slot int(u1*K) is such a slot and u2 < q (:278) the
 * step only needs the weights' left-to-right sum — which on an unweighted undirected graph with
 * dyadic 1/p, 1/q (w == NULL, symmetric, multiples of 2^-20 up to 2^10) is an exact count of
 * common neighbours; rows of at most 64 neighbours are otherwise paired in registers, larger
 * tables of up to n2v_walk_otf_lds_slots() slots in LDS, the rest in `scratch`
 * (n2v_alias_slot[scratch_slots]; the launch uses floor(scratch_slots / max_degree) wavefronts,
 * at least 4 are required when max_degree exceeds the LDS window; may be NULL otherwise).
 * status: int32[1], N2V_STATUS_ZERO_NORM on a zero-sum neighbourhood.  Other arguments as n2v_walk. */
/* Largest table the on-the-fly / hybrid kernels build in LDS, and the most wavefronts a launch uses (= scratch rows). */
int32_t n2v_walk_otf_lds_slots(void);
int32_t n2v_walk_otf_max_waves(void);
int n2v_walk_on_the_fly(const int64_t* row_ptr, const int32_t* col, const double* w, double p, double q,
                        int32_t symmetric, int64_t max_degree, const int32_t* starts, int64_t n_starts, int64_t pos_begin,
                        int64_t pos_count, int64_t round_begin, int64_t round_count,
                        int32_t walk_length, int32_t rng_mode, const double* uniforms,
                        const int64_t* walk_uoff, uint64_t seed, n2v_alias_slot* scratch,
                        int64_t scratch_slots, int32_t* walks, int32_t* lens, int32_t* status,
                        void* stream);

/* Tables under a memory budget — the middle path between n2v_walk_fat and n2v_walk_on_the_fly (the reference's own
 * answer to sum-of-deg^2 memory is to rebuild EVERY table per step, src/node2vec.py:34-53, src/settings.py:18).  The
 * caller stores fat tables for a subset of the CSR entries (n2v_build_edge_recs with negative offsets for the others,
 * n2v_build_edge_tables_wave over an `order` list of the stored ones); a step that arrives through a stored entry is
 * one gather as in n2v_walk_fat, any other step rebuilds its table as n2v_walk_on_the_fly does.  Same walks, bit for
 * bit.  Large launches run one LANE per walk (stored steps as in n2v_walk_fat; the wave serves its lanes' rebuild steps
 * one after the other), small ones one wavefront per walk.  node_fat: fat node tables (first step); fat / recs as for
 * n2v_walk_fat; scratch as for n2v_walk_on_the_fly.                                                                */
int n2v_walk_hybrid(const int64_t* row_ptr, const int32_t* col, const double* w, double p, double q, int32_t symmetric,
                    int64_t max_degree, const struct n2v_fat_slot* node_fat, const struct n2v_fat_slot* fat,
                    const n2v_edge_rec* recs, const int32_t* starts, int64_t n_starts, int64_t pos_begin,
                    int64_t pos_count, int64_t round_begin, int64_t round_count, int32_t walk_length, int32_t rng_mode,
                    const double* uniforms, const int64_t* walk_uoff, uint64_t seed, n2v_alias_slot* scratch,
                    int64_t scratch_slots, int32_t* walks, int32_t* lens, int32_t* status, void* stream);

/* ---- learn_embeddings (src/main.py:82-90 -> gensim 3.2.0 Word2Vec, sg=1, negative sampling) --
 * gensim is a third-party dependency absent from the reference tree (requirements.txt:17);
 * these entry points restate its public algorithm.
 * Embedding tables are fp32 [n_words][row_stride] row-major with row_stride a multiple of
 * 64 floats (64, 128, 256 or 512) >= dim; the padding columns stay 0.  Word index = dense
 * node id.                                                                               */

/* reset_weights: syn0 ~ U(-0.5/dim, 0.5/dim) (Philox, keyed by seed and row), syn1neg = 0. */
int n2v_sgns_init(float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t row_stride,
                  uint64_t seed, void* stream);

/* Bucket index over gensim's cumulative unigram^0.75 table (cum_table: uint32[n_words],
 * non-decreasing, last entry 2^31-1): lut[b] = bisect_left(cum_table, b << (31 - lut_bits)),
 * b in [0, 2^lut_bits]; lut: uint32[2^lut_bits + 1].  A negative draw then costs one
 * bucket read plus a search over that bucket's few entries and returns exactly
 * bisect_left(cum_table, r).                                                             */
int n2v_build_neg_lut(const uint32_t* cum_table, int64_t n_words, int32_t lut_bits, uint32_t* lut,
                      void* stream);

/* One pass over `n_walks` sentences (walks: int32[n_walks][walk_stride], -1 padded; lens may
 * be NULL = all full length).  sample_int: uint32[n_words] keep-thresholds of gensim's
 * sub-sampling (NULL = off).  Learning rate of sentence s (0-based within this call):
 *   alpha - (alpha - min_alpha) * (sentences_base + floor(s / alpha_batch) * alpha_batch
 *           * sentences_step) / sentences_total, floored at min_alpha
 * (gensim steps alpha once per job of <= 10000 words; sentences_step = number of replicas
 * advancing together, 1 on a single GPU).  seed/walk_id_base key the
 * per-sentence random streams (sub-sampling, window shrink, negative draws).
 * pair_count (may be NULL): incremented by the number of (centre, context) pairs trained.
 * update_mode: how the racing wavefronts share rows (gensim's workers race the same way):
 *   N2V_SGNS_PLAIN  plain loads/stores (per-XCD L2 copies; fastest, loses updates),
 *   N2V_SGNS_AGENT  agent-scope loads and stores (one copy at the memory side); the centre word's row, held for a
 *                   whole window, gets its accumulated change ADDED (float atomics) instead of being written back whole,
 *   N2V_SGNS_ATOMIC agent-scope loads + float atomic adds (no update is lost).
 * max_blocks <= 0 picks the default grid: 3072 workgroups of 4 wavefronts (every wave slot), at most one wavefront per 64
 * vocabulary rows, and a whole number of workgroups per CU once there is more
 * than one — the acceptance band (AUC within +-0.002 of the sequential algorithm) was measured to need both
 * (n2v_sgns_default_blocks reports that grid).
 * work_counter: device uint64[1] owned by the caller (one per model; the launch resets it in stream order): sentences are
 * handed to the wavefronts IN ORDER through it, which keeps all waves inside one moving window of the corpus — the
 * sequential algorithm's processing order up to the width of that window.  With atomic rows the link-prediction AUC then
 * equals the sequential comparator's to 3e-5 at any grid (400k-node fixture; static stride: -0.0023 at 3072 workgroups).
 * NULL: static grid stride (wave w trains sentences w, w + n_waves, ...), kept for comparison. */
#define N2V_SGNS_PLAIN 0
#define N2V_SGNS_AGENT 1
#define N2V_SGNS_ATOMIC 2
/* OR-ed into update_mode (opt-in, not gensim's sampling scheme): draw the negatives once per
 * centre word and share them among its context pairs; target rows stay in registers for the
 * whole window (negative <= 7). */
#define N2V_SGNS_SHARE_NEGATIVES 4
/* OR-ed into update_mode: lift the refusal of combinations that were never scored against the comparator
 * (walk_splits > 1 with N2V_SGNS_PLAIN / N2V_SGNS_AGENT: the wavefronts of ONE sentence hit the same rows at once). */
#define N2V_SGNS_UNCHECKED 8
int n2v_sgns_train(const int32_t* walks, const int32_t* lens, int64_t n_walks, int32_t walk_stride,
                   float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t row_stride,
                   int32_t window, int32_t negative, const uint32_t* sample_int,
                   const uint32_t* cum_table, const uint32_t* lut, int32_t lut_bits, float alpha,
                   float min_alpha, int64_t sentences_base, int64_t sentences_step,
                   int64_t sentences_total, int64_t alpha_batch, uint64_t seed, uint64_t walk_id_base,
                   unsigned long long* pair_count, int32_t update_mode, int32_t max_blocks,
                   int32_t walk_splits, unsigned long long* work_counter, void* stream);

/* The same launch with its walk range read from DEVICE memory, so that a captured launch (hipGraph) can be replayed for
 * every merge interval of a pass (the tiered merges issue ~15 000 short launches per pass at 8 GPUs; replayed from a
 * graph they cost no host time).  walks / lens: this rank's whole shard (n_local walks).  interval_state: device
 * int64[2] = {base interval index c, sentences of earlier epochs E}; the launch is sub-interval
 * s = c * subs_per_interval + sub_index of the n_sub_total the pass is cut into and trains the local walks
 * [s * n_local / n_sub_total, (s+1) * n_local / n_sub_total) with sentences_base = E + begin * sentences_step and
 * walk_id_base = E + shard_offset + begin — what n2v_sgns_train is given by the eager driver.  The caller advances
 * interval_state between replays (in stream order).  Other arguments as n2v_sgns_train.                      */
int n2v_sgns_train_span(const int32_t* walks, const int32_t* lens, int64_t n_local, int32_t walk_stride,
                        float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t row_stride,
                        int32_t window, int32_t negative, const uint32_t* sample_int,
                        const uint32_t* cum_table, const uint32_t* lut, int32_t lut_bits, float alpha,
                        float min_alpha, int64_t sentences_step, int64_t sentences_total, int64_t alpha_batch,
                        uint64_t seed, unsigned long long* pair_count, int32_t update_mode, int32_t max_blocks,
                        int32_t walk_splits, const int64_t* interval_state, int32_t sub_index,
                        int32_t subs_per_interval, int64_t n_sub_total, int64_t shard_offset,
                        unsigned long long* work_counter, void* stream);

/* Workgroups of the default SGNS grid (max_blocks <= 0) for a vocabulary of n_words rows and a row mode. */
int32_t n2v_sgns_default_blocks(int64_t n_words, int32_t update_mode);

/* ---- replica merges of the multi-GPU trainer (no counterpart in the reference, whose gensim
 * threads share one table: src/main.py:87 `workers=`) ------------------------------------------------------
 * One process per GPU trains a replica x of a table on its shard; `base` is the copy all ranks agree on.  At
 * the end of an interval a rank's change d = x - xs (xs = x at the interval's start) is sent; the summed
 * changes S come back from the all-reduce (torch.distributed / RCCL, outside this library) and every rank sets
 * base += w[r] * S.  Rows are in one of two tiers (hot_pos[r] >= 0: position in the compact hot buffers; < 0:
 * cold).  HOT rows are merged at once: n2v_merge_snapshot packs their changes, the (small) all-reduce runs,
 * n2v_merge_hot_apply folds the sum in and resets x = xs = base.  COLD rows are merged ONE INTERVAL LATE so
 * that their all-reduce can run under the next interval's n2v_sgns_train: n2v_merge_snapshot folds in the sum
 * that was sent one interval earlier (cold_sum_prev, NULL at the first interval), writes this interval's change
 * to cold_wire and keeps the rank's own not-yet-merged change in x (x = xs = base + d).  n2v_merge_flush ends a
 * pass: the last cold sum is folded in and x = xs = base on every row — identical tables on all ranks.
 * wire_bf16 != 0: the wire buffers hold bfloat16 (round to nearest even), else float.  Tables are
 * fp32[n_rows][stride]; wire buffers [n_rows][stride] (cold; hot rows' entries are written as 0) and
 * [n_hot][stride] (hot).  w: float[n_rows] weight on the SUM of the changes (1 = sum, 1/world = mean).     */
int n2v_merge_snapshot(float* x, float* xs, float* base, int64_t n_rows, int32_t stride, const float* w,
                       const int32_t* hot_pos, const void* cold_sum_prev, void* cold_wire, void* hot_wire,
                       int32_t wire_bf16, void* stream);
int n2v_merge_hot_apply(float* x, float* xs, float* base, int32_t stride, const float* w,
                        const int64_t* hot_rows, int64_t n_hot, const void* hot_sum, int32_t wire_bf16,
                        void* stream);
int n2v_merge_flush(float* x, float* xs, float* base, int64_t n_rows, int32_t stride, const float* w,
                    const int32_t* hot_pos, const void* cold_sum_last, int32_t wire_bf16, void* stream);
/* Tiered pure-sum merges (merge="tsum"): rows are merged at per-row cadences, always by the plain sum of the replicas'
 * changes.  n2v_merge_pack_rows: wire[j] = x[rows[j]] - base[rows[j]] for a row LIST (int64, the rows of the tiers that
 * are due); after the all-reduce n2v_merge_hot_apply (w == 1, xs == x) folds the sum into base and resets x.       */
int n2v_merge_pack_rows(const float* x, const float* base, int32_t stride, const int64_t* rows, int64_t n_list,
                        void* wire, int32_t wire_bf16, void* stream);
/* The same two steps for ALL tables of a merge in one launch each (the hub tiers are merged thousands of times per pass
 * and hold few rows: the launches are what they cost).  tabs[t]: table and base fp32[*][stride], the row list (NULL:
 * every row 0 .. n_rows-1 in order) and its length; the wire buffer holds the lists back to back, [sum n_rows][stride].
 * n2v_tsum_pack: wire[j] = table[row] - base[row];  n2v_tsum_apply: base[row] += wire[j], table[row] = base[row].   */
#define N2V_TSUM_MAX_TABLES 4
typedef struct {
    float* table;
    float* base;
    const int64_t* rows;
    int64_t n_rows;
} n2v_tsum_table;
int n2v_tsum_pack(const n2v_tsum_table* tabs, int32_t n_tabs, int32_t stride, void* wire, int32_t wire_bf16, void* stream);
int n2v_tsum_apply(const n2v_tsum_table* tabs, int32_t n_tabs, int32_t stride, const void* wire, int32_t wire_bf16,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* N2V_HIP_H */
