/* ORACLE — test infrastructure, not product code.
 *
 * Plain-C restatement of the reference's walk path on a dense CSR view
 * (dense id = rank of the node label, rows sorted ascending, see
 * oracle/n2v_oracle.py:to_csr).  Used by tests/ as the checker at sizes the
 * pure-Python restatement cannot reach, and by bench.py's cpu_baseline leg.
 * Nothing under node2vec-by-ecc_amd/ links or loads this file.
 *
 * Pinned: tests/test_oracle_golden.py checks every entry point against the
 * golden vectors captured from the reference's own src/node2vec.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC   (see oracle/Makefile)
 * Citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ MT19937 (numpy legacy) */
/* np.random.seed(s) == init_genrand(s); np.random.rand() == genrand_res53().           */
typedef struct { uint32_t mt[624]; int mti; } orc_mt_t;

static void mt_seed(orc_mt_t* s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->mti = 624;
}

static uint32_t mt_next(orc_mt_t* s) {
    if (s->mti >= 624) {
        uint32_t* mt = s->mt;
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->mti = 0;
    }
    uint32_t y = s->mt[s->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double mt_double(orc_mt_t* s) {
    uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

void orc_mt19937_fill(uint32_t seed, int64_t skip, int64_t n, double* out) {
    orc_mt_t s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < skip; i++) (void)mt_double(&s);
    for (int64_t i = 0; i < n; i++) out[i] = mt_double(&s);
}

/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof(c));
}

static void philox_step(uint64_t seed, uint64_t walk, uint32_t step, double* u1, double* u2) {
    uint32_t c[4] = {(uint32_t)walk, (uint32_t)(walk >> 32), step, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    *u1 = ((c[0] >> 5) * 67108864.0 + (c[1] >> 6)) / 9007199254740992.0;
    *u2 = ((c[2] >> 5) * 67108864.0 + (c[3] >> 6)) / 9007199254740992.0;
}

/* ------------------------------------------------------------------ alias_setup */
/* src/node2vec.py:240-269.  q[] holds the K probabilities on entry (already
 * normalised), K*prob is formed here exactly as :253 does.  `stack` is K ints of
 * scratch: `smaller` grows up from stack[0], `larger` grows down from stack[K-1];
 * both pop from their most recently pushed end, as list.pop() does (:260-261).  */
static void alias_setup_inplace(int64_t K, double* q, int32_t* J, int32_t* stack) {
    int64_t ns = 0, nl = 0;
    for (int64_t k = 0; k < K; k++) {
        q[k] = (double)K * q[k];
        J[k] = 0;
        if (q[k] < 1.0) stack[ns++] = (int32_t)k;
        else stack[K - (++nl)] = (int32_t)k;
    }
    while (ns > 0 && nl > 0) {
        int32_t small = stack[--ns];
        int32_t large = stack[K - nl];
        nl--;
        J[small] = large;
        double t = q[large] + q[small];
        t = t - 1.0;
        q[large] = t;
        if (t < 1.0) stack[ns++] = large;
        else stack[K - (++nl)] = large;
    }
}

int orc_alias_setup(const double* probs, int64_t K, int32_t* J, double* q) {
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)(K > 0 ? K : 1));
    if (!stack) return -2;
    for (int64_t k = 0; k < K; k++) q[k] = probs[k];
    alias_setup_inplace(K, q, J, stack);
    free(stack);
    return 0;
}

/* ------------------------------------------------------------------ table builders */
static int has_edge(const int64_t* row_ptr, const int32_t* col, int32_t u, int32_t v) {
    int64_t lo = row_ptr[u], hi = row_ptr[u + 1];
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (col[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo < row_ptr[u + 1] && col[lo] == v;
}

/* src/node2vec.py:184-188.  Returns 0, or -1 on a zero-sum neighbourhood (the
 * reference raises ZeroDivisionError at :187).  Slot k of node v lives at
 * row_ptr[v]+k in J/q.                                                           */
int orc_build_node_tables(int64_t N, const int64_t* row_ptr, const int32_t* col, const double* w,
                          int32_t* J, double* q) {
    (void)col;
    int64_t maxk = 1;
    for (int64_t v = 0; v < N; v++)
        if (row_ptr[v + 1] - row_ptr[v] > maxk) maxk = row_ptr[v + 1] - row_ptr[v];
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxk);
    if (!stack) return -2;
    for (int64_t v = 0; v < N; v++) {
        int64_t b = row_ptr[v], K = row_ptr[v + 1] - b;
        if (K == 0) continue;
        double norm = 0.0;
        for (int64_t k = 0; k < K; k++) norm = norm + (w ? w[b + k] : 1.0);
        if (norm == 0.0) { free(stack); return -1; }
        for (int64_t k = 0; k < K; k++) q[b + k] = (w ? w[b + k] : 1.0) / norm;
        alias_setup_inplace(K, q + b, J + b, stack);
    }
    free(stack);
    return 0;
}

/* Offsets of the per-edge tables: table of CSR entry e=(src->dst) has deg(dst) slots. */
void orc_edge_offsets(int64_t N, const int64_t* row_ptr, const int32_t* col, int64_t* edge_off) {
    int64_t nnz = row_ptr[N], acc = 0;
    for (int64_t e = 0; e < nnz; e++) {
        edge_off[e] = acc;
        acc += row_ptr[col[e] + 1] - row_ptr[col[e]];
    }
    edge_off[nnz] = acc;
}

/* src/node2vec.py:133-152 for one CSR entry e = (src -> dst). */
static int edge_table(const int64_t* row_ptr, const int32_t* col, const double* w, double p, double q,
                      int32_t src, int32_t dst, int32_t* J, double* qo, int32_t* stack) {
    int64_t b = row_ptr[dst], K = row_ptr[dst + 1] - b;
    double norm = 0.0;
    for (int64_t k = 0; k < K; k++) {
        int32_t nb = col[b + k];
        double wt = w ? w[b + k] : 1.0, u;
        if (nb == src) u = wt / p;
        else if (has_edge(row_ptr, col, nb, src)) u = wt;
        else u = wt / q;
        qo[k] = u;
        norm = norm + u;
    }
    if (K > 0 && norm == 0.0) return -1;
    for (int64_t k = 0; k < K; k++) qo[k] = qo[k] / norm;
    alias_setup_inplace(K, qo, J, stack);
    return 0;
}

/* src/node2vec.py:193-199: one table per CSR entry (for an undirected graph the CSR
 * already holds both (u,v) and (v,u)).                                              */
int orc_build_edge_tables(int64_t N, const int64_t* row_ptr, const int32_t* col, const double* w,
                          double p, double q, const int64_t* edge_off, int32_t* J, double* qo) {
    int64_t maxk = 1;
    for (int64_t v = 0; v < N; v++)
        if (row_ptr[v + 1] - row_ptr[v] > maxk) maxk = row_ptr[v + 1] - row_ptr[v];
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxk);
    if (!stack) return -2;
    for (int64_t src = 0; src < N; src++)
        for (int64_t e = row_ptr[src]; e < row_ptr[src + 1]; e++) {
            int rc = edge_table(row_ptr, col, w, p, q, (int32_t)src, col[e], J + edge_off[e],
                                qo + edge_off[e], stack);
            if (rc) { free(stack); return rc; }
        }
    free(stack);
    return 0;
}

/* ------------------------------------------------------------------ walks */
/* Uniform source for one step.  mode 0: sequential MT19937 stream (the reference's
 * contract: np.random.seed(seed) then two rand() per step, src/node2vec.py:277-278);
 * mode 1: caller-supplied buffer consumed sequentially; mode 2: Philox keyed by
 * (seed; global walk index, step).                                                   */
typedef struct {
    int mode;
    orc_mt_t mt;
    const double* buf;
    int64_t pos;
    uint64_t seed;
} usrc_t;

static void draw2(usrc_t* s, uint64_t walk, uint32_t step, double* u1, double* u2) {
    if (s->mode == 0) { *u1 = mt_double(&s->mt); *u2 = mt_double(&s->mt); }
    else if (s->mode == 1) { *u1 = s->buf[s->pos]; *u2 = s->buf[s->pos + 1]; }
    else philox_step(s->seed, walk, step, u1, u2);
    s->pos += 2;
}

static int32_t alias_draw(const int32_t* J, const double* q, int64_t K, double u1, double u2) {
    int64_t kk = (int64_t)floor(u1 * (double)K); /* src/node2vec.py:277 */
    return (u2 < q[kk]) ? (int32_t)kk : J[kk];   /* :278-281 */
}

/* src/node2vec.py:55-95.  Walk index w = it*n_starts + pos.  walks is W x L int32,
 * padded with -1; lens[w] = number of nodes.  If edge_off == NULL every second-order
 * table is taken to equal the destination's node table (valid iff p == q == 1).
 * Returns the number of uniforms consumed.                                           */
int64_t orc_walk_tables(int64_t N, const int64_t* row_ptr, const int32_t* col,
                        const int32_t* nodeJ, const double* nodeq,
                        const int64_t* edge_off, const int32_t* edgeJ, const double* edgeq,
                        const int32_t* starts, int64_t n_starts, int64_t num_walks, int64_t L,
                        int mode, uint64_t seed, const double* uniforms, uint64_t walk_index_base,
                        int32_t* walks, int32_t* lens) {
    (void)N;
    usrc_t us;
    us.mode = mode; us.buf = uniforms; us.pos = 0; us.seed = seed;
    if (mode == 0) mt_seed(&us.mt, (uint32_t)seed);
    for (int64_t it = 0; it < num_walks; it++)
        for (int64_t ps = 0; ps < n_starts; ps++) {
            int64_t wi = it * n_starts + ps;
            int32_t* out = walks + wi * L;
            int64_t len = 0;
            int32_t cur = starts[ps];
            int64_t e = -1; /* CSR entry (prev -> cur) */
            if (L > 0) out[len++] = cur;
            while (len < L) {
                int64_t b = row_ptr[cur], K = row_ptr[cur + 1] - b;
                if (K == 0) break;
                double u1, u2;
                draw2(&us, walk_index_base + (uint64_t)wi, (uint32_t)(len - 1), &u1, &u2);
                int32_t s;
                if (e < 0 || !edge_off) s = alias_draw(nodeJ + b, nodeq + b, K, u1, u2);
                else s = alias_draw(edgeJ + edge_off[e], edgeq + edge_off[e], K, u1, u2);
                e = b + s;
                cur = col[e];
                out[len++] = cur;
            }
            lens[wi] = (int32_t)len;
            for (int64_t t = len; t < L; t++) out[t] = -1;
        }
    return us.pos;
}

/* src/node2vec.py:34-53,97-111: the same walk with the (prev,cur) table rebuilt at
 * every step instead of looked up.                                                    */
int64_t orc_walk_on_the_fly(int64_t N, const int64_t* row_ptr, const int32_t* col, const double* w,
                            double p, double q,
                            const int32_t* starts, int64_t n_starts, int64_t num_walks, int64_t L,
                            int mode, uint64_t seed, const double* uniforms, uint64_t walk_index_base,
                            int32_t* walks, int32_t* lens) {
    int64_t maxk = 1;
    for (int64_t v = 0; v < N; v++)
        if (row_ptr[v + 1] - row_ptr[v] > maxk) maxk = row_ptr[v + 1] - row_ptr[v];
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxk);
    int32_t* J = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxk);
    double* qq = (double*)malloc(sizeof(double) * (size_t)maxk);
    if (!stack || !J || !qq) { free(stack); free(J); free(qq); return -2; }
    usrc_t us;
    us.mode = mode; us.buf = uniforms; us.pos = 0; us.seed = seed;
    if (mode == 0) mt_seed(&us.mt, (uint32_t)seed);
    int64_t rc = 0;
    for (int64_t it = 0; it < num_walks && rc == 0; it++)
        for (int64_t ps = 0; ps < n_starts; ps++) {
            int64_t wi = it * n_starts + ps;
            int32_t* out = walks + wi * L;
            int64_t len = 0;
            int32_t cur = starts[ps], prev = -1;
            if (L > 0) out[len++] = cur;
            while (len < L) {
                int64_t b = row_ptr[cur], K = row_ptr[cur + 1] - b;
                if (K == 0) break;
                if (prev < 0) {
                    double norm = 0.0;
                    for (int64_t k = 0; k < K; k++) norm = norm + (w ? w[b + k] : 1.0);
                    if (norm == 0.0) { rc = -1; break; }
                    for (int64_t k = 0; k < K; k++) qq[k] = (w ? w[b + k] : 1.0) / norm;
                    alias_setup_inplace(K, qq, J, stack);
                } else if (edge_table(row_ptr, col, w, p, q, prev, cur, J, qq, stack)) {
                    rc = -1;
                    break;
                }
                double u1, u2;
                draw2(&us, walk_index_base + (uint64_t)wi, (uint32_t)(len - 1), &u1, &u2);
                int32_t s = alias_draw(J, qq, K, u1, u2);
                prev = cur;
                cur = col[b + s];
                out[len++] = cur;
            }
            if (rc) break;
            lens[wi] = (int32_t)len;
            for (int64_t t = len; t < L; t++) out[t] = -1;
        }
    free(stack); free(J); free(qq);
    return rc ? rc : us.pos;
}
