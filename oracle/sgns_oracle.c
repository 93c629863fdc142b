/* ORACLE — test infrastructure, not product code.
 *
 * CPU restatement of the skip-gram / negative-sampling training that the reference's
 * learn_embeddings() delegates to gensim 3.2.0 (src/main.py:82-90, requirements.txt:17).
 *
 * PARITY UNPINNED: gensim is a third-party dependency that is neither in the reference
 * tree nor installable here, and the reference holds no test vector for this step.  The
 * code below follows gensim 3.2.0's published algorithm (word2vec.py: job batching of
 * <= 10000 words, linear alpha decay per job by sentences pushed; word2vec_inner.pyx:
 * train_batch_sg / fast_sentence_sg_neg, the 48-bit LCG, EXP_TABLE of 1000 bins over
 * [-6, 6), cum_table bisect) as summarised in SURVEY.md 8(a) row 9.  It is the comparator
 * for the link-prediction AUC band (+-0.002) and the CPU baseline of bench.py; nothing
 * under node2vec-by-ecc_amd/ links or loads it.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXP_TABLE_SIZE 1000
#define MAX_EXP 6
#define MAX_WORDS_IN_BATCH 10000

static float EXP_TABLE[EXP_TABLE_SIZE];
static int exp_ready = 0;

static void init_exp_table(void) {
    if (exp_ready) return;
    for (int i = 0; i < EXP_TABLE_SIZE; i++) {
        float x = ((float)i / (float)EXP_TABLE_SIZE * 2.0f - 1.0f) * (float)MAX_EXP;
        float e = (float)exp((double)x);
        EXP_TABLE[i] = (float)(e / (e + 1.0f));
    }
    exp_ready = 1;
}

/* ---- numpy legacy RandomState pieces used by gensim's model.random ------------------- */
typedef struct { uint32_t mt[624]; int mti; } mt_t;

static void mt_seed(mt_t* s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->mti = 624;
}

static uint32_t mt_next(mt_t* s) {
    if (s->mti >= 624) {
        uint32_t* mt = s->mt;
        for (int kk = 0; kk < 624; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % 624] & 0x7fffffffu);
            mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->mti = 0;
    }
    uint32_t y = s->mt[s->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* RandomState.randint(0, high): masked rejection on 32-bit draws */
static uint32_t mt_randint(mt_t* s, uint32_t high) {
    uint32_t rng = high - 1, mask = rng;
    if (rng == 0) return 0;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (mt_next(s) & mask)) > rng) {}
    return v;
}

void orc_randint_fill(uint32_t seed, uint32_t high, int64_t n, int64_t* out) {
    mt_t s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < n; i++) out[i] = mt_randint(&s, high);
}

/* ---- Philox init shared with the device kernel (same stream, so CPU and GPU runs can
 *      start from identical tables): syn0 ~ U(-0.5/d, 0.5/d), syn1neg = 0 ---------------- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

void orc_sgns_init(float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t stride, uint64_t seed) {
    for (int64_t row = 0; row < n_words; row++)
        for (int cb = 0; cb < stride / 4; cb++) {
            uint32_t c[4] = {(uint32_t)row, (uint32_t)(row >> 32), (uint32_t)cb, 0x5EEDu};
            philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            for (int t = 0; t < 4; t++) {
                int col = cb * 4 + t;
                float u = (float)(c[t] >> 8) * (1.0f / 16777216.0f);
                syn0[row * stride + col] = col < dim ? (u - 0.5f) / (float)dim : 0.0f;
                syn1neg[row * stride + col] = 0.0f;
            }
        }
}

/* ---- training ---------------------------------------------------------------------- */
typedef struct {
    const int32_t* walks; const int32_t* lens; int64_t n_walks; int32_t L;
    float* syn0; float* syn1neg; int64_t n_words; int32_t dim, stride, window, negative;
    const uint32_t* sample_int; const uint32_t* cum_table;
    float alpha, min_alpha; int32_t epochs; uint32_t seed;
    /* jobs: consecutive sentences holding <= MAX_WORDS_IN_BATCH words */
    int64_t n_jobs; const int64_t* job_start; /* n_jobs + 1 */
    volatile int64_t next_job; int64_t pairs; pthread_mutex_t mu; int n_threads;
} ctx_t;

static inline uint32_t bisect_left_u32(const uint32_t* a, int64_t n, uint32_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return (uint32_t)lo;
}

/* word2vec_inner.pyx fast_sentence_sg_neg */
static uint64_t sg_neg_pair(ctx_t* c, int32_t word_index, int32_t word2_index, float alpha, float* work,
                            uint64_t next_random) {
    const int dim = c->dim;
    float* h = c->syn0 + (int64_t)word2_index * c->stride;
    memset(work, 0, sizeof(float) * (size_t)dim);
    for (int d = 0; d < c->negative + 1; d++) {
        int32_t target; float label;
        if (d == 0) { target = word_index; label = 1.0f; }
        else {
            target = (int32_t)bisect_left_u32(c->cum_table, c->n_words,
                                              (uint32_t)((next_random >> 16) % c->cum_table[c->n_words - 1]));
            next_random = (next_random * 25214903917ULL + 11ULL) & 281474976710655ULL;
            if (target == word_index) continue;
            label = 0.0f;
        }
        float* row2 = c->syn1neg + (int64_t)target * c->stride;
        float f = 0.0f;
        for (int k = 0; k < dim; k++) f += h[k] * row2[k];
        if (f <= -MAX_EXP || f >= MAX_EXP) continue;
        f = EXP_TABLE[(int)((f + MAX_EXP) * (EXP_TABLE_SIZE / MAX_EXP / 2))];
        float g = (label - f) * alpha;
        for (int k = 0; k < dim; k++) work[k] += g * row2[k];
        for (int k = 0; k < dim; k++) row2[k] += g * h[k];
    }
    for (int k = 0; k < dim; k++) h[k] += work[k];
    return next_random;
}

/* one job = word2vec_inner.pyx train_batch_sg over its sentences */
static int64_t run_job(ctx_t* c, int64_t job, int epoch, mt_t* rs, int32_t* idx, int32_t* sent_end,
                       uint32_t* reduced, float* work) {
    int64_t total_jobs = c->n_jobs * c->epochs;
    (void)total_jobs;
    /* alpha of the job: sentences pushed before it, over all epochs (word2vec.py job_producer) */
    int64_t pushed = (int64_t)epoch * c->n_walks + c->job_start[job];
    double progress = (double)pushed / (double)((int64_t)c->epochs * c->n_walks);
    float alpha = (float)(c->alpha - (c->alpha - c->min_alpha) * progress);
    if (alpha < c->min_alpha) alpha = c->min_alpha;

    uint64_t next_random = ((uint64_t)1 << 24) * mt_randint(rs, 1u << 24);
    next_random += mt_randint(rs, 1u << 24);
    int64_t eff = 0, ns = 0;
    for (int64_t s = c->job_start[job]; s < c->job_start[job + 1]; s++) {
        int len = c->lens ? c->lens[s] : c->L;
        for (int t = 0; t < len; t++) {
            int32_t w = c->walks[s * c->L + t];
            if (w < 0) continue;
            if (c->sample_int) {
                uint32_t r = (uint32_t)(next_random >> 16);
                next_random = (next_random * 25214903917ULL + 11ULL) & 281474976710655ULL;
                if (c->sample_int[w] < r) continue;
            }
            idx[eff++] = w;
        }
        sent_end[ns++] = (int32_t)eff;
    }
    for (int64_t i = 0; i < eff; i++) reduced[i] = mt_randint(rs, (uint32_t)c->window);
    int64_t pairs = 0, start = 0;
    for (int64_t s = 0; s < ns; s++) {
        int64_t end = sent_end[s];
        for (int64_t i = start; i < end; i++) {
            int64_t j = i - c->window + reduced[i];
            if (j < start) j = start;
            int64_t k = i + c->window + 1 - reduced[i];
            if (k > end) k = end;
            for (; j < k; j++) {
                if (j == i) continue;
                next_random = sg_neg_pair(c, idx[i], idx[j], alpha, work, next_random);
                pairs++;
            }
        }
        start = end;
    }
    return pairs;
}

static void* worker(void* arg) {
    ctx_t* c = (ctx_t*)arg;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (MAX_WORDS_IN_BATCH + c->L));
    int32_t* sent_end = (int32_t*)malloc(sizeof(int32_t) * (MAX_WORDS_IN_BATCH + 1));
    uint32_t* reduced = (uint32_t*)malloc(sizeof(uint32_t) * (MAX_WORDS_IN_BATCH + c->L));
    float* work = (float*)malloc(sizeof(float) * (size_t)c->dim);
    mt_t rs;
    mt_seed(&rs, c->seed);  /* single worker: gensim's model.random = RandomState(seed) */
    int64_t pairs = 0, total = c->n_jobs * c->epochs;
    for (;;) {
        int64_t g = __sync_fetch_and_add(&c->next_job, 1);
        if (g >= total) break;
        if (c->n_threads > 1) mt_seed(&rs, c->seed + (uint32_t)g * 2654435761u + 1u);
        pairs += run_job(c, g % c->n_jobs, (int)(g / c->n_jobs), &rs, idx, sent_end, reduced, work);
    }
    pthread_mutex_lock(&c->mu);
    c->pairs += pairs;
    pthread_mutex_unlock(&c->mu);
    free(idx); free(sent_end); free(reduced); free(work);
    return NULL;
}

/* Returns the number of (centre, context) pairs trained.  n_threads == 1 is the
 * deterministic comparator; n_threads > 1 is Hogwild (lock-free shared tables) for the
 * all-core CPU baseline.                                                                */
int64_t orc_sgns_train(const int32_t* walks, const int32_t* lens, int64_t n_walks, int32_t L, float* syn0,
                       float* syn1neg, int64_t n_words, int32_t dim, int32_t stride, int32_t window,
                       int32_t negative, const uint32_t* sample_int, const uint32_t* cum_table, float alpha,
                       float min_alpha, int32_t epochs, uint32_t seed, int32_t n_threads) {
    init_exp_table();
    ctx_t c;
    memset(&c, 0, sizeof(c));
    c.walks = walks; c.lens = lens; c.n_walks = n_walks; c.L = L; c.syn0 = syn0; c.syn1neg = syn1neg;
    c.n_words = n_words; c.dim = dim; c.stride = stride; c.window = window; c.negative = negative;
    c.sample_int = sample_int; c.cum_table = cum_table; c.alpha = alpha; c.min_alpha = min_alpha;
    c.epochs = epochs; c.seed = seed; c.n_threads = n_threads < 1 ? 1 : n_threads;
    /* job boundaries (word2vec.py _job_producer: a sentence joins the batch while the raw
     * word count stays <= batch_words) */
    int64_t* js = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_walks + 2));
    int64_t nj = 0, words = 0;
    js[0] = 0;
    for (int64_t s = 0; s < n_walks; s++) {
        int len = lens ? lens[s] : L;
        if (words + len <= MAX_WORDS_IN_BATCH || words == 0) words += len;
        else { js[++nj] = s; words = len; }
    }
    js[++nj] = n_walks;
    c.n_jobs = nj; c.job_start = js;
    pthread_mutex_init(&c.mu, NULL);
    if (c.n_threads == 1) worker(&c);
    else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)c.n_threads);
        for (int t = 0; t < c.n_threads; t++) pthread_create(&th[t], NULL, worker, &c);
        for (int t = 0; t < c.n_threads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&c.mu);
    free(js);
    return c.pairs;
}
