// numpy's legacy MT19937 `random_sample` stream, generated on the device — gfx950 (MI355X).
//
// The reference draws its walk uniforms from numpy's GLOBAL MT19937 state
// (np.random.rand() twice per step, src/node2vec.py:277-278).  Parity mode must consume that
// exact stream; generating it with numpy on the host and uploading it caps the whole walk at
// the host's ~2e8 doubles/s.  MT19937 is a linear recurrence over GF(2), so the stream can be
// cut into P contiguous pieces whose start states are obtained by polynomial jump-ahead
// (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008):
//
//   window_k = A^(k*S) window_0 = g_k(A) window_0,  g_k(x) = x^(k*S) mod (x * phi(x))
//
// where `window` is 624 consecutive words of the sequence, A slides it by one word, phi is the
// degree-19937 minimal polynomial of the sequence (found once by Berlekamp-Massey) and the
// extra factor x makes the identity hold on all 19968 bits of a window (the low 31 bits of
// its oldest word are outside the 19937-bit state but may still have to be output).
// Host code below does the GF(2) polynomial work (a few ms per stream); the kernel then runs
// numpy's own algorithm — lazy block twist, tempering, (a>>5, b>>6) -> double — one wavefront
// per stream with the 624-word window in LDS, and hands back the final (key, pos) so the
// caller can leave numpy's global state exactly where the reference would have left it.
#include <cstring>
#include <mutex>
#include <vector>

#include "n2v_common.h"

namespace {

constexpr int kN = 624, kM = 397;
constexpr uint32_t kMatrixA = 0x9908b0dfu, kUpper = 0x80000000u, kLower = 0x7fffffffu;
constexpr int kDeg = 19937;               // degree of phi
constexpr int kModDeg = kDeg + 1;         // modulus x*phi(x)
constexpr int kPW = 312;                  // 64-bit words per reduced polynomial (19968 bits)

// ------------------------------------------------------------------ host: sliding window
struct Window {  // logical word j = w[(head + j) % 624]
    uint32_t w[kN];
    int head;
};

inline void window_step(Window& s) {  // A: drop x_k, append x_{k+624}
    const int i0 = s.head, i1 = (s.head + 1) % kN, im = (s.head + kM) % kN;
    const uint32_t y = (s.w[i0] & kUpper) | (s.w[i1] & kLower);
    s.w[i0] = s.w[im] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
    s.head = i1;
}

// ------------------------------------------------------------------ host: GF(2)[x]
typedef std::vector<uint64_t> Poly;  // little-endian bits

inline bool pbit(const Poly& p, int i) { return (p[i >> 6] >> (i & 63)) & 1ULL; }

// r ^= b << sh   (b has nb words)
inline void xor_shifted(uint64_t* r, const uint64_t* b, int nb, int sh) {
    const int ws = sh >> 6, bs = sh & 63;
    if (bs == 0) {
        for (int i = 0; i < nb; ++i) r[ws + i] ^= b[i];
    } else {
        uint64_t carry = 0;
        for (int i = 0; i < nb; ++i) {
            r[ws + i] ^= (b[i] << bs) | carry;
            carry = b[i] >> (64 - bs);
        }
        r[ws + nb] ^= carry;
    }
}

struct Mt {
    Poly mod;  // x * phi(x), degree kModDeg, kPW+1 words
    bool ready = false;
    std::mutex mu;
};
Mt g_mt;

// minimal polynomial of the MT19937 bit stream by Berlekamp-Massey (2*19937 bits suffice)
void build_modulus() {
    std::lock_guard<std::mutex> lk(g_mt.mu);
    if (g_mt.ready) return;
    const int nbits = 2 * kDeg + 64;
    Window s;
    s.w[0] = 5489u;  // any non-degenerate seed: init_genrand(5489)
    for (int i = 1; i < kN; ++i) s.w[i] = 1812433253u * (s.w[i - 1] ^ (s.w[i - 1] >> 30)) + (uint32_t)i;
    s.head = 0;
    const int W = (kDeg + 64) / 64 + 2;
    std::vector<uint64_t> C(W, 0), B(W, 0), T(W, 0), R(W, 0);
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int N = 0; N < nbits; ++N) {
        window_step(s);
        const uint64_t bit = s.w[(s.head + kN - 1) % kN] & 1u;  // newest word's LSB
        // R: bit i = b_{N-i}
        uint64_t carry = bit;
        for (int i = 0; i < W; ++i) {
            const uint64_t nc = R[i] >> 63;
            R[i] = (R[i] << 1) | carry;
            carry = nc;
        }
        uint64_t acc = 0;
        const int lw = L / 64 + 1;
        for (int i = 0; i < lw && i < W; ++i) acc ^= C[i] & R[i];
        if (__builtin_parityll(acc)) {
            if (2 * L <= N) {
                T = C;
                xor_shifted(C.data(), B.data(), W - (m >> 6) - 2, m);
                L = N + 1 - L;
                B = T;
                m = 1;
            } else {
                xor_shifted(C.data(), B.data(), W - (m >> 6) - 2, m);
                ++m;
            }
        } else {
            ++m;
        }
    }
    // phi(x) = x^L * C(1/x); modulus = x * phi(x)
    Poly mod(kPW + 1, 0);
    if (L == kDeg) {
        for (int i = 0; i <= L; ++i)
            if ((C[i >> 6] >> (i & 63)) & 1ULL) {
                const int e = L - i + 1;  // +1: times x
                mod[e >> 6] |= 1ULL << (e & 63);
            }
        g_mt.mod = mod;
        g_mt.ready = true;
    }
}

// r = a * b mod M   (a, b reduced: degree < kModDeg)
void mulmod(const Poly& a, const Poly& b, Poly& r) {
    std::vector<uint64_t> t(2 * kPW + 2, 0);
    for (int i = 0; i < kModDeg; ++i)
        if (pbit(a, i)) xor_shifted(t.data(), b.data(), kPW, i);
    const Poly& M = g_mt.mod;
    for (int d = 2 * kModDeg; d >= kModDeg; --d)
        if ((t[d >> 6] >> (d & 63)) & 1ULL) xor_shifted(t.data(), M.data(), kPW + 1, d - kModDeg);
    r.assign(t.begin(), t.begin() + kPW);
}

// x^n mod M by square-and-multiply (any n)
Poly powmod_x_generic(uint64_t n) {
    Poly r(kPW, 0), t;
    r[0] = 1;
    if (n == 0) return r;
    int top = 63;
    while (!((n >> top) & 1ULL)) --top;
    for (int b = top; b >= 0; --b) {
        mulmod(r, r, t);
        r.swap(t);
        if ((n >> b) & 1ULL) {  // times x: shift by one and reduce once
            uint64_t carry = 0;
            for (int i = 0; i < kPW; ++i) {
                const uint64_t nc = r[i] >> 63;
                r[i] = (r[i] << 1) | carry;
                carry = nc;
            }
            if (pbit(r, kModDeg)) {
                for (int i = 0; i < kPW; ++i) r[i] ^= g_mt.mod[i];
                // bit kModDeg lies inside word kModDeg>>6 < kPW, cleared by the xor above
            }
        }
    }
    return r;
}

// The family x^(624 * 2^m) mod M, m = 0, 1, 2, ...: every whole-block jump is a product of members (one per set
// bit of the block count), and the doubling scheme of n2v_mt19937_jump_device with a stride of 624 * 2^k words
// uses members k, k+1, ... directly — so after the first ~40 squarings (0.5 ms each) no call computes a
// polynomial power again.
struct Family {
    std::vector<Poly> p;
    std::mutex mu;
};
Family g_fam;

Poly family_member(int m) {
    std::lock_guard<std::mutex> lk(g_fam.mu);
    Poly t;
    while ((int)g_fam.p.size() <= m) {
        if (g_fam.p.empty()) {
            g_fam.p.push_back(powmod_x_generic((uint64_t)kN));
        } else {
            mulmod(g_fam.p.back(), g_fam.p.back(), t);
            g_fam.p.push_back(t);
        }
    }
    return g_fam.p[m];
}

// x^n mod M
Poly powmod_x(uint64_t n) { return powmod_x_generic(n); }

// out = g(A) in  (Horner); windows as 624 logical words
// A^i w is words i..i+623 of the sequence x that continues w, so (g(A) w)[j] = XOR over the
// set bits i of g of x[i + j]: generate x once, then one contiguous 624-word XOR per set bit
// (vectorises; ~10^4 set bits).
void apply_poly(const Poly& g, const uint32_t* in, uint32_t* out) {
    int deg = kModDeg - 1;
    while (deg > 0 && !pbit(g, deg)) --deg;
    std::vector<uint32_t> x((size_t)deg + kN + 1);
    std::memcpy(x.data(), in, sizeof(uint32_t) * kN);
    for (int n = kN; n < deg + kN; ++n) {
        const uint32_t y = (x[n - kN] & kUpper) | (x[n - kN + 1] & kLower);
        x[n] = x[n - kN + kM] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
    }
    uint32_t acc[kN];
    std::memset(acc, 0, sizeof(acc));
    for (int i = 0; i <= deg; ++i) {
        if (!pbit(g, i)) continue;
        const uint32_t* xi = x.data() + i;
        for (int j = 0; j < kN; ++j) acc[j] ^= xi[j];
    }
    std::memcpy(out, acc, sizeof(acc));
}

// ------------------------------------------------------------------ device: one wave per stream
__device__ __forceinline__ uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ __forceinline__ void lds_order() {  // one wave per workgroup: LDS is in order, pin the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// new value of word i of the block twist, from the (partly updated) window in LDS
__device__ __forceinline__ uint32_t twist_word(const uint32_t* mt, int i) {
    int i1 = i + 1;
    if (i1 == kN) i1 = 0;
    int im = i + kM;
    if (im >= kN) im -= kN;
    const uint32_t y = (mt[i] & kUpper) | (mt[i1] & kLower);
    return mt[im] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
}

// Where double d of the stream goes.  Linear: out[d].  Tiled (n2v_mt19937_fill_tiled): the stream is cut into
// segments of dpw = 2 * pairs doubles (one walk each); segment s, pair t, component c lands at
// out[2 * (((s >> 6) * pairs + t) * 64 + (s & 63)) + c], so that step t of 64 consecutive walks is 1 KiB of
// consecutive bytes.  A stream starts at segment s0, remainder r0 (one 64-bit division per stream); inside the stream
// the split of r0 + (d - d0) < 2^32 uses a host-made reciprocal with one correction step.
struct TileMap {
    uint32_t dpw, pairs, recip;   // recip = floor(2^32 / dpw)
};

template <bool TILED>
struct OutMap {
    double* out;
    int64_t d0, s0;
    uint32_t r0;
    TileMap m;
    __device__ __forceinline__ void put(int64_t d, double v) const {
        if (!TILED) {
            out[d] = v;
            return;
        }
        const uint32_t y = r0 + (uint32_t)(d - d0);
        uint32_t qd = __umulhi(y, m.recip);
        uint32_t rem = y - qd * m.dpw;
        if (rem >= m.dpw) { rem -= m.dpw; ++qd; }
        const int64_t s = s0 + qd;
        out[2 * (((s >> 6) * m.pairs + (rem >> 1)) * 64 + (s & 63)) + (rem & 1u)] = v;
    }
};

// MODE 0: linear output.  MODE 1: tiled, every double written where it lands (16-B pieces of lines: 3x the time of the
// linear fill on C3).  MODE 2: tiled through an LDS ring: a wavefront's stream is consumed in groups of four walk
// segments (8 * pairs doubles); the four segments' pairs of one step are 64 contiguous, 64-B aligned bytes of the tiled
// layout, so a finished group leaves as whole lines (lane e writes pair e >> 2 of segment e & 3).  Only the ragged ends
// of a stream (its first and last partial group) go through MODE 1's element-wise path.
constexpr int kFillLdsHead = (kN + 16 + 4) * 4;   // 2576 B = 161 * 16

template <int MODE>
__global__ void __launch_bounds__(64)
mt_fill_kernel(const uint32_t* __restrict__ states, int pos0, int64_t words_per_stream, int64_t total_words,
               double* __restrict__ out_ptr, uint32_t* __restrict__ final_state, TileMap tm, int ring_log2) {
    // all LDS in the dynamic region, carved at multiples of 16 B (a static in front would shift the ring off the 16-B
    // alignment its double2 reads need): window [640 words], the carried word, then MODE 2's ring of 2^ring_log2 doubles
    extern __shared__ __attribute__((aligned(16))) unsigned char fill_smem[];
    uint32_t* mt = reinterpret_cast<uint32_t*>(fill_smem);
    uint32_t& carry_word = mt[kN + 16];
    double* ring = reinterpret_cast<double*>(fill_smem + kFillLdsHead);
    const int lane = threadIdx.x;
    const int64_t k = blockIdx.x;
    OutMap<MODE != 0> out{out_ptr, 0, 0, 0u, tm};
    const int64_t d_begin = (k * words_per_stream) >> 1;   // words_per_stream is even: no double straddles two streams
    if (MODE != 0) {
        out.d0 = d_begin;
        out.s0 = out.d0 / tm.dpw;
        out.r0 = (uint32_t)(out.d0 - out.s0 * tm.dpw);
    }
    const uint32_t mask = MODE == 2 ? (1u << ring_log2) - 1u : 0u;
    const int64_t G = 4 * (int64_t)tm.dpw;     // doubles per group of four segments
    int64_t flushed = d_begin;                 // MODE 2: first double not yet written to global memory
    for (int i = lane; i < kN; i += 64) mt[i] = states[k * kN + i];
    lds_order();
    int pos = pos0;
    int64_t produced = k * words_per_stream;
    const int64_t w_end = min(total_words, produced + words_per_stream);

    auto emit = [&](int64_t d, double v) {
        if (MODE == 2) ring[(uint32_t)d & mask] = v;
        else out.put(d, v);
    };
    // MODE 2: write [flushed, upto) from the ring; whole groups as 64-B lines, the rest double by double
    auto flush = [&](int64_t upto) {
        while (flushed < upto) {
            const int64_t gi = flushed / G;
            const int64_t gend = (gi + 1) * G;
            if (flushed == gi * G && gend <= upto) {
                const int64_t s4 = 4 * gi;                                 // first segment of the group
                double* base = out_ptr + 2 * (((s4 >> 6) * tm.pairs) * 64 + (s4 & 63));
                for (uint32_t e = lane; e < 4 * tm.pairs; e += 64) {
                    const uint32_t j = e & 3u, t = e >> 2;
                    const uint32_t src = ((uint32_t)flushed + j * tm.dpw + 2 * t) & mask;   // even: the pair never wraps
                    const double2 v = *reinterpret_cast<const double2*>(&ring[src]);
                    *reinterpret_cast<double2*>(base + 2 * ((int64_t)t * 64 + j)) = v;
                }
                flushed = gend;
            } else {
                const int64_t stop = min(gend, upto);
                for (int64_t d = flushed + lane; d < stop; d += 64) out.put(d, ring[(uint32_t)d & mask]);
                flushed = stop;
            }
        }
    };

    while (produced < w_end) {
        if (pos >= kN) {
            // numpy's block twist, in place.  Word i needs old i, i+1 and (i < 227 ? old i+397
            // : NEW i-227).  64-word chunks c, c+1, c+2 never read each other's outputs
            // (227 > 3*64), so three chunks share one LDS round trip: all reads, then all writes.
#pragma unroll
            for (int c = 0; c < 576; c += 192) {
                const uint32_t v0 = twist_word(mt, c + lane);
                const uint32_t v1 = twist_word(mt, c + 64 + lane);
                const uint32_t v2 = twist_word(mt, c + 128 + lane);
                lds_order();
                mt[c + lane] = v0;
                mt[c + 64 + lane] = v1;
                mt[c + 128 + lane] = v2;
                lds_order();
            }
            {
                uint32_t v = 0;
                if (lane < kN - 576) v = twist_word(mt, 576 + lane);
                lds_order();
                if (lane < kN - 576) mt[576 + lane] = v;
                lds_order();
            }
            pos = 0;
        }
        const int take = (int)min((int64_t)(kN - pos), w_end - produced);
        // stream word j = produced + idx sits in mt[pos + idx]; even j = first half of a double
        const int odd = (int)(produced & 1);
        if (odd && lane == 0) {
            const uint32_t a = carry_word >> 5, b = temper(mt[pos]) >> 6;
            emit((produced - 1) >> 1, ((double)a * 67108864.0 + (double)b) / 9007199254740992.0);
        }
        for (int ia = odd + 2 * lane; ia < take; ia += 128) {
            const uint32_t ya = temper(mt[pos + ia]);
            if (ia + 1 < take) {
                const uint32_t a = ya >> 5, b = temper(mt[pos + ia + 1]) >> 6;
                emit((produced + ia) >> 1, ((double)a * 67108864.0 + (double)b) / 9007199254740992.0);
            } else {
                carry_word = ya;  // its partner is word 0 of the next block
            }
        }
        lds_order();
        pos += take;
        produced += take;
        if (MODE == 2) {
            // complete doubles so far: [d_begin, produced >> 1); keep less than one group in the ring
            const int64_t done = produced >> 1;
            if (done - flushed >= G || produced >= w_end) {
                flush(produced >= w_end ? done : (done / G) * G);
                lds_order();
            }
        }
    }
    if (w_end == total_words && final_state && k * words_per_stream < total_words) {
        for (int i = lane; i < kN; i += 64) final_state[i] = mt[i];
        if (lane == 0) final_state[kN] = (uint32_t)pos;
    }
}

// ------------------------------------------------------------------ device: jump-ahead by doubling
// states[src_count + b] = g(A) states[b] for b < n_new, g given by its 19968 coefficient bits.  One workgroup per
// new stream: the sequence x that continues the source window (deg + 624 words, 82 KB) is generated into LDS —
// 192 words per round, a word depends only on words at least 227 places back — and (g(A) w)[j] = XOR over the
// set bits i of g of x[i + j] is accumulated with every thread holding three of the 624 window words.
constexpr int kJumpThreads = 640;  // ten wavefronts: one window word per thread, latency hidden by the other waves
constexpr int kSeqWords = kModDeg + kN;   // 20592
constexpr int kPosRow = 19968;            // uint32 per polynomial: [0] = number of set bits, then their positions

__global__ void __launch_bounds__(kJumpThreads)
mt_jump_kernel(uint32_t* __restrict__ states, const uint32_t* __restrict__ pos, int src_count, int n_new) {
    extern __shared__ uint32_t x[];  // [kSeqWords] words of the sequence, then the set-bit positions as uint16
    uint16_t* P = reinterpret_cast<uint16_t*>(x + kSeqWords);
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= n_new) return;
    const uint32_t* in = states + (size_t)b * kN;
    for (int i = t; i < kN; i += kJumpThreads) x[i] = in[i];
    const int n_set = (int)pos[0];
    const int deg = n_set ? (int)pos[n_set] : 0;  // positions ascend: the last one is the degree
    for (int i = t; i < n_set; i += kJumpThreads) P[i] = (uint16_t)pos[1 + i];
    __syncthreads();
    for (int n0 = kN; n0 < deg + kN; n0 += 192) {
        const int n = n0 + t;
        if (t < 192 && n < deg + kN) {
            const uint32_t y = (x[n - kN] & kUpper) | (x[n - kN + 1] & kLower);
            x[n] = x[n - kN + kM] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
        }
        __syncthreads();
    }
    // per set bit a thread issues one LDS read and one XOR; the positions sit in LDS as uint16 (staged by the
    // first loop above) and are consumed eight at a time so that the reads of one group overlap
    uint32_t a0 = 0;
    const uint32_t* xt = x + (t < kN ? t : 0);
    int k = 0;
    for (; k + 8 <= n_set; k += 8) {
        uint32_t i[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) i[j] = P[k + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) a0 ^= xt[i[j]];
    }
    for (; k < n_set; ++k) a0 ^= xt[P[k]];
    if (t < kN) states[(size_t)(src_count + b) * kN + t] = a0;
}

}  // namespace

extern "C" int n2v_mt19937_jump_host(const uint32_t* key_host, int64_t stride_words, int32_t n_streams,
                                     uint32_t* states_host) {
    if (!key_host || !states_host || n_streams < 1 || stride_words < 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_mt19937_jump_host: bad argument");
    build_modulus();
    if (!g_mt.ready) return n2v::fail(N2V_ERR_INVALID, "n2v_mt19937_jump_host: minimal polynomial not found");
    std::memcpy(states_host, key_host, sizeof(uint32_t) * kN);
    if (n_streams == 1) return N2V_OK;
    if (stride_words > 0 && stride_words % kN == 0) {
        // whole blocks: apply the family members of the set bits of the block count one after another
        // (0.5 ms each) instead of raising x to a new power
        uint32_t tmp[kN];
        for (int k = 1; k < n_streams; ++k) {
            uint32_t* cur = states_host + (size_t)k * kN;
            std::memcpy(cur, states_host + (size_t)(k - 1) * kN, sizeof(uint32_t) * kN);
            uint64_t blocks = (uint64_t)(stride_words / kN);
            for (int m = 0; blocks; ++m, blocks >>= 1)
                if (blocks & 1ULL) {
                    apply_poly(family_member(m), cur, tmp);
                    std::memcpy(cur, tmp, sizeof(tmp));
                }
        }
        return N2V_OK;
    }
    // jump polynomials of the last few strides (a walk alternates between the per-stream stride
    // and the whole-batch stride that moves numpy's global state)
    static std::mutex cache_mu;
    static std::vector<std::pair<uint64_t, Poly>> cache;
    Poly g;
    {
        std::lock_guard<std::mutex> lk(cache_mu);
        bool hit = false;
        for (size_t i = 0; i < cache.size(); ++i)
            if (cache[i].first == (uint64_t)stride_words) {
                g = cache[i].second;
                hit = true;
                break;
            }
        if (!hit) {
            g = powmod_x((uint64_t)stride_words);
            if (cache.size() >= 8) cache.erase(cache.begin());
            cache.emplace_back((uint64_t)stride_words, g);
        }
    }
    for (int k = 1; k < n_streams; ++k)
        apply_poly(g, states_host + (size_t)(k - 1) * kN, states_host + (size_t)k * kN);
    return N2V_OK;
}

// Set-bit positions of x^(stride_words * 2^r) mod x*phi(x), r = 0 .. n_rounds-1, as uint32[n_rounds][19968] on the
// host: word 0 = number of set bits, then the positions in ascending order — the jump polynomials of the
// doubling scheme used by n2v_mt19937_jump_device.
extern "C" int n2v_mt19937_jump_polys_host(int64_t stride_words, int32_t n_rounds, uint32_t* polys_host) {
    if (!polys_host || stride_words < 1 || n_rounds < 1 || n_rounds > 32)
        return n2v::fail(N2V_ERR_INVALID, "n2v_mt19937_jump_polys_host: bad argument");
    build_modulus();
    if (!g_mt.ready) return n2v::fail(N2V_ERR_INVALID, "n2v_mt19937_jump_polys_host: minimal polynomial not found");
    // a stride of 624 * 2^k words: the rounds' polynomials are family members k, k+1, ...
    int fam_k = -1;
    if (stride_words % kN == 0) {
        const uint64_t blocks = (uint64_t)(stride_words / kN);
        if ((blocks & (blocks - 1)) == 0) fam_k = __builtin_ctzll(blocks);
    }
    Poly g = fam_k >= 0 ? family_member(fam_k) : powmod_x((uint64_t)stride_words), t;
    for (int r = 0; r < n_rounds; ++r) {
        if (fam_k >= 0 && r > 0) g = family_member(fam_k + r);
        uint32_t* row = polys_host + (size_t)r * kPosRow;
        uint32_t n = 0;
        for (int i = 0; i < kModDeg; ++i)
            if (pbit(g, i)) row[++n] = (uint32_t)i;
        row[0] = n;
        if (fam_k < 0 && r + 1 < n_rounds) {
            mulmod(g, g, t);
            g.swap(t);
        }
    }
    return N2V_OK;
}

// states: DEVICE uint32[n_streams][624] with states[0] filled in; on return states[k] is the window advanced
// by k * stride_words outputs, for every k < n_streams.  polys: DEVICE copy of the n_rounds >= ceil(log2
// (n_streams)) polynomials above.  Round r computes streams [2^r, 2^(r+1)) from streams [0, 2^r): one launch
// per round, every new stream in its own workgroup.
extern "C" int n2v_mt19937_jump_device(uint32_t* states, int32_t n_streams, const uint32_t* polys, int32_t n_rounds,
                                       void* stream) {
    if (!states || n_streams < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_mt19937_jump_device: bad argument");
    int need = 0;
    while ((1 << need) < n_streams) ++need;
    if (need == 0) return N2V_OK;
    if (!polys || n_rounds < need)
        return n2v::fail(N2V_ERR_INVALID, "n2v_mt19937_jump_device: %d streams need %d polynomials, got %d", (int)n_streams,
                         need, (int)n_rounds);
    const size_t lds = sizeof(uint32_t) * kSeqWords + sizeof(uint16_t) * kPosRow;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt_jump_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return n2v::fail(N2V_ERR_HIP, "n2v_mt19937_jump_device: %s", hipGetErrorString(e));
        attr_set = true;
    }
    for (int r = 0; r < need; ++r) {
        const int src = 1 << r;
        const int n_new = n_streams - src < src ? n_streams - src : src;
        hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)n_new), dim3(kJumpThreads), lds, (hipStream_t)stream, states,
                           polys + (size_t)r * kPosRow, src, n_new);
    }
    return n2v::check_launch("n2v_mt19937_jump_device");
}

static int fill_common(const char* what, const uint32_t* states, int32_t n_streams, int32_t pos, int64_t words_per_stream,
                       int64_t n_doubles, int32_t pairs_per_walk, double* out, uint32_t* final_state, void* stream) {
    if (!states || n_streams < 1 || pos < 0 || pos > kN || words_per_stream < 2 || (words_per_stream & 1) ||
        n_doubles < 0 || (int64_t)n_streams * words_per_stream < 2 * n_doubles)
        return n2v::fail(N2V_ERR_INVALID, "%s: bad argument (streams %d, pos %d, stride %lld, n %lld)", what,
                         (int)n_streams, (int)pos, (long long)words_per_stream, (long long)n_doubles);
    if (n_doubles == 0) return N2V_OK;
    if (!out) return n2v::fail(N2V_ERR_INVALID, "%s: null output", what);
    if (pairs_per_walk == 0) {
        hipLaunchKernelGGL(mt_fill_kernel<0>, dim3((unsigned)n_streams), dim3(64), kFillLdsHead, (hipStream_t)stream, states,
                           (int)pos, words_per_stream, 2 * n_doubles, out, final_state, TileMap{0u, 0u, 0u}, 0);
        return n2v::check_launch(what);
    }
    // the in-stream split works on 32-bit offsets: a stream's doubles plus one segment must stay below 2^32
    if (pairs_per_walk < 1 || pairs_per_walk > (1 << 24) || words_per_stream / 2 + 2 * (int64_t)pairs_per_walk + 2 >= (1LL << 32))
        return n2v::fail(N2V_ERR_INVALID, "%s: pairs_per_walk %d / stride %lld out of range", what, (int)pairs_per_walk,
                         (long long)words_per_stream);
    if (((uintptr_t)out & 63) != 0) return n2v::fail(N2V_ERR_INVALID, "%s: output not 64-byte aligned", what);
    const uint32_t dpw = 2u * (uint32_t)pairs_per_walk;
    const TileMap tm{dpw, (uint32_t)pairs_per_walk, (uint32_t)((1ULL << 32) / dpw)};
    // ring: one group of four segments (4 * dpw doubles) plus one block of fresh doubles (312) and the straddling one
    int ring_log2 = 0;
    while ((1LL << ring_log2) < 4LL * dpw + 320) ++ring_log2;
    if (ring_log2 <= 12) {                                // <= 32 KiB of LDS per wavefront (L <= 473)
        hipLaunchKernelGGL(mt_fill_kernel<2>, dim3((unsigned)n_streams), dim3(64), kFillLdsHead + (sizeof(double) << ring_log2),
                           (hipStream_t)stream, states, (int)pos, words_per_stream, 2 * n_doubles, out, final_state, tm,
                           ring_log2);
    } else {
        hipLaunchKernelGGL(mt_fill_kernel<1>, dim3((unsigned)n_streams), dim3(64), kFillLdsHead, (hipStream_t)stream, states,
                           (int)pos, words_per_stream, 2 * n_doubles, out, final_state, tm, 0);
    }
    return n2v::check_launch(what);
}

extern "C" int n2v_mt19937_fill(const uint32_t* states, int32_t n_streams, int32_t pos, int64_t words_per_stream,
                                int64_t n_doubles, double* out, uint32_t* final_state, void* stream) {
    return fill_common("n2v_mt19937_fill", states, n_streams, pos, words_per_stream, n_doubles, 0, out, final_state, stream);
}

extern "C" int n2v_mt19937_fill_tiled(const uint32_t* states, int32_t n_streams, int32_t pos, int64_t words_per_stream,
                                      int64_t n_doubles, int32_t pairs_per_walk, double* out, uint32_t* final_state,
                                      void* stream) {
    if (pairs_per_walk < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_mt19937_fill_tiled: pairs_per_walk %d", (int)pairs_per_walk);
    return fill_common("n2v_mt19937_fill_tiled", states, n_streams, pos, words_per_stream, n_doubles, pairs_per_walk, out,
                       final_state, stream);
}
