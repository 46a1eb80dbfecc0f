// Host entropy coding of libcae_hip.so: quantised-CDF construction and the rANS64 range coder
// with 4-bit bypass escape, bit-compatible with the byte streams the reference writes through
// compressai (`EntropyBottleneck.compress/decompress`, reference call sites
// src/models/tasks/_autoencoders.py:549-551 and :568-571; format in SURVEY.md Appendix A.3).
//
// Design (differs from the upstream two-pass coder on purpose):
//   * the encoder walks the symbols of a stream BACKWARDS once and writes 32-bit words backwards
//     into an exactly pre-sized buffer -- no intermediate symbol stack;
//   * x / freq is an exact fixed-point reciprocal multiply (Alverson) with per-(row, value)
//     constants built once per model in EntropyTables::build_tables();
//   * the decoder finds the symbol by binary search in the CDF row;
//   * streams (one per tile) are independent and coded on a pool of host threads.
#include "cae_hip.h"
#include "cae_internal.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include <sched.h>

namespace cae {

static constexpr uint64_t kRansL = 1ull << 31;
static constexpr uint32_t kPrecision = 16;
static constexpr uint32_t kBypassBits = 4;
static constexpr uint32_t kMaxBypass = (1u << kBypassBits) - 1;
#ifndef CAE_CODER_LOCKSTEP
#define CAE_CODER_LOCKSTEP 2
#endif
static constexpr int kLockMax = 4;  // widest lockstep group built (streams one thread codes together)

static inline uint64_t mul_hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

void EntropyTables::build_tables() {
    // decoder: bucket table over the top kLutBits of the 16-bit cumulative frequency
    lut.assign((size_t)channels << kLutBits, 0);
    for (int c = 0; c < channels; ++c) {
        const int32_t *row = cdf.data() + (size_t)c * stride;
        const int n = len[c];
        int v = 0;
        for (int b = 0; b < (1 << kLutBits); ++b) {
            const int32_t cum = b << (16 - kLutBits);
            while (v + 1 < n - 1 && row[v + 1] <= cum) ++v;
            lut[((size_t)c << kLutBits) + b] = (uint16_t)v;
        }
    }
    enc.assign((size_t)channels * stride, EncSym{0, 0, 0, 0, 0});
    for (int c = 0; c < channels; ++c) {
        const int32_t *row = cdf.data() + (size_t)c * stride;
        for (int v = 0; v + 1 < len[c]; ++v) {
            const uint32_t start = (uint32_t)row[v] & 0xFFFFu;
            const uint32_t freq = (uint32_t)(row[v + 1] - row[v]) & 0xFFFFu;  // uint16 fields upstream
            EncSym &s = enc[(size_t)c * stride + v];
            s.freq = freq;
            s.cmpl_freq = (1u << kPrecision) - freq;
            if (freq < 2) {
                // q = mul_hi(x, 2^64-1) = x - 1  ->  x*M + start = bias + x + (x-1)(M-1)
                s.rcp_freq = ~0ull;
                s.rcp_shift = 0;
                s.bias = start + (1u << kPrecision) - 1;
            } else {
                uint32_t shift = 0;
                while (freq > (1u << shift)) ++shift;
                // ceil(2^(shift+63) / freq) by two 64-bit divides
                uint64_t x0 = freq - 1;
                const uint64_t x1 = 1ull << (shift + 31);
                const uint64_t t1 = x1 / freq;
                x0 += (x1 % freq) << 32;
                const uint64_t t0 = x0 / freq;
                s.rcp_freq = t0 + (t1 << 32);
                s.rcp_shift = shift - 1;
                s.bias = start;
            }
        }
    }
}

namespace {

struct BackWriter {
    uint32_t *ptr;
    inline void put(uint32_t w) { *--ptr = w; }
};

inline void put_bits(uint64_t &x, BackWriter &w, uint32_t val) {
    // Rans64EncPutBits(val, 4): freq = 1 << 12
    const uint64_t x_max = ((kRansL >> 16) << 32) * (uint64_t)(1u << (16 - kBypassBits));
    if (x >= x_max) {
        w.put((uint32_t)x);
        x >>= 32;
    }
    x = (x << kBypassBits) | val;
}

inline int bypass_digits(uint32_t raw) {
    int n = 0;
    while ((raw >> (n * kBypassBits)) != 0) ++n;
    return n;
}

// number of coder steps symbol (value - offset) costs beyond its own
inline void classify(int32_t sym, int32_t offset, int32_t max_value, int32_t &value, uint32_t &raw, bool &esc) {
    value = sym - offset;
    raw = 0;
    esc = false;
    if (value < 0) {
        raw = (uint32_t)(-2 * value - 1);
        value = max_value;
        esc = true;
    } else if (value >= max_value) {
        raw = (uint32_t)(2 * (value - max_value));
        value = max_value;
        esc = true;
    }
}

// ---- NS streams coded in lockstep by one thread ---------------------------------------------------
// A range coder is one long dependency chain (state -> state); two independent streams walked
// together let the out-of-order core overlap the two chains (~1.6x symbols per core-second).

// exact number of coder steps of one stream (one 32-bit word can be emitted per step)
static int count_steps(const EntropyTables &T, const int32_t *symbols, int hw, size_t *steps_out) {
    size_t steps = 0;
    for (int c = 0; c < T.channels; ++c) {
        const int32_t off = T.off[c], maxv = T.len[c] - 2;
        const int32_t *s = symbols + (size_t)c * hw;
        size_t esc_steps = 0;
        unsigned nesc = 0;  // branch-free (vectorisable) scan; escapes are rare
        for (int i = 0; i < hw; ++i) nesc += (unsigned)(s[i] - off) >= (unsigned)maxv;
        for (int i = 0; nesc && i < hw; ++i) {
            const int32_t value = s[i] - off;
            if (value < 0 || value >= maxv) {
                int32_t v;
                uint32_t raw;
                bool esc;
                classify(s[i], off, maxv, v, raw, esc);
                // the escape code carries raw in 4-bit digits counted by a 32-bit shift loop upstream:
                // values with raw >= 2^28 are not representable (upstream shifts by 32 there)
                if (raw >= (1u << 28) || s[i] > (1 << 27) || s[i] < -(1 << 27))
                    return fail(CAE_ERR_ARG, "symbol %d (channel %d) is outside the codable range", s[i], c);
                const int nb = bypass_digits(raw);
                esc_steps += (size_t)(nb / (int)kMaxBypass) + 1 + nb;
            }
        }
        steps += (size_t)hw + esc_steps;
    }
    *steps_out = steps;
    return CAE_OK;
}

// symbols[k]: (channels, hw) int32 in (c, y, x) raster order
// `headroom`: bytes left free in front of every returned stream (the codec's 16-byte chunk header)
template <int NS>
int encode_streams(const EntropyTables &T, const int32_t *const *symbols, int hw, uint8_t **out, size_t *out_len,
                   size_t headroom = 0) {
    uint32_t *buf[NS] = {};
    size_t cap[NS];
    BackWriter w[NS];
    uint64_t x[NS];
    for (int k = 0; k < NS; ++k) {
        size_t steps;
        int rc = count_steps(T, symbols[k], hw, &steps);
        if (rc == CAE_OK) {
            cap[k] = steps + 2;
            buf[k] = (uint32_t *)malloc(cap[k] * sizeof(uint32_t));
            if (!buf[k]) rc = fail(CAE_ERR_NOMEM, "out of memory (%zu words)", cap[k]);
        }
        if (rc != CAE_OK) {
            for (int j = 0; j < k; ++j) free(buf[j]);
            return rc;
        }
        w[k].ptr = buf[k] + cap[k];
        x[k] = kRansL;
    }
    // last symbol first; within a symbol the upstream stack order is
    //   [main][prefix 15..][prefix rem][digit 0 .. digit n-1]  ->  popped in reverse
    for (int c = T.channels - 1; c >= 0; --c) {
        const int32_t off = T.off[c], maxv = T.len[c] - 2;
        const EntropyTables::EncSym *es = T.enc.data() + (size_t)c * T.stride;
        const int32_t *s[NS];
        for (int k = 0; k < NS; ++k) s[k] = symbols[k] + (size_t)c * hw;
        auto step = [&](uint64_t &xs, uint32_t *&wp, int32_t sym) __attribute__((always_inline)) {
            int32_t v = sym - off;
            if (__builtin_expect((unsigned)v >= (unsigned)maxv, 0)) {
                uint32_t raw;
                bool esc;
                BackWriter bw{wp};
                classify(sym, off, maxv, v, raw, esc);
                const int nb = bypass_digits(raw);
                for (int j = nb - 1; j >= 0; --j) put_bits(xs, bw, (raw >> (j * kBypassBits)) & kMaxBypass);
                put_bits(xs, bw, (uint32_t)(nb % (int)kMaxBypass));
                for (int q = 0; q < nb / (int)kMaxBypass; ++q) put_bits(xs, bw, kMaxBypass);
                wp = bw.ptr;
            }
            const EntropyTables::EncSym &e = es[v];
            const uint64_t x_max = ((kRansL >> kPrecision) << 32) * (uint64_t)e.freq;
            uint64_t xx = xs;
            if (xx >= x_max) {
                *--wp = (uint32_t)xx;
                xx >>= 32;
            }
            const uint64_t q = mul_hi(xx, e.rcp_freq) >> e.rcp_shift;
            xs = xx + e.bias + q * e.cmpl_freq;
        };
        {
            // NS independent dependency chains per loop iteration (locals, so they stay in registers)
            uint64_t xl[NS];
            uint32_t *pl[NS];
            for (int k = 0; k < NS; ++k) {
                xl[k] = x[k];
                pl[k] = w[k].ptr;
            }
            for (int i = hw - 1; i >= 0; --i) {
#pragma GCC unroll 8
                for (int k = 0; k < NS; ++k) step(xl[k], pl[k], s[k][i]);
            }
            for (int k = 0; k < NS; ++k) {
                x[k] = xl[k];
                w[k].ptr = pl[k];
            }
        }
    }
    int rc = CAE_OK;
    for (int k = 0; k < NS; ++k) {
        w[k].put((uint32_t)(x[k] >> 32));
        w[k].put((uint32_t)x[k]);
        const size_t nbytes = (size_t)((buf[k] + cap[k]) - w[k].ptr) * sizeof(uint32_t);
        uint8_t *res = (uint8_t *)malloc(headroom + nbytes ? headroom + nbytes : 1);
        if (!res) {
            rc = fail(CAE_ERR_NOMEM, "out of memory (%zu bytes)", headroom + nbytes);
        } else {
            memcpy(res + headroom, w[k].ptr, nbytes);
            out[k] = res;
            out_len[k] = headroom + nbytes;
        }
        free(buf[k]);
    }
    return rc;
}

struct Reader {
    const uint8_t *p, *end;
    bool bad = false;
    inline uint32_t next() {
        if (__builtin_expect(p + 4 > end, 0)) {
            bad = true;
            return 0;
        }
        uint32_t w;
        memcpy(&w, p, 4);
        p += 4;
        return w;
    }
};

inline uint32_t get_bits(uint64_t &x, Reader &r) {
    const uint32_t val = (uint32_t)(x & kMaxBypass);
    x >>= kBypassBits;
    if (x < kRansL) x = (x << 32) | r.next();
    return val;
}

template <int NS>
int decode_streams(const EntropyTables &T, const uint8_t *const *bufs, const size_t *lens, int hw,
                   int32_t *const *symbols) {
    Reader r[NS];
    uint64_t x[NS];
    for (int k = 0; k < NS; ++k) {
        r[k] = Reader{bufs[k], bufs[k] + lens[k]};
        x[k] = r[k].next();
        x[k] |= (uint64_t)r[k].next() << 32;
        if (r[k].bad) return fail(CAE_ERR_CORRUPT, "bitstream shorter than the 8-byte coder state");
    }
    for (int c = 0; c < T.channels; ++c) {
        const int32_t *row = T.cdf.data() + (size_t)c * T.stride;
        const int32_t n = T.len[c], maxv = n - 2, off = T.off[c];
        const uint16_t *lutc = T.lut.data() + ((size_t)c << EntropyTables::kLutBits);
        int32_t *s[NS];
        for (int k = 0; k < NS; ++k) s[k] = symbols[k] + (size_t)c * hw;
        for (int i = 0; i < hw; ++i) {
#pragma GCC unroll 4
            for (int k = 0; k < NS; ++k) {
                uint64_t xx = x[k];
                const uint32_t cum = (uint32_t)(xx & 0xFFFFu);
                // largest v with row[v] <= cum: start at the bucket's first symbol, scan forward
                int32_t v = lutc[cum >> (16 - EntropyTables::kLutBits)];
                while (v < maxv && (uint32_t)row[v + 1] <= cum) ++v;
                const uint32_t start = (uint32_t)row[v], freq = (uint32_t)(row[v + 1] - row[v]);
                xx = (uint64_t)freq * (xx >> kPrecision) + (xx & 0xFFFFu) - start;
                if (xx < kRansL) xx = (xx << 32) | r[k].next();
                if (__builtin_expect(v == maxv, 0)) {
                    int32_t val = (int32_t)get_bits(xx, r[k]);
                    int32_t nb = val;
                    while (val == (int32_t)kMaxBypass && !r[k].bad) {
                        val = (int32_t)get_bits(xx, r[k]);
                        nb += val;
                    }
                    // a 32-bit escape value has at most 8 four-bit digits: a larger count can only come from a damaged
                    // stream (and would shift by 32 or more below: found by UBSan, tests/test_sanitize.py)
                    if (nb > 8) {
                        r[k].bad = true;
                        nb = 0;
                    }
                    uint32_t raw = 0;
                    for (int j = 0; j < nb && !r[k].bad; ++j) {
                        val = (int32_t)get_bits(xx, r[k]);
                        raw |= (uint32_t)val << (j * kBypassBits);
                    }
                    v = (int32_t)(raw >> 1);
                    if (raw & 1)
                        v = -v - 1;
                    else
                        v = (int32_t)((uint32_t)v + (uint32_t)maxv);  // (wraps instead of overflowing on damaged input)
                }
                x[k] = xx;
                s[k][i] = (int32_t)((uint32_t)v + (uint32_t)off);
            }
            bool bad = false;
            for (int k = 0; k < NS; ++k) bad |= r[k].bad;
            if (__builtin_expect(bad, 0))
                return fail(CAE_ERR_CORRUPT, "bitstream ran past its end (channel %d, element %d)", c, i);
        }
    }
    return CAE_OK;
}

// CPUs this process may keep busy: its affinity mask, capped by the cgroup CPU quota (a container's `cpu.max`; running
// more threads than the quota allows gets the WHOLE process throttled for the rest of each scheduler period -- on the
// 16-CPU share of a one-GPU box 2 x 16 coder threads stalled the thread that feeds the GPU), divided among the ranks
// of the node (torch.distributed.run exports LOCAL_WORLD_SIZE).
int cpu_budget() {
    static const int cached = [] {
        int ncpu = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) ncpu = std::min(ncpu > 0 ? ncpu : 1 << 20, CPU_COUNT(&set));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
            char q[64] = {0};
            long period = 0;
            if (fscanf(f, "%63s %ld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) {
                const long quota = atol(q);
                if (quota > 0) ncpu = std::min<long>(ncpu, std::max<long>(1, (quota + period - 1) / period));
            }
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
            long quota = -1, period = 100000;
            if (fscanf(g, "%ld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%ld", &period) != 1) period = 100000;
                fclose(h);
            }
            if (quota > 0 && period > 0) ncpu = std::min<long>(ncpu, std::max<long>(1, (quota + period - 1) / period));
        }
        int local_world = 1;
        if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) local_world = std::max(1, std::atoi(e));
        return std::max(1, ncpu / local_world);
    }();
    return cached;
}

// Default size of one coder pool: CAE_CODER_THREADS, else the CPU budget, capped at 16.  (The pipelined drivers, which
// run an encode and a decode pool side by side, split the budget themselves: slide.SlideCoder.)
int default_threads() {
    static const int cached = [] {
        if (const char *e = std::getenv("CAE_CODER_THREADS")) {
            const int v = std::atoi(e);
            if (v > 0) return v;
        }
        return std::max(1, std::min(cpu_budget(), 16));
    }();
    return cached;
}

// Streams one coder thread walks in lockstep: 2 or 4 independent dependency chains per loop iteration.  Four chains use
// fewer CPU-seconds per symbol (measured: the pipelined round trip at the same tiles/s with 11.2 instead of 13.2 busy CPUs)
// but a batch then has half as many work items to spread over the pool, so the wider group is the choice when CPUs, not
// work items, are short: CAE_CODER_LOCKSTEP = 2 | 4, else 4 when this process may keep fewer than 16 CPUs busy.
int lockstep_width() {
    static const int cached = [] {
        if (const char *e = std::getenv("CAE_CODER_LOCKSTEP")) {
            const int v = std::atoi(e);
            if (v == 1 || v == 2 || v == 4) return v;
        }
        return cpu_budget() < 16 ? 4 : CAE_CODER_LOCKSTEP;
    }();
    return cached;
}

template <class F>
int parallel_streams(int n, int threads, F f) {
    if (threads <= 0) threads = default_threads();
    threads = std::max(1, std::min(threads, n));
    std::atomic<int> next{0};
    std::atomic<int> rc{0};
    std::string first_err;
    std::mutex err_mu;
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            const int r = f(i);
            if (r != 0) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (rc.load() == 0) {
                    rc.store(r);
                    first_err = cae_last_error();
                }
            }
        }
    };
    if (threads == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (rc.load() != 0) return fail(rc.load(), "%s", first_err.c_str());
    return CAE_OK;
}

}  // namespace

// one / two streams coded by the calling thread (the codec front door: cae_door.hip)
int rans_encode_chunk(const EntropyTables &T, const int32_t *symbols, int hw, size_t headroom, uint8_t **out,
                      size_t *out_len) {
    const int32_t *sy[1] = {symbols};
    return encode_streams<1>(T, sy, hw, out, out_len, headroom);
}

int rans_encode_chunk_pair(const EntropyTables &T, const int32_t *const *symbols, int hw, size_t headroom, uint8_t **out,
                           size_t *out_len) {
    return encode_streams<2>(T, symbols, hw, out, out_len, headroom);
}

int rans_decode_chunk(const EntropyTables &T, const uint8_t *buf, size_t len, int hw, int32_t *symbols) {
    const uint8_t *b[1] = {buf};
    const size_t l[1] = {len};
    int32_t *sy[1] = {symbols};
    return decode_streams<1>(T, b, l, hw, sy);
}

int rans_decode_chunk_pair(const EntropyTables &T, const uint8_t *const *bufs, const size_t *lens, int hw,
                           int32_t *const *symbols) {
    return decode_streams<2>(T, bufs, lens, hw, symbols);
}

}  // namespace cae

using namespace cae;

extern "C" {

int cae_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
    if (!pmf || !cdf || n < 1) return fail(CAE_ERR_ARG, "NULL or empty pmf");
    if (precision < 1 || precision > 16) return fail(CAE_ERR_ARG, "precision %d out of range", precision);
    for (int i = 0; i < n; ++i)
        if (!(pmf[i] >= 0.0f) || !std::isfinite(pmf[i]))
            return fail(CAE_ERR_ARG, "Invalid `pmf`, non-finite or negative element found");
    const uint32_t top = 1u << precision;
    // quantise, then rescale so the frequencies sum to at most 2^precision
    std::vector<uint32_t> f(n);
    uint32_t total = 0;
    for (int i = 0; i < n; ++i) {
        f[i] = (uint32_t)std::round(pmf[i] * (float)top);
        total += f[i];
    }
    if (total == 0) return fail(CAE_ERR_ARG, "Invalid `pmf`: at least one element must have a non-zero probability.");
    cdf[0] = 0;
    uint32_t run = 0;
    for (int i = 0; i < n; ++i) {
        run += (uint32_t)(((uint64_t)top * f[i]) / total);
        cdf[i + 1] = run;
    }
    cdf[n] = top;
    // repair zero-width bins by stealing one count from the narrowest bin wider than 1
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best = ~0u;
        int steal = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t w = cdf[j + 1] - cdf[j];
            if (w > 1 && w < best) {
                best = w;
                steal = j;
            }
        }
        if (steal < 0) return fail(CAE_ERR_ARG, "pmf has more symbols than 2^precision can resolve");
        if (steal < i)
            for (int j = steal + 1; j <= i; ++j) cdf[j]--;
        else
            for (int j = i + 1; j <= steal; ++j) cdf[j]++;
    }
    return CAE_OK;
}

int cae_cpu_budget(void) { return cpu_budget(); }

int cae_coder_lockstep(void) { return lockstep_width(); }

int cae_coder_threads(int requested, int n_streams) {
    int t = requested > 0 ? requested : default_threads();
    if (n_streams > 0) t = std::min(t, n_streams);
    return std::max(1, t);
}

int cae_rans_encode_batch(cae_model_t *mm, const int32_t *symbols, int n_streams, int hw, uint8_t **out_bufs,
                          size_t *out_lens, int threads) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !symbols || !out_bufs || !out_lens) return fail(CAE_ERR_ARG, "NULL argument");
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (n_streams < 1 || hw < 0) return fail(CAE_ERR_ARG, "bad shape");
    for (int i = 0; i < n_streams; ++i) {
        out_bufs[i] = nullptr;
        out_lens[i] = 0;
    }
    const EntropyTables &T = m->ent;
    const size_t per = (size_t)T.channels * hw;
    // work item = `lock` streams coded in lockstep (the remainder in smaller groups)
    const int lock = lockstep_width();
    const int items = (n_streams + lock - 1) / lock;
    int rc = parallel_streams(items, threads, [&](int it) {
        int i = lock * it;
        const int end = std::min(n_streams, i + lock);
        int r = CAE_OK;
        const int32_t *sy[kLockMax];
        for (; r == CAE_OK && i + 3 < end; i += 4) {
            for (int k = 0; k < 4; ++k) sy[k] = symbols + per * (i + k);
            r = encode_streams<4>(T, sy, hw, &out_bufs[i], &out_lens[i]);
        }
        for (; r == CAE_OK && i + 1 < end; i += 2) {
            sy[0] = symbols + per * i;
            sy[1] = symbols + per * (i + 1);
            r = encode_streams<2>(T, sy, hw, &out_bufs[i], &out_lens[i]);
        }
        if (r == CAE_OK && i < end) {
            sy[0] = symbols + per * i;
            r = encode_streams<1>(T, sy, hw, &out_bufs[i], &out_lens[i]);
        }
        return r;
    });
    if (rc != 0)
        for (int i = 0; i < n_streams; ++i) {
            free(out_bufs[i]);
            out_bufs[i] = nullptr;
        }
    return rc;
}

int cae_rans_encode_packed(cae_model_t *mm, const int32_t *symbols, int n_streams, int hw, uint8_t **out_buf,
                           size_t *offsets, int threads) {
    if (!out_buf || !offsets) return fail(CAE_ERR_ARG, "NULL argument");
    *out_buf = nullptr;
    std::vector<uint8_t *> bufs((size_t)std::max(n_streams, 1), nullptr);
    std::vector<size_t> lens((size_t)std::max(n_streams, 1), 0);
    int rc = cae_rans_encode_batch(mm, symbols, n_streams, hw, bufs.data(), lens.data(), threads);
    if (rc) return rc;
    offsets[0] = 0;
    for (int i = 0; i < n_streams; ++i) offsets[i + 1] = offsets[i] + lens[i];
    uint8_t *packed = (uint8_t *)malloc(offsets[n_streams] ? offsets[n_streams] : 1);
    if (!packed) rc = fail(CAE_ERR_NOMEM, "out of memory (%zu bytes)", offsets[n_streams]);
    if (rc == CAE_OK)
        rc = parallel_streams(n_streams, threads, [&](int i) {
            memcpy(packed + offsets[i], bufs[i], lens[i]);
            return (int)CAE_OK;
        });
    for (int i = 0; i < n_streams; ++i) free(bufs[i]);
    if (rc) {
        free(packed);
        return rc;
    }
    *out_buf = packed;
    return CAE_OK;
}

int cae_rans_decode_batch(cae_model_t *mm, const uint8_t *const *bufs, const size_t *lens, int n_streams, int hw,
                          int32_t *symbols, int threads) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !symbols || !bufs || !lens) return fail(CAE_ERR_ARG, "NULL argument");
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (n_streams < 1 || hw < 0) return fail(CAE_ERR_ARG, "bad shape");
    const EntropyTables &T = m->ent;
    const size_t per = (size_t)T.channels * hw;
    const int lock = lockstep_width();
    const int items = (n_streams + lock - 1) / lock;
    return parallel_streams(items, threads, [&](int it) {
        int i = lock * it;
        const int end = std::min(n_streams, i + lock);
        int r = CAE_OK;
        int32_t *sy[kLockMax];
        for (; r == CAE_OK && i + 3 < end; i += 4) {
            for (int k = 0; k < 4; ++k) sy[k] = symbols + per * (i + k);
            r = decode_streams<4>(T, bufs + i, lens + i, hw, sy);
        }
        for (; r == CAE_OK && i + 1 < end; i += 2) {
            sy[0] = symbols + per * i;
            sy[1] = symbols + per * (i + 1);
            r = decode_streams<2>(T, bufs + i, lens + i, hw, sy);
        }
        if (r == CAE_OK && i < end) {
            sy[0] = symbols + per * i;
            r = decode_streams<1>(T, bufs + i, lens + i, hw, sy);
        }
        return r;
    });
}

}  // extern "C"
