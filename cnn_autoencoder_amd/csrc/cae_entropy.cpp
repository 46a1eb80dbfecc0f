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
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

namespace cae {

static constexpr uint64_t kRansL = 1ull << 31;
static constexpr uint32_t kPrecision = 16;
static constexpr uint32_t kBypassBits = 4;
static constexpr uint32_t kMaxBypass = (1u << kBypassBits) - 1;

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

// symbols: (channels, hw) int32 in (c, y, x) raster order
int encode_stream(const EntropyTables &T, const int32_t *symbols, int hw, uint8_t **out, size_t *out_len) {
    // pass 1: exact upper bound on emitted words (one per coder step) + 2 flush words
    size_t steps = 0;
    for (int c = 0; c < T.channels; ++c) {
        const int32_t off = T.off[c], maxv = T.len[c] - 2;
        const int32_t *s = symbols + (size_t)c * hw;
        for (int i = 0; i < hw; ++i) {
            int32_t v;
            uint32_t raw;
            bool esc;
            classify(s[i], off, maxv, v, raw, esc);
            steps += 1;
            // the escape code carries raw in 4-bit digits counted by a 32-bit shift loop upstream:
            // values with raw >= 2^28 are not representable (upstream shifts by 32 there)
            if (raw >= (1u << 28) || s[i] > (1 << 27) || s[i] < -(1 << 27))
                return fail(CAE_ERR_ARG, "symbol %d (channel %d) is outside the codable range", s[i], c);
            if (esc) {
                const int nb = bypass_digits(raw);
                steps += (size_t)(nb / (int)kMaxBypass) + 1 + nb;
            }
        }
    }
    const size_t cap = steps + 2;
    uint32_t *buf = (uint32_t *)malloc(cap * sizeof(uint32_t));
    if (!buf) return fail(CAE_ERR_NOMEM, "out of memory (%zu words)", cap);
    BackWriter w{buf + cap};
    uint64_t x = kRansL;

    // pass 2: last symbol first; within a symbol the upstream stack order is
    //   [main][prefix 15..][prefix rem][digit 0 .. digit n-1]  ->  popped in reverse
    for (int c = T.channels - 1; c >= 0; --c) {
        const int32_t off = T.off[c], maxv = T.len[c] - 2;
        const int32_t *s = symbols + (size_t)c * hw;
        const EntropyTables::EncSym *es = T.enc.data() + (size_t)c * T.stride;
        for (int i = hw - 1; i >= 0; --i) {
            int32_t v;
            uint32_t raw;
            bool esc;
            classify(s[i], off, maxv, v, raw, esc);
            if (esc) {
                const int nb = bypass_digits(raw);
                for (int j = nb - 1; j >= 0; --j) put_bits(x, w, (raw >> (j * kBypassBits)) & kMaxBypass);
                put_bits(x, w, (uint32_t)(nb % (int)kMaxBypass));
                for (int k = 0; k < nb / (int)kMaxBypass; ++k) put_bits(x, w, kMaxBypass);
            }
            const EntropyTables::EncSym &e = es[v];
            const uint64_t x_max = ((kRansL >> kPrecision) << 32) * (uint64_t)e.freq;
            if (x >= x_max) {
                w.put((uint32_t)x);
                x >>= 32;
            }
            const uint64_t q = mul_hi(x, e.rcp_freq) >> e.rcp_shift;
            x = x + e.bias + q * e.cmpl_freq;
        }
    }
    w.put((uint32_t)(x >> 32));
    w.put((uint32_t)x);
    const size_t nbytes = (size_t)((buf + cap) - w.ptr) * sizeof(uint32_t);
    uint8_t *res = (uint8_t *)malloc(nbytes ? nbytes : 1);
    if (!res) {
        free(buf);
        return fail(CAE_ERR_NOMEM, "out of memory (%zu bytes)", nbytes);
    }
    memcpy(res, w.ptr, nbytes);
    free(buf);
    *out = res;
    *out_len = nbytes;
    return CAE_OK;
}

struct Reader {
    const uint8_t *p, *end;
    bool bad = false;
    inline uint32_t next() {
        if (p + 4 > end) {
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

int decode_stream(const EntropyTables &T, const uint8_t *buf, size_t len, int hw, int32_t *symbols) {
    Reader r{buf, buf + len};
    uint64_t x = r.next();
    x |= (uint64_t)r.next() << 32;
    if (r.bad) return fail(CAE_ERR_CORRUPT, "bitstream shorter than the 8-byte coder state");
    for (int c = 0; c < T.channels; ++c) {
        const int32_t *row = T.cdf.data() + (size_t)c * T.stride;
        const int32_t n = T.len[c], maxv = n - 2, off = T.off[c];
        const uint16_t *lutc = T.lut.data() + ((size_t)c << EntropyTables::kLutBits);
        int32_t *s = symbols + (size_t)c * hw;
        for (int i = 0; i < hw; ++i) {
            const uint32_t cum = (uint32_t)(x & 0xFFFFu);
            // largest v with row[v] <= cum: start at the bucket's first symbol, scan forward
            int32_t v = lutc[cum >> (16 - EntropyTables::kLutBits)];
            while (v < maxv && (uint32_t)row[v + 1] <= cum) ++v;
            const uint32_t start = (uint32_t)row[v], freq = (uint32_t)(row[v + 1] - row[v]);
            x = (uint64_t)freq * (x >> kPrecision) + (x & 0xFFFFu) - start;
            if (x < kRansL) x = (x << 32) | r.next();
            if (v == maxv) {
                int32_t val = (int32_t)get_bits(x, r);
                int32_t nb = val;
                while (val == (int32_t)kMaxBypass && !r.bad) {
                    val = (int32_t)get_bits(x, r);
                    nb += val;
                }
                int32_t raw = 0;
                for (int j = 0; j < nb && !r.bad; ++j) {
                    val = (int32_t)get_bits(x, r);
                    raw |= (int32_t)((uint32_t)val << (j * kBypassBits));
                }
                v = raw >> 1;
                if (raw & 1)
                    v = -v - 1;
                else
                    v += maxv;
            }
            s[i] = v + off;
            if (r.bad) return fail(CAE_ERR_CORRUPT, "bitstream ran past its end (channel %d, element %d)", c, i);
        }
    }
    return CAE_OK;
}

template <class F>
int parallel_streams(int n, int threads, F f) {
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
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
    int rc = parallel_streams(n_streams, threads,
                              [&](int i) { return encode_stream(T, symbols + per * i, hw, &out_bufs[i], &out_lens[i]); });
    if (rc != 0)
        for (int i = 0; i < n_streams; ++i) {
            free(out_bufs[i]);
            out_bufs[i] = nullptr;
        }
    return rc;
}

int cae_rans_decode_batch(cae_model_t *mm, const uint8_t *const *bufs, const size_t *lens, int n_streams, int hw,
                          int32_t *symbols, int threads) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !symbols || !bufs || !lens) return fail(CAE_ERR_ARG, "NULL argument");
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (n_streams < 1 || hw < 0) return fail(CAE_ERR_ARG, "bad shape");
    const EntropyTables &T = m->ent;
    const size_t per = (size_t)T.channels * hw;
    return parallel_streams(n_streams, threads,
                            [&](int i) { return decode_stream(T, bufs[i], lens[i], hw, symbols + per * i); });
}

}  // extern "C"
