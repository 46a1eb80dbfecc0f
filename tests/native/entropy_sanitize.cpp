// Host range coder under AddressSanitizer / UBSan (CPU build only; GPU sanitizers are not available on the pool).
// Links the product's own cae_entropy.cpp: random CDF tables from cae_pmf_to_quantized_cdf, random symbols including
// values outside the support (bypass coding), every lockstep width (1, 2, 4 streams per thread through the batch entry
// points, the single-chunk entry points of the codec front door), round trips, and damaged / truncated streams, which
// must come back as CAE_ERR_CORRUPT without touching memory they do not own.
// Build + run: tests/test_sanitize.py.
#include "cae_hip.h"
#include "cae_internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace cae {
thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
// (the device half of the handle is not part of this build: a host-only Model never owns device memory)
Model::~Model() {}
}  // namespace cae
extern "C" const char *cae_last_error(void) { return cae::g_err.c_str(); }

using namespace cae;

#define REQUIRE(cond)                                                                  \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, cae_last_error()); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static int run(unsigned seed, int channels, int hw, int n_streams) {
    std::mt19937 rng(seed);
    Model m;
    EntropyTables &T = m.ent;
    std::vector<int> lens(channels);
    int stride = 0;
    for (int c = 0; c < channels; ++c) {
        lens[c] = 3 + (int)(rng() % 40);  // symbols in the support + 1 tail bin + 1
        stride = std::max(stride, lens[c]);
    }
    T.channels = channels;
    T.stride = stride;
    T.cdf.assign((size_t)channels * stride, 0);
    T.len.assign(channels, 0);
    T.off.assign(channels, 0);
    T.medians.assign(channels, 0.0f);
    for (int c = 0; c < channels; ++c) {
        const int n = lens[c] - 1;  // pmf entries incl. the tail mass
        std::vector<float> pmf(n);
        for (auto &p : pmf) p = (float)(1 + rng() % 1000) / 1000.0f;
        if (c % 3 == 0) pmf[rng() % n] = 50.0f;   // a dominant symbol
        if (c % 4 == 1) pmf[rng() % n] = 1e-7f;   // a bin that needs the stealing loop
        std::vector<uint32_t> cdf(n + 1);
        REQUIRE(cae_pmf_to_quantized_cdf(pmf.data(), n, 16, cdf.data()) == 0);
        REQUIRE(cdf[0] == 0 && cdf[n] == 65536);
        for (int i = 0; i < n; ++i) REQUIRE(cdf[i] < cdf[i + 1]);
        for (int i = 0; i <= n; ++i) T.cdf[(size_t)c * stride + i] = (int32_t)cdf[i];
        T.len[c] = n + 1;
        T.off[c] = -(int)(rng() % (unsigned)(n - 1));
    }
    T.build_tables();
    std::vector<int32_t> sym((size_t)n_streams * channels * hw);
    for (int s = 0; s < n_streams; ++s)
        for (int c = 0; c < channels; ++c)
            for (int i = 0; i < hw; ++i) {
                const int maxv = T.len[c] - 2;
                int v = (int)(rng() % (unsigned)maxv) + T.off[c];
                const unsigned r = rng() % 64;
                if (r == 0) v = T.off[c] - 1 - (int)(rng() % 5000);          // below the support: bypass
                if (r == 1) v = T.off[c] + maxv + (int)(rng() % 5000);       // above
                if (r == 2) v = (int)(rng() % 200000000u) - 100000000;       // many bypass digits
                sym[((size_t)s * channels + c) * hw + i] = v;
            }
    cae_model_t *mm = reinterpret_cast<cae_model_t *>(&m);
    for (const char *lock : {"1", "2", "4"}) {
        (void)lock;  // (the width is latched at first use; tests/test_sanitize.py runs the binary once per width)
    }
    std::vector<uint8_t *> bufs(n_streams, nullptr);
    std::vector<size_t> ls(n_streams, 0);
    REQUIRE(cae_rans_encode_batch(mm, sym.data(), n_streams, hw, bufs.data(), ls.data(), 3) == 0);
    std::vector<int32_t> back(sym.size(), 12345);
    REQUIRE(cae_rans_decode_batch(mm, bufs.data(), ls.data(), n_streams, hw, back.data(), 3) == 0);
    REQUIRE(back == sym);
    // packed form
    uint8_t *packed = nullptr;
    std::vector<size_t> offs(n_streams + 1);
    REQUIRE(cae_rans_encode_packed(mm, sym.data(), n_streams, hw, &packed, offs.data(), 2) == 0);
    for (int s = 0; s < n_streams; ++s) {
        REQUIRE(offs[s + 1] - offs[s] == ls[s]);
        REQUIRE(memcmp(packed + offs[s], bufs[s], ls[s]) == 0);
    }
    free(packed);
    // the front door's single-chunk entry points: same bytes behind 16 bytes of headroom
    const size_t per = (size_t)channels * hw;
    for (int s = 0; s < n_streams; ++s) {
        uint8_t *one = nullptr;
        size_t len = 0;
        REQUIRE(rans_encode_chunk(T, sym.data() + s * per, hw, 16, &one, &len) == 0);
        REQUIRE(len == ls[s] + 16 && memcmp(one + 16, bufs[s], ls[s]) == 0);
        std::vector<int32_t> d(per);
        REQUIRE(rans_decode_chunk(T, one + 16, len - 16, hw, d.data()) == 0);
        REQUIRE(memcmp(d.data(), sym.data() + s * per, per * 4) == 0);
        free(one);
    }
    // damaged streams: truncated at every length of the first stream's head and tail, random bit flips; never a crash,
    // truncation is always reported
    {
        std::vector<int32_t> d(per);
        for (size_t cut : {(size_t)0, (size_t)3, (size_t)7, (size_t)8, ls[0] / 2, ls[0] - 4, ls[0] - 1}) {
            if (cut >= ls[0]) continue;
            std::vector<uint8_t> t(bufs[0], bufs[0] + cut);  // exact-size heap copy: a read past the end is an ASan error
            const int rc = rans_decode_chunk(T, t.data(), t.size(), hw, d.data());
            REQUIRE(rc == CAE_ERR_CORRUPT);
        }
        for (int k = 0; k < 50; ++k) {
            std::vector<uint8_t> t(bufs[0], bufs[0] + ls[0]);
            t[rng() % t.size()] ^= (uint8_t)(1u << (rng() % 8));
            (void)rans_decode_chunk(T, t.data(), t.size(), hw, d.data());  // any symbols or CAE_ERR_CORRUPT, no crash
        }
    }
    for (auto *b : bufs) free(b);
    // a symbol whose escape value would need 2^28 or more is refused (upstream shifts a uint32 by 32 there)
    std::vector<int32_t> bad(per, 0);
    bad[per / 2] = 1 << 29;
    uint8_t *one = nullptr;
    size_t len = 0;
    REQUIRE(rans_encode_chunk(T, bad.data(), hw, 0, &one, &len) == CAE_ERR_ARG);
    return 0;
}

int main() {
    const int shapes[][3] = {{1, 1, 1}, {3, 17, 2}, {5, 64, 5}, {7, 33, 9}, {2, 4096, 4}};
    for (unsigned seed = 0; seed < 6; ++seed)
        for (auto &s : shapes)
            if (int rc = run(1000 * seed + (unsigned)s[1], s[0], s[1], s[2])) return rc;
    printf("entropy_sanitize: ok (lockstep %d)\n", cae_coder_lockstep());
    return 0;
}
