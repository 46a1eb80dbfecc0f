/* CPU oracle: plain-C restatement of the compressai entropy-coder primitives.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or called from the product
 * library (cnn_autoencoder_amd/csrc).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so.
 *
 * Restates (third-party dependency `compressai>=1.2.4`, reference requirements.txt:25;
 * source absent from /root/reference and from this image -> PARITY UNPINNED, see
 * oracle/cae_oracle.py header and SURVEY.md Appendix A):
 *   - compressai/cpp_exts/ops/ops.cpp            pmf_to_quantized_cdf
 *   - compressai/cpp_exts/rans/rans_interface.cpp BufferedRansEncoder::encode_with_indexes,
 *                                                 ::flush, RansDecoder::decode_with_indexes
 *   - ryg_rans rans64.h                           Rans64EncPut / EncFlush / DecInit / DecAdvance
 * Reference call sites that reach them: src/models/tasks/_autoencoders.py:502 (update ->
 * pmf_to_quantized_cdf), :549-551 and :645-647 (compress -> encode_with_indexes),
 * :568-571 and :662-665 (decompress -> decode_with_indexes).
 *
 * The structure deliberately mirrors the upstream two-pass form (build the symbol stack in
 * forward order, then pop it in reverse while writing 32-bit words backwards) so that the
 * product's single-pass coder is checked against an independently shaped implementation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RANS64_L (1ull << 31)
#define PRECISION 16u
#define BYPASS_PRECISION 4u
#define MAX_BYPASS_VAL ((1u << BYPASS_PRECISION) - 1u)

typedef struct {
    uint16_t start;
    uint16_t range;
    uint8_t bypass;
} oracle_sym_t;

/* returns 0 ok, -1 invalid pmf, -2 zero total */
int oracle_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf /* n+1 */) {
    for (int i = 0; i < n; ++i) {
        if (pmf[i] < 0 || !isfinite(pmf[i])) return -1;
    }
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) {
        float v = roundf(pmf[i] * (float)(1 << precision));
        cdf[i + 1] = (uint32_t)v;
    }
    int acc = 0;
    for (int i = 0; i <= n; ++i) acc += (int)cdf[i];
    const uint32_t total = (uint32_t)acc;
    if (total == 0) return -2;
    for (int i = 0; i <= n; ++i) {
        cdf[i] = (uint32_t)(((uint64_t)(1 << precision) * cdf[i]) / total);
    }
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;

    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best_steal = -1;
            for (int j = 0; j < n; ++j) {
                uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) {
                    best_freq = freq;
                    best_steal = j;
                }
            }
            if (best_steal == -1) return -3;
            if (best_steal < i) {
                for (int j = best_steal + 1; j <= i; ++j) cdf[j]--;
            } else {
                for (int j = i + 1; j <= best_steal; ++j) cdf[j]++;
            }
        }
    }
    return 0;
}

static void enc_put(uint64_t *r, uint32_t **pptr, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    uint64_t x = *r;
    uint64_t x_max = ((RANS64_L >> scale_bits) << 32) * freq;
    if (x >= x_max) {
        *pptr -= 1;
        **pptr = (uint32_t)x;
        x >>= 32;
    }
    *r = ((x / freq) << scale_bits) + (x % freq) + start;
}

static void enc_put_bits(uint64_t *r, uint32_t **pptr, uint32_t val, uint32_t nbits) {
    uint64_t x = *r;
    uint32_t freq = 1u << (16 - nbits);
    uint64_t x_max = ((RANS64_L >> 16) << 32) * freq;
    if (x >= x_max) {
        *pptr -= 1;
        **pptr = (uint32_t)x;
        x >>= 32;
    }
    *r = (x << nbits) | val;
}

static uint32_t dec_get_bits(uint64_t *r, const uint32_t **pptr, uint32_t nbits) {
    uint64_t x = *r;
    uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < RANS64_L) {
        x = (x << 32) | **pptr;
        *pptr += 1;
    }
    *r = x;
    return val;
}

/* symbols/indexes: n entries.  cdfs: row-major (n_cdf, cdf_stride) int32.
 * out: caller buffer of out_cap bytes.  Returns payload length in bytes, or -1 if out_cap is
 * too small / allocation failed. */
long oracle_rans_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, long n,
                                     const int32_t *cdfs, int cdf_stride,
                                     const int32_t *cdf_lengths, const int32_t *offsets,
                                     uint8_t *out, long out_cap) {
    size_t cap = (size_t)n + 16, cnt = 0;
    oracle_sym_t *syms = (oracle_sym_t *)malloc(cap * sizeof(oracle_sym_t));
    if (!syms) return -1;
#define PUSH(s_, r_, b_)                                                              \
    do {                                                                              \
        if (cnt == cap) {                                                             \
            cap *= 2;                                                                 \
            syms = (oracle_sym_t *)realloc(syms, cap * sizeof(oracle_sym_t));         \
            if (!syms) return -1;                                                     \
        }                                                                             \
        syms[cnt].start = (uint16_t)(s_);                                             \
        syms[cnt].range = (uint16_t)(r_);                                             \
        syms[cnt].bypass = (uint8_t)(b_);                                             \
        cnt++;                                                                        \
    } while (0)

    for (long i = 0; i < n; ++i) {
        const int32_t idx = indexes[i];
        const int32_t *cdf = cdfs + (size_t)idx * cdf_stride;
        const int32_t max_value = cdf_lengths[idx] - 2;
        int32_t value = symbols[i] - offsets[idx];
        uint32_t raw_val = 0;
        if (value < 0) {
            raw_val = (uint32_t)(-2 * value - 1);
            value = max_value;
        } else if (value >= max_value) {
            raw_val = (uint32_t)(2 * (value - max_value));
            value = max_value;
        }
        if (raw_val >= (1u << 28)) { /* upstream shifts a uint32 by 32 here (undefined) */
            free(syms);
            return -2;
        }
        PUSH(cdf[value], cdf[value + 1] - cdf[value], 0);
        if (value == max_value) {
            int32_t n_bypass = 0;
            while ((raw_val >> (n_bypass * BYPASS_PRECISION)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= (int32_t)MAX_BYPASS_VAL) {
                PUSH(MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, 1);
                val -= MAX_BYPASS_VAL;
            }
            PUSH(val, val + 1, 1);
            for (int32_t j = 0; j < n_bypass; ++j) {
                const int32_t v = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL;
                PUSH(v, v + 1, 1);
            }
        }
    }
#undef PUSH

    /* flush */
    size_t nwords = cnt + 2;
    uint32_t *buf = (uint32_t *)malloc(nwords * sizeof(uint32_t));
    if (!buf) {
        free(syms);
        return -1;
    }
    uint32_t *ptr = buf + nwords;
    uint64_t rans = RANS64_L;
    while (cnt > 0) {
        const oracle_sym_t s = syms[cnt - 1];
        if (!s.bypass) {
            enc_put(&rans, &ptr, s.start, s.range, PRECISION);
        } else {
            enc_put_bits(&rans, &ptr, s.start, BYPASS_PRECISION);
        }
        --cnt;
    }
    ptr -= 2;
    ptr[0] = (uint32_t)(rans >> 0);
    ptr[1] = (uint32_t)(rans >> 32);
    long nbytes = (long)((buf + nwords) - ptr) * (long)sizeof(uint32_t);
    long ret = nbytes;
    if (nbytes > out_cap) {
        ret = -1;
    } else {
        memcpy(out, ptr, (size_t)nbytes);
    }
    free(buf);
    free(syms);
    return ret;
}

/* Returns 0 ok. */
int oracle_rans_decode_with_indexes(const uint8_t *encoded, long nbytes, const int32_t *indexes, long n,
                                    const int32_t *cdfs, int cdf_stride,
                                    const int32_t *cdf_lengths, const int32_t *offsets,
                                    int32_t *out) {
    (void)nbytes;
    const uint32_t *ptr = (const uint32_t *)encoded;
    uint64_t rans = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32);
    ptr += 2;
    for (long i = 0; i < n; ++i) {
        const int32_t idx = indexes[i];
        const int32_t *cdf = cdfs + (size_t)idx * cdf_stride;
        const int32_t len = cdf_lengths[idx];
        const int32_t max_value = len - 2;
        const int32_t offset = offsets[idx];
        const uint32_t cum = (uint32_t)(rans & ((1u << PRECISION) - 1));
        int k = 0;
        while (k < len && !((uint32_t)cdf[k] > cum)) ++k;
        const uint32_t s = (uint32_t)(k - 1);
        {
            const uint64_t mask = (1ull << PRECISION) - 1;
            const uint32_t start = (uint32_t)cdf[s];
            const uint32_t freq = (uint32_t)(cdf[s + 1] - cdf[s]);
            uint64_t x = rans;
            x = freq * (x >> PRECISION) + (x & mask) - start;
            if (x < RANS64_L) {
                x = (x << 32) | *ptr;
                ptr += 1;
            }
            rans = x;
        }
        int32_t value = (int32_t)s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_get_bits(&rans, &ptr, BYPASS_PRECISION);
            int32_t n_bypass = val;
            while (val == (int32_t)MAX_BYPASS_VAL) {
                val = (int32_t)dec_get_bits(&rans, &ptr, BYPASS_PRECISION);
                n_bypass += val;
            }
            int32_t raw_val = 0;
            for (int j = 0; j < n_bypass; ++j) {
                val = (int32_t)dec_get_bits(&rans, &ptr, BYPASS_PRECISION);
                raw_val |= val << (j * BYPASS_PRECISION);
            }
            value = raw_val >> 1;
            if (raw_val & 1) {
                value = -value - 1;
            } else {
                value += max_value;
            }
        }
        out[i] = value + offset;
    }
    return 0;
}
