// Host side of libcae_hip.so: model handle, weight packing, kernel dispatch (see include/cae_hip.h).
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_kernels.hpp"
#include "cae_kernels_f16.hpp"
#include "cae_launch.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace cae {

thread_local std::string g_last_error;
thread_local int64_t g_last_ticket = 0;  // range ticket of this thread's latest analysis / synthesis call (0: fp32 call)
thread_local int g_force_fp32 = 0;       // cae_thread_force_fp32

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

static int round_ct(int c) {
    const int t = (c + 31) / 32;
    if (t <= 1) return 1;
    if (t <= 2) return 2;
    if (t <= 4) return 4;
    if (t <= 6) return 6;
    return -1;
}

// ---- packing -----------------------------------------------------------------------------------
// weights -> [chunk][ky][kx][ct][lane][j]:  value W(cout = 32ct + (lane&31), cin = 8chunk + 4(lane>>5) + j, ky, kx)
static std::vector<float> pack_weights(const float *w, bool transposed, int cin, int cout, int ks, int ct,
                                       bool flip = false) {
    const int chunks = (cin + 7) / 8;
    std::vector<float> out((size_t)chunks * ks * ks * ct * 256, 0.0f);
    size_t o = 0;
    for (int c = 0; c < chunks; ++c)
        for (int ky = 0; ky < ks; ++ky)
            for (int kx = 0; kx < ks; ++kx)
                for (int t = 0; t < ct; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j, ++o) {
                            const int co = 32 * t + (lane & 31);
                            const int ci = 8 * c + 4 * (lane >> 5) + j;
                            if (co < cout && ci < cin) {
                                // conv: (cout,cin,k,k); transposed conv: (cin,cout,k,k)
                                const int sy = flip ? ks - 1 - ky : ky, sx = flip ? ks - 1 - kx : kx;
                                const size_t idx = transposed ? (((size_t)ci * cout + co) * ks + sy) * ks + sx
                                                              : (((size_t)co * cin + ci) * ks + sy) * ks + sx;
                                out[o] = w[idx];
                            }
                        }
    return out;
}

// gamma -> [jt][co][q][lane][jj]: value G(c = 32co + (lane&31), j = 32jt + row(4q+jj) + 4(lane>>5))
static std::vector<float> pack_gamma(const float *g, int C, int ct) {
    std::vector<float> out((size_t)ct * ct * 4 * 256, 0.0f);
    size_t o = 0;
    for (int jt = 0; jt < ct; ++jt)
        for (int co = 0; co < ct; ++co)
            for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 4; ++jj, ++o) {
                        const int s = 4 * q + jj;
                        const int c = 32 * co + (lane & 31);
                        const int j = 32 * jt + (s & 3) + 8 * (s >> 2) + 4 * (lane >> 5);
                        if (c < C && j < C) out[o] = g[(size_t)c * C + j];
                    }
    return out;
}

// first analysis layer (cin <= 4): [tap][ct][lane][2]: W(cout = 32ct + (lane&31), ch = 2j + (lane>>5), tap)
static std::vector<float> pack_first(const float *w, int cin, int cout, int ks, int ct) {
    std::vector<float> out((size_t)ks * ks * ct * 128, 0.0f);
    size_t o = 0;
    for (int tap = 0; tap < ks * ks; ++tap)
        for (int t = 0; t < ct; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 2; ++j, ++o) {
                    const int co = 32 * t + (lane & 31), ch = 2 * j + (lane >> 5);
                    if (co < cout && ch < cin) out[o] = w[((size_t)co * cin + ch) * ks * ks + tap];
                }
    return out;
}

// last synthesis layer (cout <= 4): [nd][ndx][q][lane][s]:
//   A(row = lane&15 = 4c + 2py + px, cin = 16q + 4(lane>>4) + s) = W[cin][c][2d+py+P][2dx+px+P]
static std::vector<float> pack_last(const float *w, int cin, int cout, int ks) {
    const int P = ks / 2, dlo = -((P + 1) / 2), dhi = (ks - 1 - P) / 2, nb = dhi - dlo + 1;
    const int nq = (cin + 15) / 16;
    std::vector<float> out((size_t)nb * nb * nq * 256, 0.0f);
    size_t o = 0;
    for (int nd = 0; nd < nb; ++nd)
        for (int ndx = 0; ndx < nb; ++ndx)
            for (int q = 0; q < nq; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s2 = 0; s2 < 4; ++s2, ++o) {
                        const int row = lane & 15, c = row >> 2, py = (row >> 1) & 1, px = row & 1;
                        const int ci = 16 * q + 4 * (lane >> 4) + s2;
                        const int ky = 2 * (dlo + nd) + py + P, kx = 2 * (dlo + ndx) + px + P;
                        if (c < cout && ci < cin && ky >= 0 && ky < ks && kx >= 0 && kx < ks)
                            out[o] = w[(((size_t)ci * cout + c) * ks + ky) * ks + kx];
                    }
    return out;
}

// ---- f16x3 packing -------------------------------------------------------------------------------
static inline void split_half(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// every entry representable in the split format (finite, |v| <= 65504)?
static bool fits_f16(const float *v, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!(std::fabs(v[i]) <= 65504.0f)) return false;
    return true;
}

// weights -> [q][ky][kx][ct][hl][lane][8]: W(cout = 32ct + (lane&31), cin = 16q + 8(lane>>5) + j, ky, kx)
static std::vector<_Float16> pack_weights_f16(const float *w, bool transposed, int cin, int cout, int ks, int ct,
                                              bool flip = false) {
    const int nq = (cin + 15) / 16;
    std::vector<_Float16> out((size_t)nq * ks * ks * ct * 2 * 512, (_Float16)0.0f);
    for (int q = 0; q < nq; ++q)
        for (int ky = 0; ky < ks; ++ky)
            for (int kx = 0; kx < ks; ++kx)
                for (int t = 0; t < ct; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = 32 * t + (lane & 31);
                            const int ci = 16 * q + 8 * (lane >> 5) + j;
                            float v = 0.0f;
                            if (co < cout && ci < cin) {
                                const int sy = flip ? ks - 1 - ky : ky, sx = flip ? ks - 1 - kx : kx;
                                v = transposed ? w[(((size_t)ci * cout + co) * ks + sy) * ks + sx]
                                               : w[(((size_t)co * cin + ci) * ks + sy) * ks + sx];
                            }
                            _Float16 hi, lo;
                            split_half(v, hi, lo);
                            const size_t base = (((((size_t)q * ks + ky) * ks + kx) * ct + t) * 2) * 512;
                            out[base + (size_t)lane * 8 + j] = hi;
                            out[base + 512 + (size_t)lane * 8 + j] = lo;
                        }
    return out;
}

// gamma -> [jt][co][s][hl][lane][8]: G(c = 32co + (lane&31), j = 32jt + row(8s+e) + 4(lane>>5))
static std::vector<_Float16> pack_gamma_f16(const float *g, int C, int ct) {
    std::vector<_Float16> out((size_t)ct * ct * 2 * 2 * 512, (_Float16)0.0f);
    for (int jt = 0; jt < ct; ++jt)
        for (int co = 0; co < ct; ++co)
            for (int s2 = 0; s2 < 2; ++s2)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int r = 8 * s2 + e;
                        const int c = 32 * co + (lane & 31);
                        const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const float v = (c < C && j < C) ? g[(size_t)c * C + j] : 0.0f;
                        _Float16 hi, lo;
                        split_half(v, hi, lo);
                        const size_t base = ((((size_t)jt * ct + co) * 2 + s2) * 2) * 512;
                        out[base + (size_t)lane * 8 + e] = hi;
                        out[base + 512 + (size_t)lane * 8 + e] = lo;
                    }
    return out;
}

// first layer f16x3: [s][ct][hl][lane][8]: W(cout = 32ct + (lane&31), tap = 4s + 2(lane>>5) + (j>>2), ch = j&3)
static std::vector<_Float16> pack_first_f16(const float *w, int cin, int cout, int ks, int ct) {
    const int ns = (ks * ks + 3) / 4;
    std::vector<_Float16> out((size_t)ns * ct * 2 * 512, (_Float16)0.0f);
    for (int s2 = 0; s2 < ns; ++s2)
        for (int t = 0; t < ct; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = 32 * t + (lane & 31), tap = 4 * s2 + 2 * (lane >> 5) + (j >> 2), ch = j & 3;
                    float v = 0.0f;
                    if (co < cout && ch < cin && tap < ks * ks) v = w[((size_t)co * cin + ch) * ks * ks + tap];
                    _Float16 hi, lo;
                    split_half(v, hi, lo);
                    const size_t base = (((size_t)s2 * ct + t) * 2) * 512;
                    out[base + (size_t)lane * 8 + j] = hi;
                    out[base + 512 + (size_t)lane * 8 + j] = lo;
                }
    return out;
}

// last layer f16x3: [nd][ndx][q][hl][lane][8]: A(row = lane&15 = 4c + 2py + px, cin = 32q + 8(lane>>4) + j)
static std::vector<_Float16> pack_last_f16(const float *w, int cin, int cout, int ks) {
    const int P = ks / 2, dlo = -((P + 1) / 2), dhi = (ks - 1 - P) / 2, nb = dhi - dlo + 1;
    const int nq = (cin + 31) / 32;
    std::vector<_Float16> out((size_t)nb * nb * nq * 2 * 512, (_Float16)0.0f);
    for (int nd = 0; nd < nb; ++nd)
        for (int ndx = 0; ndx < nb; ++ndx)
            for (int q = 0; q < nq; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = lane & 15, c = row >> 2, py = (row >> 1) & 1, px = row & 1;
                        const int ci = 32 * q + 8 * (lane >> 4) + j;
                        const int ky = 2 * (dlo + nd) + py + P, kx = 2 * (dlo + ndx) + px + P;
                        float v = 0.0f;
                        if (c < cout && ci < cin && ky >= 0 && ky < ks && kx >= 0 && kx < ks)
                            v = w[(((size_t)ci * cout + c) * ks + ky) * ks + kx];
                        _Float16 hi, lo;
                        split_half(v, hi, lo);
                        const size_t base = ((((size_t)nd * nb + ndx) * nq + q) * 2) * 512;
                        out[base + (size_t)lane * 8 + j] = hi;
                        out[base + 512 + (size_t)lane * 8 + j] = lo;
                    }
    return out;
}

// last layer as a product map (cae_kernels_f16.hpp, pmap): [jt][s][hl][lane][8]:
//   A(row = lane&31 = 3 tap + c, k = 32jt + row(8s+e) + 4(lane>>5)) = W[cin = k][c][ky][kx], tap = 3 ky + kx  (k = 3)
static std::vector<_Float16> pack_pmap_f16(const float *w, int cin, int cout, int njt) {
    // njt = channel tiles of the PRODUCING layer's accumulators (round_ct(cin): 96 channels live in 4 tiles), zero padded
    std::vector<_Float16> out((size_t)njt * 2 * 2 * 512, (_Float16)0.0f);
    for (int jt = 0; jt < njt; ++jt)
        for (int s2 = 0; s2 < 2; ++s2)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int r = 8 * s2 + e;
                    // map row (lane & 31) = slot of the record as pmap_gather_kernel reads it (cae_kernels_f16.hpp):
                    // [taps 4, 5, 7, 8 | taps 3, 6, pad 2 | taps 1, 2, pad 2 | tap 0, pad 1] x 3 channels
                    static const int slot_tap[32] = {4, 4, 4, 5, 5, 5, 7, 7, 7, 8, 8, 8, 3, 3, 3, 6, 6, 6, -1, -1,
                                                     1, 1, 1, 2, 2, 2, -1, -1, 0, 0, 0, -1};
                    static const int slot_c[32] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 0,
                                                   0, 1, 2, 0, 1, 2, 0, 0, 0, 1, 2, 0};
                    const int row = lane & 31, tap = slot_tap[row], c = slot_c[row];
                    const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = 0.0f;
                    if (tap >= 0 && c < cout && j < cin) v = w[((size_t)j * cout + c) * 9 + tap];
                    _Float16 hi, lo;
                    split_half(v, hi, lo);
                    const size_t base = (((size_t)jt * 2 + s2) * 2) * 512;
                    out[base + (size_t)lane * 8 + e] = hi;
                    out[base + 512 + (size_t)lane * 8 + e] = lo;
                }
    return out;
}

static int upload_raw(const void *src, size_t bytes, void **dev) {
    if (*dev) {
        (void)hipFree(*dev);
        *dev = nullptr;
    }
    HIP_TRY(hipMalloc(dev, bytes));
    HIP_TRY(hipMemcpy(*dev, src, bytes, hipMemcpyHostToDevice));
    return CAE_OK;
}

static int upload(const std::vector<float> &v, float **dev) {
    if (*dev) {
        (void)hipFree(*dev);
        *dev = nullptr;
    }
    HIP_TRY(hipMalloc((void **)dev, v.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(*dev, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return CAE_OK;
}

int Model::ensure_ws(int which, size_t bytes) {
    if (ws_bytes[which] >= bytes) return CAE_OK;
    if (ws[which]) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(ws[which]);
        ws[which] = nullptr;
        ws_bytes[which] = 0;
    }
    bytes = (bytes + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    HIP_TRY(hipMalloc(&ws[which], bytes));
    ws_bytes[which] = bytes;
    return CAE_OK;
}

int Model::order_stream(void *stream) {
    if (last_stream_set && last_stream != stream) {
        if (!order_event) HIP_TRY(hipEventCreateWithFlags((hipEvent_t *)&order_event, hipEventDisableTiming));
        HIP_TRY(hipEventRecord((hipEvent_t)order_event, (hipStream_t)last_stream));
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)order_event, 0));
    }
    last_stream = stream;
    last_stream_set = true;
    return CAE_OK;
}

int Model::ensure_device() {
    if (!zero) {
        HIP_TRY(hipMalloc((void **)&zero, 1024));  // zero page: out-of-range halo source, null medians (192 floats)
        HIP_TRY(hipMemset(zero, 0, 1024));
    }
    if (medians_dirty && ent.channels > 0) {
        int rc = upload(ent.medians, &medians_dev);
        if (rc) return rc;
        medians_dirty = false;
    }
    if (density_dirty && !density.empty()) {
        int rc = upload(density, &density_dev);
        if (rc) return rc;
        density_dirty = false;
    }
    if (!flags) {
        HIP_TRY(hipHostMalloc((void **)&flags, kFlagSlots * sizeof(int), hipHostMallocMapped));
        memset(flags, 0, kFlagSlots * sizeof(int));
        HIP_TRY(hipHostGetDevicePointer((void **)&flags_dev, flags, 0));
    }
    return CAE_OK;
}

int *Model::next_flag(int64_t *ticket) {
    *ticket = ++flag_seq;
    const int slot = (int)(*ticket % kFlagSlots);
    *(volatile int *)(flags + slot) = 0;  // the word's previous user was kFlagSlots calls ago
    g_last_ticket = *ticket;
    return flags_dev + slot;
}

bool Model::f16_usable() const {
    if (precision != 1 || g_force_fp32) return false;
    for (auto *tr : {&enc, &dec})
        for (auto &l : *tr)
            if (l.set && l.f16_bad) return false;
    return true;
}

static void free_stages(Layer &l) {
    for (auto &sg : l.stages) {
        if (sg.wp16) (void)hipFree(sg.wp16);
        if (sg.gp16) (void)hipFree(sg.gp16);
        for (float *q : {sg.wp, sg.bias, sg.gp, sg.beta})
            if (q) (void)hipFree(q);
    }
    l.stages.clear();
}

Model::~Model() {
    if (order_event) (void)hipEventDestroy((hipEvent_t)order_event);
    for (auto *tr : {&enc, &dec})
        for (auto &l : *tr) {
            if (l.wp) (void)hipFree(l.wp);
            if (l.bias) (void)hipFree(l.bias);
            if (l.gp) (void)hipFree(l.gp);
            if (l.beta) (void)hipFree(l.beta);
            if (l.wp_edge) (void)hipFree(l.wp_edge);
            free_stages(l);
            if (l.color_wp) (void)hipFree(l.color_wp);
            if (l.color_bias) (void)hipFree(l.color_bias);
            if (l.color_wp16) (void)hipFree(l.color_wp16);
            if (l.wp16) (void)hipFree(l.wp16);
            if (l.gp16) (void)hipFree(l.gp16);
            if (l.wp_edge16) (void)hipFree(l.wp_edge16);
            if (l.wp_pmap16) (void)hipFree(l.wp_pmap16);
        }
    for (int i = 0; i < 4; ++i)
        if (ws[i]) (void)hipFree(ws[i]);
    for (int i = 0; i < 2; ++i)
        if (ws16[i]) (void)hipFree(ws16[i]);
    if (zero) (void)hipFree(zero);
    if (flags) (void)hipHostFree(flags);
    if (medians_dev) (void)hipFree(medians_dev);
    if (density_dev) (void)hipFree(density_dev);
    if (bits_ws) (void)hipFree(bits_ws);
}

// ---- kernel dispatch ---------------------------------------------------------------------------
struct ProfScope {
    Model *m;
    int track;
    hipStream_t st;
    std::vector<std::pair<void *, void *>> ev;
    hipEvent_t cur = nullptr;
    ProfScope(Model *m_, int track_, hipStream_t st_) : m(m_), track(track_), st(st_) {}
    void begin() {
        if (!m->profiling) return;
        hipEvent_t a;
        if (hipEventCreate(&a) != hipSuccess) return;
        (void)hipEventRecord(a, st);
        cur = a;
    }
    void end() {
        if (!m->profiling || !cur) return;
        hipEvent_t b;
        if (hipEventCreate(&b) != hipSuccess) return;
        (void)hipEventRecord(b, st);
        ev.emplace_back((void *)cur, (void *)b);
        cur = nullptr;
    }
    ~ProfScope() {
        if (m->profiling && !ev.empty()) m->prof[track].push_back(std::move(ev));
    }
};

static unsigned ew_grid(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)std::min<size_t>(std::max<size_t>(b, 1), 256 * 8 * 4);
}

}  // namespace cae

using namespace cae;

template <int R>
static void launch_likelihood(Model *m, const float *y, int n, int hw, float *yhat, float *lik, double *part,
                              hipStream_t st) {
    hipLaunchKernelGGL(likelihood_kernel<R>, dim3(m->c_bn, n), dim3(256), 0, st, y, m->medians_dev, m->density_dev,
                       m->density_per_channel, m->density_k, m->density_bound, m->likelihood_plain, m->c_bn, hw, yhat, lik, part);
}

// (re)builds stage `stage` of a unit: a stride-1 (transposed) convolution cin -> cin with its epilogue
static int set_stage(Model *m, int track, Layer &l, int stage, const float *w, const float *bias, const float *beta,
                     const float *gamma, int act, int add_residual, int post_act) {
    const int ctin = round_ct(l.cin);
    if (ctin < 0) return fail(CAE_ERR_UNSUPPORTED, "more than 192 channels not supported");
    if ((int)l.stages.size() <= stage) l.stages.resize(stage + 1);
    Layer::Stage &sg = l.stages[stage];
    // synthesis: ConvTranspose2d(stride 1, padding k//2) == zero-padded correlation with the flipped kernel
    const bool tr = track == CAE_SYNTHESIS;
    int rc = upload(pack_weights(w, tr, l.cin, l.cin, m->ks, ctin, tr), &sg.wp);
    if (rc) return rc;
    if (bias) {
        std::vector<float> b(ctin * 32, 0.0f);
        std::copy(bias, bias + l.cin, b.begin());
        if ((rc = upload(b, &sg.bias))) return rc;
    } else if (sg.bias) {
        (void)hipFree(sg.bias);
        sg.bias = nullptr;
    }
    sg.gdn = beta != nullptr;
    if (sg.gdn) {
        std::vector<float> b(ctin * 32, 1.0f);
        std::copy(beta, beta + l.cin, b.begin());
        if ((rc = upload(b, &sg.beta))) return rc;
        if ((rc = upload(pack_gamma(gamma, l.cin, ctin), &sg.gp))) return rc;
    }
    sg.act = act;
    sg.add_res = add_residual != 0;
    sg.post_act = post_act;
    if (m->precision == 1) {  // f16x3: the stage's weights as split halves (same packing as the strided layers)
        if (!fits_f16(w, (size_t)l.cin * l.cin * m->ks * m->ks)) l.f16_bad = true;
        auto w16 = pack_weights_f16(w, tr, l.cin, l.cin, m->ks, ctin, tr);
        if ((rc = upload_raw(w16.data(), w16.size() * sizeof(_Float16), &sg.wp16))) return rc;
        if (sg.gdn) {
            if (!fits_f16(gamma, (size_t)l.cin * l.cin)) l.f16_bad = true;
            auto g16 = pack_gamma_f16(gamma, l.cin, ctin);
            if ((rc = upload_raw(g16.data(), g16.size() * sizeof(_Float16), &sg.gp16))) return rc;
        }
    }
    if (track == CAE_ANALYSIS) {  // the fused first-layer kernels read the raw tile; a stage sits in between
        if (l.wp_edge) (void)hipFree(l.wp_edge);
        if (l.wp_edge16) (void)hipFree(l.wp_edge16);
        l.wp_edge = nullptr;
        l.wp_edge16 = nullptr;
    }
    return CAE_OK;
}

// Runs the stride-1 stages of a unit.  `cur`/`cur_idx`: the unit's input and the workspace slot it lives in; on
// return they describe the strided layer's input.  Slots 1..3 rotate so that the unit input survives until the
// residual sum has read it.
static int pick_slot(int a, int b) {
    for (int k = 1; k <= 3; ++k)
        if (k != a && k != b) return k;
    return 1;
}

// The split-f16 stride-1 kernel carries the (I)GDN / residual-sum epilogue up to 128 channels (registers); a wider unit
// takes its stages on the fp32 kernels, between two layout conversions.
static bool stages_need_fp32(const Layer &l) {
    if (round_ct(l.cin) <= 4) return false;
    for (const Layer::Stage &sg : l.stages)
        if (sg.gdn || sg.add_res || sg.post_act) return true;
    return false;
}

// f16: the split-f16 kernels (activation stages up to 192 channels; GDN / residual stages up to 128)
static int run_stages(Model *m, const Layer &l, bool synthesis, int n, int ch, int cw, const float *&cur, int &cur_idx,
                      int &cur_planes, hipStream_t st, bool f16 = false, int *flag = nullptr) {
    if (f16 && stages_need_fp32(l)) {
        // split rows -> fp32 C8 (the unit input is dead afterwards), the stages on the fp32 kernels, fp32 C8 -> split rows
        const int tmp = pick_slot(cur_idx, cur_idx);
        const size_t rows = (size_t)n * cur_planes * ch;
        if (synthesis)
            hipLaunchKernelGGL(c8s_to_c8_kernel<true>, dim3(ew_grid(rows * cw)), dim3(256), 0, st, (const char *)cur,
                               (float *)m->ws[tmp], rows, cw);
        else
            hipLaunchKernelGGL(c8s_to_c8_kernel<false>, dim3(ew_grid(rows * cw)), dim3(256), 0, st, (const char *)cur,
                               (float *)m->ws[tmp], rows, cw);
        HIP_TRY(hipGetLastError());
        cur = (const float *)m->ws[tmp];
        cur_idx = tmp;
        int rc = run_stages(m, l, synthesis, n, ch, cw, cur, cur_idx, cur_planes, st, false, nullptr);
        if (rc) return rc;
        const int back = pick_slot(cur_idx, cur_idx);
        const size_t orows = (size_t)n * cur_planes * ch;
        if (synthesis)
            hipLaunchKernelGGL(c8_to_c8s_kernel<true>, dim3(ew_grid(orows * cw)), dim3(256), 0, st, cur, (char *)m->ws[back],
                               orows, cw, flag);
        else
            hipLaunchKernelGGL(c8_to_c8s_kernel<false>, dim3(ew_grid(orows * cw)), dim3(256), 0, st, cur, (char *)m->ws[back],
                               orows, cw, flag);
        HIP_TRY(hipGetLastError());
        cur = (const float *)m->ws[back];
        cur_idx = back;
        return CAE_OK;
    }
    const float *unit_in = cur;
    const int unit_idx = cur_idx, unit_planes = cur_planes;
    for (const Layer::Stage &sg : l.stages) {
        LayerArgs b{};
        const int ctin = round_ct(l.cin);
        const int out_idx = pick_slot(unit_idx, cur_idx);
        b.in = cur;
        b.out = m->ws[out_idx];
        b.wp = sg.wp;
        b.bias = sg.bias;
        b.gp = sg.gp;
        b.beta = sg.beta;
        b.zero = m->zero;
        b.medians = m->zero;
        b.N = n;
        b.H = ch;
        b.W = cw;
        b.OH = ch;
        b.OW = cw;
        b.in_planes = cur_planes;
        b.cci = l.chunks;
        b.out_planes = ctin * 4;
        b.cout = l.cin;
        b.tiles_x = (cw + 15) / 16;
        b.tiles_y = (ch + 2 * CAE_CONV_NW - 1) / (2 * CAE_CONV_NW);
        b.outfmt = OUT_C8;
        b.act = sg.act;
        b.res = sg.add_res ? unit_in : nullptr;
        b.res_planes = unit_planes;
        b.post_act = sg.post_act;
        int rc;
        if (f16) {
            if (!sg.wp16 || (sg.gdn && !sg.gp16)) return fail(CAE_ERR_ARG, "stage uploaded before precision 1 was selected");
            b.wp = (const float *)sg.wp16;
            b.gp = (const float *)sg.gp16;
            b.cci = (l.cin + 15) / 16;
            b.tiles_y = (ch + 15) / 16;
            b.flag = flag;
            rc = launch_conv_s1_f16(m->ks, ctin, synthesis, sg.gdn, b, st);
        } else {
            rc = launch_conv_s1(m->ks, ctin, synthesis, sg.gdn, b, st);
        }
        if (rc) return rc;
        cur = (const float *)b.out;
        cur_idx = out_idx;
        cur_planes = ctin * 4;
    }
    return CAE_OK;
}

// LDS need of conv_s2_f16_kernel<KS,CT,GDN> (same formula as the kernel's constexprs): two stage buffers of
// weights + halo.  k=5 with 192 output channels needs 192 KiB: such a layer runs on the fp32 kernel instead.
// LDS need of deconv_last_f16_kernel (LastGeomF16: <3, 8, 3> and <5, 4, 2>): halo ring + all weights resident.  A last
// layer with more than 160 input channels does not fit and runs on the generic transposed-convolution kernel.
static bool last_f16_fits(int ks, int cin) {
    const int nq = (cin + 31) / 32;
    const int nb = ks == 3 ? 2 : 3, nw = ks == 3 ? 8 : 4, depth = ks == 3 ? 3 : 2;
    const int halo_instr = (8 * (nw + nb - 1) * (32 + nb - 1) + 63) / 64;
    const int stage = ((halo_instr + nw - 1) / nw) * nw * 1024;
    return depth * stage + nb * nb * nq * 2048 <= 160 * 1024;
}

static bool conv_f16_fits(int ks, int ct, bool gdn) {
    if (gdn && ct > 4) gdn = false;  // wider than 128 channels: convolution without the epilogue + gdn_f16_kernel
    const int halo_instr = (4 * 16 * (2 * 16 + ks - 2) + 63) / 64;
    const int stage = std::max(ks * ct * 2048 + halo_instr * 1024, gdn ? ct * 4096 : 0);
    return 2 * stage <= 160 * 1024;
}

extern "C" {

int cae_version(void) { return 1; }
const char *cae_last_error(void) { return g_last_error.c_str(); }
void cae_free(void *p) { free(p); }

int cae_model_create(int channels_org, int channels_net, int channels_bn, int compression_level, int kernel_size,
                     cae_model_t **out) {
    if (!out) return fail(CAE_ERR_ARG, "out is NULL");
    if (channels_org < 1 || channels_net < 1 || channels_bn < 1 || compression_level < 1)
        return fail(CAE_ERR_ARG, "bad model dimensions");
    if (kernel_size != 3 && kernel_size != 5)
        return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", kernel_size);
    // (the 192-channel limit of the conv kernels is enforced per layer in cae_model_set_layer;
    //  a handle that only carries entropy tables may have any number of channels)
    Model *m = new Model();
    m->c_org = channels_org;
    m->c_net = channels_net;
    m->c_bn = channels_bn;
    m->L = compression_level;
    m->ks = kernel_size;
    m->enc.resize(m->L);
    m->dec.resize(m->L);
    // no HIP call here: the host entropy coder of a handle works without a GPU; device
    // state is created on first device use (Model::ensure_device)
    *out = reinterpret_cast<cae_model_t *>(m);
    return CAE_OK;
}

void cae_model_destroy(cae_model_t *mm) {
    if (mm) {
        (void)hipDeviceSynchronize();
        delete reinterpret_cast<Model *>(mm);
    }
}

int cae_model_set_layer(cae_model_t *mm, int track, int index, int cin, int cout, const float *w, const float *bias,
                        const float *beta, const float *gamma) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !w) return fail(CAE_ERR_ARG, "NULL model or weight");
    if (track != CAE_ANALYSIS && track != CAE_SYNTHESIS) return fail(CAE_ERR_ARG, "bad track %d", track);
    if (index < 0 || index >= m->L) return fail(CAE_ERR_ARG, "layer index %d out of range", index);
    if ((beta == nullptr) != (gamma == nullptr)) return fail(CAE_ERR_ARG, "beta and gamma must come together");
    const int ct = round_ct(cout);
    if (ct < 0 || round_ct(cin) < 0) return fail(CAE_ERR_UNSUPPORTED, "more than 192 channels not supported");
    std::lock_guard<std::mutex> lk(m->mu);
    Layer &l = (track == CAE_ANALYSIS ? m->enc : m->dec)[index];
    l.cin = cin;
    l.cout = cout;
    l.ct = ct;
    l.chunks = (cin + 7) / 8;
    l.set = true;
    int rc = upload(pack_weights(w, track == CAE_SYNTHESIS, cin, cout, m->ks, ct), &l.wp);
    if (rc) return rc;
    if (bias) {
        std::vector<float> b(ct * 32, 0.0f);
        std::copy(bias, bias + cout, b.begin());
        if ((rc = upload(b, &l.bias))) return rc;
    } else if (l.bias) {
        (void)hipFree(l.bias);
        l.bias = nullptr;
    }
    if (l.wp_edge) {
        (void)hipFree(l.wp_edge);
        l.wp_edge = nullptr;
    }
    if (track == CAE_ANALYSIS && index == 0 && cin <= 4) {
        if ((rc = upload(pack_first(w, cin, cout, m->ks, ct), &l.wp_edge))) return rc;
    } else if (track == CAE_SYNTHESIS && index == m->L - 1 && cout <= 4 && beta == nullptr) {
        if ((rc = upload(pack_last(w, cin, cout, m->ks), &l.wp_edge))) return rc;
    }
    l.f16_bad = false;
    if (m->precision == 1) {
        // the split format holds |v| <= 65504: a model with larger (or non-finite) weights runs on the fp32 kernels
        const size_t nw = (size_t)cin * cout * m->ks * m->ks;
        l.f16_bad = !fits_f16(w, nw) || (gamma && !fits_f16(gamma, (size_t)cout * cout));
        if (l.wp_edge16) {
            (void)hipFree(l.wp_edge16);
            l.wp_edge16 = nullptr;
        }
        if (track == CAE_ANALYSIS && index == 0 && cin <= 4) {
            auto e16 = pack_first_f16(w, cin, cout, m->ks, ct);
            if ((rc = upload_raw(e16.data(), e16.size() * sizeof(_Float16), &l.wp_edge16))) return rc;
        } else if (track == CAE_SYNTHESIS && index == m->L - 1 && cout <= 4 && beta == nullptr) {
            auto e16 = pack_last_f16(w, cin, cout, m->ks);
            if ((rc = upload_raw(e16.data(), e16.size() * sizeof(_Float16), &l.wp_edge16))) return rc;
        }
        if (l.wp_pmap16) {
            (void)hipFree(l.wp_pmap16);
            l.wp_pmap16 = nullptr;
        }
        if (track == CAE_SYNTHESIS && index == m->L - 1 && m->ks == 3 && cout <= 3 && beta == nullptr) {
            auto pm16 = pack_pmap_f16(w, cin, cout, round_ct(cin));
            if ((rc = upload_raw(pm16.data(), pm16.size() * sizeof(_Float16), &l.wp_pmap16))) return rc;
        }
        auto w16 = pack_weights_f16(w, track == CAE_SYNTHESIS, cin, cout, m->ks, ct);
        if ((rc = upload_raw(w16.data(), w16.size() * sizeof(_Float16), &l.wp16))) return rc;
        if (gamma) {
            auto g16 = pack_gamma_f16(gamma, cout, ct);
            if ((rc = upload_raw(g16.data(), g16.size() * sizeof(_Float16), &l.gp16))) return rc;
        }
    }
    l.gdn = beta != nullptr;
    if (l.gdn) {
        std::vector<float> b(ct * 32, 1.0f);
        std::copy(beta, beta + cout, b.begin());
        if ((rc = upload(b, &l.beta))) return rc;
        if ((rc = upload(pack_gamma(gamma, cout, ct), &l.gp))) return rc;
    }
    return CAE_OK;
}

int cae_model_set_layer_act(cae_model_t *mm, int track, int index, int act, const float *pre_w, const float *pre_b) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m) return fail(CAE_ERR_ARG, "NULL model");
    if (track != CAE_ANALYSIS && track != CAE_SYNTHESIS) return fail(CAE_ERR_ARG, "bad track %d", track);
    if (index < 0 || index >= m->L) return fail(CAE_ERR_ARG, "layer index %d out of range", index);
    if (act < 0 || act > 2) return fail(CAE_ERR_ARG, "bad activation %d", act);
    std::lock_guard<std::mutex> lk(m->mu);
    Layer &l = (track == CAE_ANALYSIS ? m->enc : m->dec)[index];
    if (!l.set) return fail(CAE_ERR_ARG, "set the layer before its activation");
    if (l.gdn && (act != 0 || pre_w)) return fail(CAE_ERR_ARG, "a GDN unit has no other activation");
    if (pre_b && !pre_w) return fail(CAE_ERR_ARG, "pre-convolution bias without weight");
    l.act = act;
    free_stages(l);
    if (pre_w) {
        int rc = set_stage(m, track, l, 0, pre_w, pre_b, nullptr, nullptr, act, 0, 0);
        if (rc) return rc;
    }
    return CAE_OK;
}

int cae_model_set_layer_stage(cae_model_t *mm, int track, int index, int stage, const float *w, const float *bias,
                              const float *beta, const float *gamma, int act, int add_residual, int post_act) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !w) return fail(CAE_ERR_ARG, "NULL model or weight");
    if (track != CAE_ANALYSIS && track != CAE_SYNTHESIS) return fail(CAE_ERR_ARG, "bad track %d", track);
    if (index < 0 || index >= m->L) return fail(CAE_ERR_ARG, "layer index %d out of range", index);
    if (stage < 0 || stage > 1) return fail(CAE_ERR_ARG, "a unit has at most two stride-1 stages");
    if (act < 0 || act > 2 || post_act < 0 || post_act > 2) return fail(CAE_ERR_ARG, "bad activation");
    if ((beta == nullptr) != (gamma == nullptr)) return fail(CAE_ERR_ARG, "beta and gamma must come together");
    if (beta && act != 0) return fail(CAE_ERR_ARG, "a GDN stage has no other activation");
    std::lock_guard<std::mutex> lk(m->mu);
    Layer &l = (track == CAE_ANALYSIS ? m->enc : m->dec)[index];
    if (!l.set) return fail(CAE_ERR_ARG, "set the layer before its stages");
    if (stage > (int)l.stages.size()) return fail(CAE_ERR_ARG, "set stage 0 before stage 1");
    return set_stage(m, track, l, stage, w, bias, beta, gamma, act, add_residual, post_act);
}

int cae_model_set_color_layer(cae_model_t *mm, int index, int cin, int cout, const float *w, const float *bias) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !w) return fail(CAE_ERR_ARG, "NULL model or weight");
    if (index < 0 || index >= m->L - 1) return fail(CAE_ERR_ARG, "colour layer index %d out of range", index);
    if (cout < 1 || round_ct(cout) < 0 || round_ct(cin) < 0) return fail(CAE_ERR_UNSUPPORTED, "more than 192 channels not supported");
    std::lock_guard<std::mutex> lk(m->mu);
    Layer &l = m->dec[index];
    if (!l.set) return fail(CAE_ERR_ARG, "set the synthesis layer before its colour layer");
    if (cin != l.cout) return fail(CAE_ERR_ARG, "colour layer %d expects %d input channels, the level produces %d", index, cin, l.cout);
    const int ct = round_ct(cout);
    int rc = upload(pack_weights(w, false, cin, cout, m->ks, ct), &l.color_wp);
    if (rc) return rc;
    if (bias) {
        std::vector<float> b(ct * 32, 0.0f);
        std::copy(bias, bias + cout, b.begin());
        if ((rc = upload(b, &l.color_bias))) return rc;
    } else if (l.color_bias) {
        (void)hipFree(l.color_bias);
        l.color_bias = nullptr;
    }
    l.color_cout = cout;
    if (l.color_wp16) {
        (void)hipFree(l.color_wp16);
        l.color_wp16 = nullptr;
    }
    if (m->precision == 1 && ct == 1) {  // f16x3: colour layers to at most 32 channels (wider: the fp32 path)
        if (!fits_f16(w, (size_t)cin * cout * m->ks * m->ks)) l.f16_bad = true;
        auto w16 = pack_weights_f16(w, false, cin, cout, m->ks, ct);
        if ((rc = upload_raw(w16.data(), w16.size() * sizeof(_Float16), &l.color_wp16))) return rc;
    }
    return CAE_OK;
}

int cae_model_set_precision(cae_model_t *mm, int precision) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m) return fail(CAE_ERR_ARG, "NULL model");
    if (precision != 0 && precision != 1) return fail(CAE_ERR_ARG, "precision must be 0 (fp32) or 1 (f16x3)");
    std::lock_guard<std::mutex> lk(m->mu);
    m->precision = precision;
    return CAE_OK;
}

int64_t cae_last_range_ticket(void) { return g_last_ticket; }

void cae_thread_force_fp32(int on) { g_force_fp32 = on != 0; }

int cae_range_check(cae_model_t *mm, int64_t ticket, int *overflowed) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !overflowed) return fail(CAE_ERR_ARG, "NULL argument");
    *overflowed = 0;
    if (ticket == 0) return CAE_OK;  // an fp32 call: nothing to check
    std::lock_guard<std::mutex> lk(m->mu);
    if (ticket < 0 || ticket > m->flag_seq || !m->flags) return fail(CAE_ERR_ARG, "unknown range ticket");
    if (m->flag_seq - ticket >= Model::kFlagSlots)
        return fail(CAE_ERR_ARG, "range ticket too old (%d calls are tracked)", Model::kFlagSlots);
    *overflowed = *(volatile int *)(m->flags + ticket % Model::kFlagSlots) != 0;
    return CAE_OK;
}

int cae_model_effective_precision(cae_model_t *mm, int *precision) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !precision) return fail(CAE_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(m->mu);
    *precision = m->f16_usable() ? 1 : 0;
    return CAE_OK;
}

int cae_model_set_entropy(cae_model_t *mm, int channels, int cdf_stride, const int32_t *cdf, const int32_t *cdf_length,
                          const int32_t *offset, const float *medians) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !cdf || !cdf_length || !offset || !medians) return fail(CAE_ERR_ARG, "NULL argument");
    if (channels != m->c_bn) return fail(CAE_ERR_ARG, "entropy model has %d channels, model %d", channels, m->c_bn);
    for (int c = 0; c < channels; ++c) {
        if (cdf_length[c] < 2 || cdf_length[c] > cdf_stride)
            return fail(CAE_ERR_ARG, "cdf_length[%d]=%d out of range", c, cdf_length[c]);
    }
    std::lock_guard<std::mutex> lk(m->mu);
    m->ent.channels = channels;
    m->ent.stride = cdf_stride;
    m->ent.cdf.assign(cdf, cdf + (size_t)channels * cdf_stride);
    m->ent.len.assign(cdf_length, cdf_length + channels);
    m->ent.off.assign(offset, offset + channels);
    m->ent.medians.assign(medians, medians + channels);
    m->ent.build_tables();
    m->medians_dirty = true;
    return CAE_OK;
}

static int analysis_impl(cae_model_t *mm, const void *tiles, int fmt, int n, int h, int w, void *latents, bool symbols,
                         void *stream);

int cae_analysis(cae_model_t *mm, const void *tiles, int fmt, int n, int h, int w, float *latents, void *stream) {
    return analysis_impl(mm, tiles, fmt, n, h, w, latents, false, stream);
}

int cae_analysis_symbols(cae_model_t *mm, const void *tiles, int fmt, int n, int h, int w, int32_t *symbols,
                         void *stream) {
    return analysis_impl(mm, tiles, fmt, n, h, w, symbols, true, stream);
}

static int analysis_impl(cae_model_t *mm, const void *tiles, int fmt, int n, int h, int w, void *latents, bool symbols,
                         void *stream) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !tiles || !latents) return fail(CAE_ERR_ARG, "NULL argument");
    if (symbols && m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (n < 1 || h < 2 || w < 2) return fail(CAE_ERR_ARG, "bad tile batch %dx%dx%d", n, h, w);
    if (fmt != CAE_FMT_U8_HWC && fmt != CAE_FMT_F32_NCHW) return fail(CAE_ERR_ARG, "bad pixel format %d", fmt);
    for (auto &l : m->enc)
        if (!l.set) return fail(CAE_ERR_ARG, "analysis layer not set");
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(m->mu);
    int rc;
    if ((rc = m->ensure_device()) || (rc = m->order_stream(stream))) return rc;
    const bool f16 = m->f16_usable();
    const bool first_fused = f16 ? m->enc[0].wp_edge16 != nullptr : m->enc[0].wp_edge != nullptr;
    int64_t ticket = 0;
    g_last_ticket = 0;
    int *flag = f16 ? m->next_flag(&ticket) : m->flags_dev;  // (fp32 kernels never write it)

    // workspace: ws[0] = converted input, ws[1]/ws[2] ping-pong.  fp32 C8 and split C8S records are
    // both 32 B per (plane, pixel), so the same buffers serve either precision.
    const int p0 = f16 ? 2 * ((m->c_org + 15) / 16) : (m->c_org + 7) / 8;
    // f16x3: rows are padded to whole 32-pixel groups (C8S, cae_kernels_f16.hpp)
    auto row_bytes = [&](int cw) { return f16 ? c8s_row_bytes<false>(cw) : (size_t)cw * 32; };
    size_t in_bytes = first_fused ? 0 : (size_t)n * p0 * h * row_bytes(w);
    size_t maxact = 0;
    {
        int ch = h, cw = w;
        for (int i = 0; i < m->L; ++i) {
            if (!m->enc[i].stages.empty())  // stride-1 stages: same size, cin channels
                maxact = std::max(maxact, (size_t)n * round_ct(m->enc[i].cin) * 4 * ch * row_bytes(cw));
            ch = (ch + 1) / 2;
            cw = (cw + 1) / 2;
            if (i + 1 < m->L) maxact = std::max(maxact, (size_t)n * m->enc[i].ct * 4 * ch * row_bytes(cw));
        }
    }
    const bool u8_wide = f16 && !first_fused && fmt == CAE_FMT_U8_HWC;  // staged through fp32 C8 in ws[1]
    if (u8_wide) maxact = std::max(maxact, (size_t)n * p0 * h * w * 32);
    bool need_third_slot = false;  // two-stage residual units keep the unit input alive across both stages
    for (auto &l : m->enc)
        need_third_slot |= l.stages.size() > 1 || (f16 && (!conv_f16_fits(m->ks, l.ct, l.gdn) || stages_need_fp32(l)));
    if ((rc = m->ensure_ws(0, in_bytes))) return rc;
    if (maxact && ((rc = m->ensure_ws(1, maxact)) || (rc = m->ensure_ws(2, maxact)))) return rc;
    if (maxact && need_third_slot && (rc = m->ensure_ws(3, maxact))) return rc;

    ProfScope prof(m, CAE_ANALYSIS, st);
    prof.begin();
    if (!first_fused) {
        const size_t tot = (size_t)n * p0 * h * w;
        if (f16) {
            if (fmt == CAE_FMT_U8_HWC) {
                // rare (more than 4 input channels): uint8 -> fp32 C8 (exact /255) in ws[1] -> split rows in ws[0]
                hipLaunchKernelGGL(u8hwc_to_c8_kernel, dim3(ew_grid(tot)), dim3(256), 0, st, (const uint8_t *)tiles,
                                   (float *)m->ws[1], n, h, w, m->c_org, p0);
                hipLaunchKernelGGL(c8_to_c8s_kernel<false>, dim3(ew_grid(tot)), dim3(256), 0, st, (const float *)m->ws[1],
                                   (char *)m->ws[0], (size_t)n * p0 * h, w, flag);
            } else {
                hipLaunchKernelGGL(nchw_to_c8s_kernel<false>, dim3(ew_grid(tot)), dim3(256), 0, st,
                                   (const float *)tiles, (char *)m->ws[0], n, m->c_org, h, w, p0, flag);
            }
        } else if (fmt == CAE_FMT_U8_HWC) {
            hipLaunchKernelGGL(u8hwc_to_c8_kernel, dim3(ew_grid(tot)), dim3(256), 0, st, (const uint8_t *)tiles,
                               (float *)m->ws[0], n, h, w, m->c_org, p0);
        } else {
            hipLaunchKernelGGL(nchw_to_c8_kernel, dim3(ew_grid(tot)), dim3(256), 0, st, (const float *)tiles,
                               (float *)m->ws[0], n, m->c_org, h * w, p0);
        }
        HIP_TRY(hipGetLastError());
    }
    prof.end();

    const float *cur = (const float *)m->ws[0];
    int cur_planes = p0, ch = h, cw = w;
    int cur_idx = 0;  // workspace slot holding `cur` (0 = converted input; 1..3 rotate)
    for (int i = 0; i < m->L; ++i) {
        const Layer &l = m->enc[i];
        const bool last = i == m->L - 1;
        if (!l.stages.empty() && (rc = run_stages(m, l, false, n, ch, cw, cur, cur_idx, cur_planes, st, f16, flag))) return rc;
        LayerArgs a{};
        a.in = cur;
        const int out_idx = pick_slot(cur_idx, cur_idx);
        a.out = last ? latents : m->ws[out_idx];
        a.act = l.act;
        a.wp = l.wp;
        a.bias = l.bias;
        a.gp = l.gp;
        a.beta = l.beta;
        a.zero = m->zero;
        a.medians = m->zero;
        a.flag = flag;
        a.N = n;
        a.H = ch;
        a.W = cw;
        a.OH = (ch + 1) / 2;
        a.OW = (cw + 1) / 2;
        a.in_planes = cur_planes;
        a.cci = l.chunks;
        a.out_planes = l.ct * 4;
        a.cout = l.cout;
        a.tiles_x = (a.OW + 15) / 16;
        a.tiles_y = (a.OH + 2 * CAE_CONV_NW - 1) / (2 * CAE_CONV_NW);
        a.outfmt = last ? (symbols ? OUT_SYM : OUT_NCHW) : OUT_C8;
        if (last && symbols) a.medians = m->medians_dev;
        prof.begin();
        if (i == 0 && first_fused) {
            FirstArgs f{tiles, fmt == CAE_FMT_U8_HWC ? 1 : 0, l.cin};
            a.tiles_y = (a.OH + 7) / 8;
            if (f16) {
                a.wp = (const float *)l.wp_edge16;
                a.gp = (const float *)l.gp16;
                if ((rc = launch_first_f16(m->ks, l.ct, l.gdn, a, f, st))) return rc;
            } else {
                a.wp = l.wp_edge;
                if ((rc = launch_first(m->ks, l.ct, l.gdn, a, f, st))) return rc;
            }
        } else if (f16 && conv_f16_fits(m->ks, l.ct, l.gdn) && !(l.gdn && l.ct > 4 && last)) {
            a.wp = (const float *)l.wp16;
            a.gp = (const float *)l.gp16;
            a.cci = (l.cin + 15) / 16;
            a.tiles_y = (a.OH + 15) / 16;
            if ((rc = launch_conv_f16(m->ks, l.ct, l.gdn, a, st))) return rc;
        } else if (f16) {
            // this layer on the exact-fp32 kernel: split rows -> fp32 C8, convolution, (fp32 C8 -> split rows)
            const int tmp_in = pick_slot(cur_idx, out_idx);
            const size_t rows = (size_t)n * cur_planes * ch;
            hipLaunchKernelGGL(c8s_to_c8_kernel<false>, dim3(ew_grid(rows * cw)), dim3(256), 0, st, (const char *)cur,
                               (float *)m->ws[tmp_in], rows, cw);
            HIP_TRY(hipGetLastError());
            a.in = (const float *)m->ws[tmp_in];
            void *final_out = a.out;
            if (!last) a.out = m->ws[cur_idx == 0 ? pick_slot(tmp_in, out_idx) : cur_idx];  // the input slot is free now
            if ((rc = launch_conv(m->ks, l.ct, l.gdn, a, st))) return rc;
            if (!last) {
                const size_t orows = (size_t)n * l.ct * 4 * a.OH;
                hipLaunchKernelGGL(c8_to_c8s_kernel<false>, dim3(ew_grid(orows * a.OW)), dim3(256), 0, st, (const float *)a.out,
                                   (char *)final_out, orows, a.OW, flag);
                HIP_TRY(hipGetLastError());
                a.out = final_out;
            }
        } else {
            if ((rc = launch_conv(m->ks, l.ct, l.gdn, a, st))) return rc;
        }
        prof.end();
        cur = (const float *)a.out;
        cur_planes = l.ct * 4;
        ch = a.OH;
        cw = a.OW;
        cur_idx = out_idx;
    }
    return CAE_OK;
}

static int synthesis_impl(cae_model_t *mm, const float *latents, const int32_t *symbols, int n, int lh, int lw, void *out,
                          int fmt, float *const *bridges, float *const *colors, void *stream);

int cae_synthesis(cae_model_t *mm, const float *latents, int n, int lh, int lw, void *out, int fmt,
                  float *const *bridges, void *stream) {
    if (!latents) return fail(CAE_ERR_ARG, "NULL argument");
    return synthesis_impl(mm, latents, nullptr, n, lh, lw, out, fmt, bridges, nullptr, stream);
}

int cae_synthesis_multiscale(cae_model_t *mm, const float *latents, int n, int lh, int lw, void *out, int fmt,
                             float *const *bridges, float *const *colors, void *stream) {
    if (!latents) return fail(CAE_ERR_ARG, "NULL argument");
    return synthesis_impl(mm, latents, nullptr, n, lh, lw, out, fmt, bridges, colors, stream);
}

int cae_synthesis_symbols(cae_model_t *mm, const int32_t *symbols, int n, int lh, int lw, void *out, int fmt,
                          void *stream) {
    if (!symbols) return fail(CAE_ERR_ARG, "NULL argument");
    return synthesis_impl(mm, nullptr, symbols, n, lh, lw, out, fmt, nullptr, nullptr, stream);
}

static int synthesis_impl(cae_model_t *mm, const float *latents, const int32_t *symbols, int n, int lh, int lw, void *out,
                          int fmt, float *const *bridges, float *const *colors, void *stream) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !out) return fail(CAE_ERR_ARG, "NULL argument");
    if (symbols && m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (n < 1 || lh < 1 || lw < 1) return fail(CAE_ERR_ARG, "bad latent batch %dx%dx%d", n, lh, lw);
    if (fmt != CAE_FMT_U8_HWC && fmt != CAE_FMT_F32_NCHW) return fail(CAE_ERR_ARG, "bad pixel format %d", fmt);
    for (auto &l : m->dec)
        if (!l.set) return fail(CAE_ERR_ARG, "synthesis layer not set");
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(m->mu);
    int rc;
    if ((rc = m->ensure_device()) || (rc = m->order_stream(stream))) return rc;
    const bool f16 = m->f16_usable();
    int64_t ticket = 0;
    g_last_ticket = 0;
    int *flag = f16 ? m->next_flag(&ticket) : m->flags_dev;

    // planes of the converted latents: padded so whole MFMA k-steps can be read (zero channels)
    // (fp32: an EVEN number of 8-channel planes -- the last-layer kernel consumes 16-channel groups, and with one layer it
    //  reads these planes directly: 72 latent channels = 9 planes made it read a tenth one past the buffer)
    const int p0 = f16 ? 4 * ((m->c_bn + 31) / 32) : 2 * ((m->c_bn + 15) / 16);
    // f16x3: the synthesis track keeps its activations in C8SP rows (pitch = whole 64-pixel blocks)
    auto row_bytes = [&](int cw) { return f16 ? c8s_row_bytes<true>(cw) : (size_t)cw * 32; };
    size_t in_bytes = (size_t)n * p0 * lh * row_bytes(lw);
    size_t maxact = 0;
    {
        int ch = lh, cw = lw;
        for (int i = 0; i < m->L; ++i) {
            if (!m->dec[i].stages.empty())
                maxact = std::max(maxact, (size_t)n * round_ct(m->dec[i].cin) * 4 * ch * row_bytes(cw));
            ch *= 2;
            cw *= 2;
            if (i + 1 < m->L) maxact = std::max(maxact, (size_t)n * m->dec[i].ct * 4 * ch * row_bytes(cw));
        }
    }
    bool need_third_slot = false;
    for (auto &l : m->dec) need_third_slot |= l.stages.size() > 1 || (f16 && stages_need_fp32(l));
    if ((rc = m->ensure_ws(0, in_bytes))) return rc;
    if (maxact && ((rc = m->ensure_ws(1, maxact)) || (rc = m->ensure_ws(2, maxact)))) return rc;
    if (maxact && need_third_slot && (rc = m->ensure_ws(3, maxact))) return rc;

    ProfScope prof(m, CAE_SYNTHESIS, st);
    prof.begin();
    const size_t tot = (size_t)n * p0 * lh * lw;
    if (f16)
        hipLaunchKernelGGL(nchw_to_c8s_kernel<true>, dim3(ew_grid(tot)), dim3(256), 0, st, latents, (char *)m->ws[0], n,
                           m->c_bn, lh, lw, p0, flag, symbols, (const float *)m->medians_dev);
    else
        hipLaunchKernelGGL(nchw_to_c8_kernel, dim3(ew_grid(tot)), dim3(256), 0, st, latents, (float *)m->ws[0], n,
                           m->c_bn, lh * lw, p0, symbols, (const float *)m->medians_dev);
    HIP_TRY(hipGetLastError());
    prof.end();

    // Product-map form of the last two layers (cae_kernels_f16.hpp, pmap): layer L-2 stores the products of its output
    // with the last layer's weights, the last layer is a gather.  Needs the f16x3 transposed-convolution kernel for
    // layer L-2, k = 3, at most 3 image channels, and nobody asking for layer L-2's own output (bridges / colours).
    bool use_pmap = false;
    if (f16 && m->L >= 2 && m->ks == 3 && getenv("CAE_NO_PMAP") == nullptr) {
        const Layer &lp = m->dec[m->L - 2], &ll = m->dec[m->L - 1];
        // (conv_f16-style LDS budget: two 33-KiB stages + the transpose buffers fit for k = 3 and up to 128 channels)
        use_pmap = ll.wp_pmap16 && !ll.gdn && ll.stages.empty() && lp.stages.empty() && lp.ct <= 4 && lp.cout == ll.cin &&
                   !(bridges && bridges[m->L - 2]) && !(colors && colors[m->L - 2]);
    }
    if (use_pmap) maxact = std::max(maxact, (size_t)n * (lh << (m->L - 1)) * (lw << (m->L - 1)) * 128);
    if (use_pmap && ((rc = m->ensure_ws(1, maxact)) || (rc = m->ensure_ws(2, maxact)))) return rc;

    const float *cur = (const float *)m->ws[0];
    int cur_planes = p0, ch = lh, cw = lw;
    int cur_idx = 0;
    for (int i = 0; i < m->L; ++i) {
        const Layer &l = m->dec[i];
        const bool last = i == m->L - 1;
        if (use_pmap && last) {  // the gather half of the product map
            prof.begin();
            const int gtx = (cw + 15) / 16, gty = (ch + 15) / 16;
            hipLaunchKernelGGL(pmap_gather_kernel, dim3((unsigned)((size_t)n * gtx * gty)), dim3(256), 0, st, cur,
                               (const float *)l.bias, out, n, ch, cw, l.cout, fmt == CAE_FMT_U8_HWC ? OUT_U8HWC : OUT_NCHW, gtx,
                               gty);
            HIP_TRY(hipGetLastError());
            prof.end();
            break;
        }
        if (!l.stages.empty() && (rc = run_stages(m, l, true, n, ch, cw, cur, cur_idx, cur_planes, st, f16, flag))) return rc;
        LayerArgs a{};
        a.in = cur;
        const int out_idx = pick_slot(cur_idx, cur_idx);
        a.out = last ? out : m->ws[out_idx];
        a.act = l.act;
        a.wp = l.wp;
        a.bias = l.bias;
        a.gp = l.gp;
        a.beta = l.beta;
        a.zero = m->zero;
        a.medians = m->zero;
        a.flag = flag;
        a.N = n;
        a.H = ch;
        a.W = cw;
        a.OH = 2 * ch;
        a.OW = 2 * cw;
        a.in_planes = cur_planes;
        a.cci = l.chunks;
        a.out_planes = l.ct * 4;
        a.cout = l.cout;
        a.tiles_x = (cw + 31) / 32;
        a.tiles_y = (ch + CAE_DECONV_NW - 1) / CAE_DECONV_NW;
        a.outfmt = last ? (fmt == CAE_FMT_U8_HWC ? OUT_U8HWC : OUT_NCHW) : OUT_C8;
        if (use_pmap && i == m->L - 2) {
            a.outfmt = OUT_PMAP;
            a.pm = m->dec[m->L - 1].wp_pmap16;
        }
        prof.begin();
        if (f16) {
            a.gp = (const float *)l.gp16;
            if (last && l.wp_edge16 && last_f16_fits(m->ks, l.cin)) {
                a.wp = (const float *)l.wp_edge16;
                a.cci = (l.cin + 31) / 32;
                a.tiles_x = (cw + 31) / 32;
                a.tiles_y = (ch + 3) / 4;
                if ((rc = launch_last_f16(m->ks, a, st))) return rc;
            } else {
                if (l.act && l.ct > 4)  // (the 192-channel transposed-convolution kernel carries no activation: registers)
                    return fail(CAE_ERR_UNSUPPORTED, "f16x3: LeakyReLU / ReLU synthesis layers wider than 128 channels run "
                                                     "on the fp32 path: set precision 0");
                a.wp = (const float *)l.wp16;
                a.cci = (l.cin + 15) / 16;
                a.tiles_y = (ch + 7) / 8;
                if ((rc = launch_deconv_f16(m->ks, l.ct, l.gdn, a, st))) return rc;
            }
        } else if (last && l.wp_edge) {
            a.wp = l.wp_edge;
            a.cci = (l.cin + 15) / 16;
            a.tiles_x = (cw + 63) / 64;
            a.tiles_y = (ch + 3) / 4;
            if ((rc = launch_last(m->ks, a, st))) return rc;
        } else {
            if ((rc = launch_deconv(m->ks, l.ct, l.gdn, a, st))) return rc;
        }
        prof.end();
        if (!last && colors && colors[i]) {  // colour layer of this level (_autoencoders.py:417-436, :448-449)
            if (!l.color_wp) return fail(CAE_ERR_ARG, "colour layer %d not set", i);
            if (f16 && !l.color_wp16)
                return fail(CAE_ERR_UNSUPPORTED, "f16x3: colour layers to more than 32 channels run on the fp32 path: set precision 0");
            LayerArgs c{};
            c.in = (const float *)a.out;
            c.out = colors[i];
            c.wp = l.color_wp;
            c.bias = l.color_bias;
            c.zero = m->zero;
            c.medians = m->zero;
            c.N = n;
            c.H = a.OH;
            c.W = a.OW;
            c.OH = a.OH;
            c.OW = a.OW;
            c.in_planes = l.ct * 4;
            c.cci = (l.cout + 7) / 8;
            c.out_planes = round_ct(l.color_cout) * 4;
            c.cout = l.color_cout;
            c.tiles_x = (a.OW + 15) / 16;
            c.tiles_y = (a.OH + 2 * CAE_CONV_NW - 1) / (2 * CAE_CONV_NW);
            c.outfmt = OUT_NCHW;
            c.act = 0;
            if (f16) {  // split rows in (C8SP), reflect padding, NCHW fp32 out
                c.wp = (const float *)l.color_wp16;
                c.cci = (l.cout + 15) / 16;
                c.tiles_y = (a.OH + 15) / 16;
                c.flag = flag;
                if ((rc = launch_color_f16(m->ks, c, st))) return rc;
            } else if ((rc = launch_conv_s1(m->ks, round_ct(l.color_cout), false, false, c, st))) {
                return rc;
            }
        }
        if (!last && bridges && bridges[i]) {
            const size_t t2 = (size_t)n * l.cout * a.OH * a.OW;
            if (f16)
                hipLaunchKernelGGL(c8s_to_nchw_kernel<true>, dim3(ew_grid(t2)), dim3(256), 0, st, (const char *)a.out,
                                   bridges[i], n, l.cout, a.OH, a.OW, l.ct * 4);
            else
                hipLaunchKernelGGL(c8_to_nchw_kernel, dim3(ew_grid(t2)), dim3(256), 0, st, (const float *)a.out,
                                   bridges[i], n, l.cout, a.OH * a.OW, l.ct * 4);
            HIP_TRY(hipGetLastError());
        }
        cur = (const float *)a.out;
        cur_planes = l.ct * 4;
        ch = a.OH;
        cw = a.OW;
        cur_idx = out_idx;
    }
    return CAE_OK;
}

int cae_gdn_forward(cae_model_t *mm, int track, int index, const float *x, int n, int h, int w, float *y, void *stream) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !x || !y) return fail(CAE_ERR_ARG, "NULL argument");
    if (track != CAE_ANALYSIS && track != CAE_SYNTHESIS) return fail(CAE_ERR_ARG, "bad track %d", track);
    if (index < 0 || index >= m->L) return fail(CAE_ERR_ARG, "layer index %d out of range", index);
    if (n < 1 || h < 1 || w < 1) return fail(CAE_ERR_ARG, "bad tensor shape");
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(m->mu);
    const Layer &l = (track == CAE_ANALYSIS ? m->enc : m->dec)[index];
    if (!l.set || !l.gdn) return fail(CAE_ERR_ARG, "layer has no GDN");
    const int planes = l.ct * 4;
    int rc;
    if ((rc = m->ensure_device()) || (rc = m->order_stream(stream))) return rc;
    if ((rc = m->ensure_ws(0, (size_t)n * planes * h * w * 32))) return rc;
    const size_t tot = (size_t)n * planes * h * w;
    hipLaunchKernelGGL(nchw_to_c8_kernel, dim3(ew_grid(tot)), dim3(256), 0, st, x, (float *)m->ws[0], n, l.cout, h * w,
                       planes);
    HIP_TRY(hipGetLastError());
    LayerArgs a{};
    a.in = (const float *)m->ws[0];
    a.out = y;
    a.gp = l.gp;
    a.beta = l.beta;
    a.N = n;
    a.H = h;
    a.W = w;
    a.OH = h;
    a.OW = w;
    a.in_planes = planes;
    a.out_planes = planes;
    a.cout = l.cout;
    a.outfmt = OUT_NCHW;
    a.zero = m->zero;
    a.medians = m->zero;
    return launch_gdn(l.ct, track == CAE_SYNTHESIS, a, st);
}

int cae_tile_sse(const uint8_t *a, const uint8_t *b, int n, size_t elems, double *sse, void *stream) {
    if (!a || !b || !sse) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || elems < 1) return fail(CAE_ERR_ARG, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    // the float64 output doubles as the exact integer accumulator (same 8-byte cells)
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(sse);
    HIP_TRY(hipMemsetAsync(acc, 0, (size_t)n * 8, st));
    const unsigned bx = (unsigned)std::min<size_t>(std::max<size_t>(elems / 16 / 256, 1), 64);  // (byte loop: same grid)
    hipLaunchKernelGGL(tile_sse_kernel, dim3(bx, n), dim3(256), 0, st, a, b, elems, acc);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(u64_to_f64_kernel, dim3((n + 255) / 256), dim3(256), 0, st, acc, sse, n);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_tile_ssim(const uint8_t *a, const uint8_t *b, int n, int h, int w, int c, double *ssim, double *workspace,
                  size_t workspace_elems, void *stream) {
    if (!a || !b || !ssim || !workspace) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || n > 65535 || c < 1) return fail(CAE_ERR_ARG, "bad shape");
    if (h < 7 || w < 7) return fail(CAE_ERR_ARG, "tiles must be at least 7x7 (the SSIM window)");
    const int oh = h - 6, ow = w - 6;
    const int bxr = (ow + 31) / 32, bpt = bxr * ((oh + 31) / 32);
    if (workspace_elems < (size_t)n * bpt)
        return fail(CAE_ERR_ARG, "workspace too small: %zu doubles needed", (size_t)n * bpt);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tile_ssim_kernel, dim3(bpt, n), dim3(256), 0, st, a, b, h, w, c, bxr, bpt, workspace);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ssim_reduce_kernel, dim3(n), dim3(256), 0, st, workspace, bpt, (double)oh * ow * c, ssim);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_tile_delta_e(const uint8_t *a, const uint8_t *b, int n, size_t pixels, double *delta, double *workspace,
                     size_t workspace_elems, void *stream) {
    if (!a || !b || !delta || !workspace) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || n > 65535 || pixels < 1) return fail(CAE_ERR_ARG, "bad shape");
    const int bpt = (int)std::min<size_t>((pixels + 255) / 256, 128);
    if (workspace_elems < (size_t)n * bpt)
        return fail(CAE_ERR_ARG, "workspace too small: %zu doubles needed", (size_t)n * bpt);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tile_delta_e_kernel, dim3(bpt, n), dim3(256), 0, st, a, b, pixels, bpt, workspace);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ssim_reduce_kernel, dim3(n), dim3(256), 0, st, workspace, bpt, (double)pixels, delta);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_u8hwc_to_planes(const uint8_t *tiles, int n, int h, int w, int c, float *planes, void *stream) {
    if (!tiles || !planes) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1 || c < 1) return fail(CAE_ERR_ARG, "bad shape");
    hipLaunchKernelGGL(u8hwc_to_planes_kernel, dim3(ew_grid((size_t)n * c * h * w)), dim3(256), 0, (hipStream_t)stream,
                       tiles, planes, n, h, w, c);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_avgpool2(const float *in, int planes, int h, int w, float *out, void *stream) {
    if (!in || !out) return fail(CAE_ERR_ARG, "NULL argument");
    if (planes < 1 || h < 1 || w < 1) return fail(CAE_ERR_ARG, "bad shape");
    const int oh = (h + 2 * (h & 1) - 2) / 2 + 1, ow = (w + 2 * (w & 1) - 2) / 2 + 1;
    hipLaunchKernelGGL(avgpool2_kernel, dim3(ew_grid((size_t)planes * oh * ow)), dim3(256), 0, (hipStream_t)stream, in,
                       out, planes, h, w, oh, ow);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_msssim_level(const float *x, const float *y, int planes, int h, int w, const float *window11, double *ssim_cs,
                     double *workspace, size_t workspace_elems, void *stream) {
    if (!x || !y || !window11 || !ssim_cs || !workspace) return fail(CAE_ERR_ARG, "NULL argument");
    if (planes < 1 || planes > 65535) return fail(CAE_ERR_ARG, "bad plane count");
    if (h < 11 || w < 11) return fail(CAE_ERR_ARG, "image smaller than the 11-tap window");
    const int oh = h - 10, ow = w - 10;
    const int bxr = (ow + 31) / 32, bpp = bxr * ((oh + 31) / 32);
    if (workspace_elems < (size_t)planes * bpp * 2)
        return fail(CAE_ERR_ARG, "workspace too small: %zu doubles needed", (size_t)planes * bpp * 2);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(msssim_level_kernel, dim3(bpp, planes), dim3(256), 0, st, x, y, h, w, bxr, bpp, window11, workspace);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(msssim_reduce_kernel, dim3(planes), dim3(256), 0, st, workspace, bpp, (double)oh * ow, ssim_cs);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_model_set_profiling(cae_model_t *mm, int enable) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m) return fail(CAE_ERR_ARG, "NULL model");
    std::lock_guard<std::mutex> lk(m->mu);
    m->profiling = enable != 0;
    return CAE_OK;
}

int cae_model_get_profile(cae_model_t *mm, int track, double *ms, int n_slots, int *calls, int reset) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !ms || n_slots < 1) return fail(CAE_ERR_ARG, "NULL argument");
    if (track != CAE_ANALYSIS && track != CAE_SYNTHESIS) return fail(CAE_ERR_ARG, "bad track %d", track);
    std::lock_guard<std::mutex> lk(m->mu);
    for (int i = 0; i < n_slots; ++i) ms[i] = 0.0;
    auto &calls_v = m->prof[track];
    for (auto &call : calls_v)
        for (size_t i = 0; i < call.size(); ++i) {
            HIP_TRY(hipEventSynchronize((hipEvent_t)call[i].second));
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, (hipEvent_t)call[i].first, (hipEvent_t)call[i].second));
            if ((int)i < n_slots) ms[i] += t;
        }
    if (calls) *calls = (int)calls_v.size();
    if (reset) {
        for (auto &call : calls_v)
            for (auto &e : call) {
                (void)hipEventDestroy((hipEvent_t)e.first);
                (void)hipEventDestroy((hipEvent_t)e.second);
            }
        calls_v.clear();
    }
    return CAE_OK;
}

int cae_quantize(cae_model_t *mm, const float *latents, int n, int hw, int32_t *symbols, void *stream) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !latents || !symbols) return fail(CAE_ERR_ARG, "NULL argument");
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (n < 1 || hw < 1) return fail(CAE_ERR_ARG, "bad shape");
    {
        std::lock_guard<std::mutex> lk(m->mu);
        int rc = m->ensure_device();
        if (rc) return rc;
    }
    const size_t total = (size_t)n * m->c_bn * hw;
    hipLaunchKernelGGL(quantize_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, latents,
                       m->medians_dev, symbols, m->c_bn, hw, total);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_model_set_density(cae_model_t *mm, int channels, int n_filters, const int *filters, const float *const *matrices,
                          const float *const *biases, const float *const *factors, float likelihood_bound) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !filters || !matrices || !biases || !factors) return fail(CAE_ERR_ARG, "NULL argument");
    if (channels != m->c_bn) return fail(CAE_ERR_ARG, "density model has %d channels, model %d", channels, m->c_bn);
    if (n_filters < 1) return fail(CAE_ERR_UNSUPPORTED, "density network needs at least one hidden layer");
    int R = 0;
    for (int i = 0; i < n_filters; ++i) {
        if (filters[i] < 1) return fail(CAE_ERR_ARG, "filters[%d]=%d", i, filters[i]);
        R = std::max(R, filters[i]);
    }
    if (R > 8) return fail(CAE_ERR_UNSUPPORTED, "density filters wider than 8 are not supported (got %d)", R);
    for (int i = 0; i <= n_filters; ++i)
        if (!matrices[i] || !biases[i] || (i < n_filters && !factors[i])) return fail(CAE_ERR_ARG, "NULL parameter %d", i);
    const int K = n_filters;
    const int per = 3 * R + (K - 1) * (R * R + 2 * R) + R + 1;
    std::vector<float> packed((size_t)channels * per, 0.f);
    // narrower layers are embedded in width R with zero weights: the extra units stay exactly 0
    auto width = [&](int i) { return i == 0 ? 1 : (i == K + 1 ? 1 : filters[i - 1]); };  // F[i]
    for (int c = 0; c < channels; ++c) {
        float *o = packed.data() + (size_t)c * per;
        for (int i = 0; i <= K; ++i) {
            const int fin = width(i), fout = width(i + 1);
            const float *M = matrices[i] + (size_t)c * fout * fin;
            const float *b = biases[i] + (size_t)c * fout;
            const float *t = i < K ? factors[i] + (size_t)c * fout : nullptr;
            if (i == 0) {
                for (int j = 0; j < fout; ++j) { o[j] = M[j]; o[R + j] = b[j]; o[2 * R + j] = t[j]; }
                o += 3 * R;
            } else if (i < K) {
                for (int j = 0; j < fout; ++j) {
                    for (int k = 0; k < fin; ++k) o[j * R + k] = M[j * fin + k];
                    o[R * R + j] = b[j];
                    o[R * R + R + j] = t[j];
                }
                o += R * R + 2 * R;
            } else {
                for (int k = 0; k < fin; ++k) o[k] = M[k];
                o[R] = b[0];
            }
        }
    }
    std::lock_guard<std::mutex> lk(m->mu);
    m->density.swap(packed);
    m->density_r = R;
    m->density_k = K;
    m->density_per_channel = per;
    m->density_bound = likelihood_bound > 0.f ? likelihood_bound : 0.f;
    m->density_dirty = true;
    return CAE_OK;
}

int cae_model_set_likelihood_form(cae_model_t *mm, int form) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m) return fail(CAE_ERR_ARG, "NULL model");
    if (form != 0 && form != 1) return fail(CAE_ERR_ARG, "likelihood form must be 0 (plain) or 1 (sign trick)");
    std::lock_guard<std::mutex> lk(m->mu);
    m->likelihood_plain = form == 0;
    return CAE_OK;
}

int cae_likelihood(cae_model_t *mm, const float *latents, int n, int hw, float *y_hat, float *likelihood, double *bits,
                   void *stream) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !latents) return fail(CAE_ERR_ARG, "NULL argument");
    if (!y_hat && !likelihood && !bits) return fail(CAE_ERR_ARG, "no output requested");
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (m->density.empty()) return fail(CAE_ERR_ARG, "density model not set");
    if (n < 1 || hw < 1 || n > 65535) return fail(CAE_ERR_ARG, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(m->mu);
    int rc = m->ensure_device();
    if (rc) return rc;
    double *part = nullptr;
    if (bits) {
        const size_t need = (size_t)n * m->c_bn;
        if (m->bits_ws_elems < need) {
            if (m->bits_ws) {
                HIP_TRY(hipDeviceSynchronize());
                (void)hipFree(m->bits_ws);
                m->bits_ws = nullptr;
                m->bits_ws_elems = 0;
            }
            HIP_TRY(hipMalloc((void **)&m->bits_ws, need * sizeof(double)));
            m->bits_ws_elems = need;
        }
        part = m->bits_ws;
    }
    switch (m->density_r) {
        case 1: launch_likelihood<1>(m, latents, n, hw, y_hat, likelihood, part, st); break;
        case 2: launch_likelihood<2>(m, latents, n, hw, y_hat, likelihood, part, st); break;
        case 3: launch_likelihood<3>(m, latents, n, hw, y_hat, likelihood, part, st); break;
        case 4: launch_likelihood<4>(m, latents, n, hw, y_hat, likelihood, part, st); break;
        case 5: launch_likelihood<5>(m, latents, n, hw, y_hat, likelihood, part, st); break;
        case 6: launch_likelihood<6>(m, latents, n, hw, y_hat, likelihood, part, st); break;
        case 7: launch_likelihood<7>(m, latents, n, hw, y_hat, likelihood, part, st); break;
        default: launch_likelihood<8>(m, latents, n, hw, y_hat, likelihood, part, st); break;
    }
    HIP_TRY(hipGetLastError());
    if (bits) {
        hipLaunchKernelGGL(bits_reduce_kernel, dim3(n), dim3(256), 0, st, part, m->c_bn, bits);
        HIP_TRY(hipGetLastError());
    }
    return CAE_OK;
}

int cae_quantize_export(cae_model_t *mm, const float *latents, int n, int hw, int32_t *symbols_host, int max_blocks,
                        void *stream) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !latents || !symbols_host) return fail(CAE_ERR_ARG, "NULL argument");
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (n < 1 || hw < 1) return fail(CAE_ERR_ARG, "bad shape");
    {
        std::lock_guard<std::mutex> lk(m->mu);
        int rc = m->ensure_device();
        if (rc) return rc;
    }
    const size_t total = (size_t)n * m->c_bn * hw;
    // few, fat workgroups: a CU that hosts an export workgroup cannot host a workgroup of the register-file-filling
    // conv / deconv kernels, so the link is kept busy from as few CUs as possible
    const unsigned cap = (unsigned)std::max(max_blocks, 1);
    if (hw % 4 == 0 && (((uintptr_t)latents | (uintptr_t)symbols_host) & 15) == 0) {
        const size_t total4 = total / 4;
        const unsigned blocks = (unsigned)std::min<size_t>((total4 + 1023) / 1024, (size_t)cap);
        hipLaunchKernelGGL(quantize_export4_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, latents,
                           m->medians_dev, symbols_host, m->c_bn, hw / 4, total4);
    } else {
        const unsigned blocks = (unsigned)std::min<size_t>((total + 1023) / 1024, (size_t)cap);
        hipLaunchKernelGGL(quantize_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, latents, m->medians_dev,
                           symbols_host, m->c_bn, hw, total);
    }
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_dequantize(cae_model_t *mm, const int32_t *symbols, int n, int hw, float *latents, void *stream) {
    Model *m = reinterpret_cast<Model *>(mm);
    if (!m || !latents || !symbols) return fail(CAE_ERR_ARG, "NULL argument");
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (n < 1 || hw < 1) return fail(CAE_ERR_ARG, "bad shape");
    {
        std::lock_guard<std::mutex> lk(m->mu);
        int rc = m->ensure_device();
        if (rc) return rc;
    }
    const size_t total = (size_t)n * m->c_bn * hw;
    hipLaunchKernelGGL(dequantize_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, symbols,
                       m->medians_dev, latents, m->c_bn, hw, total);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

}  // extern "C"
