// Internal types of libcae_hip.so (not part of the ABI).
#pragma once
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <utility>
#include <vector>

namespace cae {

int fail(int code, const char *fmt, ...);

struct Layer {
    bool set = false;
    int cin = 0, cout = 0, ct = 0, chunks = 0;
    bool gdn = false;
    float *wp = nullptr;    // packed weights (device)
    float *bias = nullptr;  // [ct*32] (device) or null
    float *gp = nullptr;    // packed gamma (device)
    float *beta = nullptr;  // [ct*32] (device)
    float *wp_edge = nullptr;  // packed weights of the specialised first-conv / last-deconv kernel, or null
    // LeakyReLU / ReLU units: stride-1 convolution (cin -> cin) + activation in front of the strided one
    // stride-1 (transposed) convolutions cin -> cin in front of the strided layer: the pre-convolution of the
    // LeakyReLU / ReLU units and the res_model of the residual units (_autoencoders.py:62-76, :104-174, :230-304)
    struct Stage {
        float *wp = nullptr, *bias = nullptr, *gp = nullptr, *beta = nullptr;
        void *wp16 = nullptr;  // f16x3 path: packed hi/lo weights
        void *gp16 = nullptr;  // f16x3 path: packed hi/lo gamma (GDN / IGDN stages of residual units)
        bool gdn = false;      // GDN (analysis) / IGDN (synthesis) after the convolution, else `act`
        int act = 0;
        bool add_res = false;  // + the unit's input after the activation (residual units)
        int post_act = 0;      // activation after the residual sum (the strided layer's pre-activation)
    };
    std::vector<Stage> stages;
    // multiscale colour layer on this synthesis level's output (stride-1 reflect conv to the image channels), or null
    float *color_wp = nullptr;
    float *color_bias = nullptr;
    void *color_wp16 = nullptr;  // f16x3 path: packed hi/lo weights of the colour layer
    int color_cout = 0;
    int act = 0;               // activation after the pre-convolution and after this layer (0 none, 1 LeakyReLU, 2 ReLU)
    void *wp16 = nullptr;      // f16x3 path: packed hi/lo weights
    void *gp16 = nullptr;      // f16x3 path: packed hi/lo gamma
    void *wp_edge16 = nullptr; // f16x3 path: packed weights of the first-conv / last-deconv kernel
    void *wp_pmap16 = nullptr; // f16x3 path, last synthesis layer (k = 3, cout <= 3): its weights as the product map
    bool f16_bad = false;      // a weight / gamma entry is not finite in f16: the model runs on the fp32 kernels
};

// Integer tables of the factorized entropy model + per-row encoder constants.
struct EntropyTables {
    int channels = 0, stride = 0;
    std::vector<int32_t> cdf, len, off;
    std::vector<float> medians;
    // encoder: per (row, value) exact-division constants (ryg_rans "Rans64EncSymbol" form)
    struct EncSym {
        uint64_t rcp_freq;   // fixed-point reciprocal
        uint32_t freq;
        uint32_t bias;
        uint32_t cmpl_freq;  // (1 << 16) - freq
        uint32_t rcp_shift;
    };
    std::vector<EncSym> enc;  // [channels * stride]
    static constexpr int kLutBits = 10;
    std::vector<uint16_t> lut;  // [channels << kLutBits]: first symbol of each cumulative-frequency bucket
    void build_tables();
};

// single chunks coded by the calling thread (cae_entropy.cpp); `headroom` bytes stay free in front of the stream
int rans_encode_chunk(const EntropyTables &T, const int32_t *symbols, int hw, size_t headroom, uint8_t **out,
                      size_t *out_len);
int rans_encode_chunk_pair(const EntropyTables &T, const int32_t *const *symbols, int hw, size_t headroom, uint8_t **out,
                           size_t *out_len);
int rans_decode_chunk(const EntropyTables &T, const uint8_t *buf, size_t len, int hw, int32_t *symbols);
int rans_decode_chunk_pair(const EntropyTables &T, const uint8_t *const *bufs, const size_t *lens, int hw,
                           int32_t *const *symbols);

struct Model {
    int c_org = 0, c_net = 0, c_bn = 0, L = 0, ks = 3;
    std::vector<Layer> enc, dec;
    EntropyTables ent;
    float *medians_dev = nullptr;
    bool medians_dirty = false;
    // factorized density network (cae_model_set_density): effective parameters, uniform width
    std::vector<float> density;
    int density_r = 0, density_k = 0, density_per_channel = 0;
    float density_bound = 0.f;
    int likelihood_plain = 1;  // cae_model_set_likelihood_form
    float *density_dev = nullptr;
    bool density_dirty = false;
    double *bits_ws = nullptr;
    size_t bits_ws_elems = 0;
    float *zero = nullptr;
    void *ws[4] = {nullptr, nullptr, nullptr, nullptr};  // converted input + three activation buffers
    size_t ws_bytes[4] = {0, 0, 0, 0};
    std::mutex mu;
    // profiling: per track, per profiled call, the event pairs of every launched kernel
    int precision = 0;  // 0 = fp32 MFMA, 1 = f16x3 split MFMA
    void *ws16[2] = {nullptr, nullptr};
    size_t ws16_bytes[2] = {0, 0};
    bool profiling = false;
    std::vector<std::vector<std::pair<void *, void *>>> prof[2];
    // f16x3 range guard: ring of overflow words in pinned host memory (device-visible), one per call (ticket % kFlagSlots)
    static constexpr int kFlagSlots = 1024;
    int *flags = nullptr;      // host view
    int *flags_dev = nullptr;  // device view of the same words
    int64_t flag_seq = 0;
    int *next_flag(int64_t *ticket);  // takes a ticket, clears its word, remembers it as this thread's last ticket
    bool f16_usable() const;
    int ensure_ws(int which, size_t bytes);
    int ensure_device();
    // The workspaces are used in CALL order whatever stream a call names: a call on another stream than the one before
    // waits (on the device) for that one's work (order_stream, under `mu`).
    void *last_stream = nullptr;
    bool last_stream_set = false;
    void *order_event = nullptr;
    int order_stream(void *stream);
    ~Model();
};

}  // namespace cae
