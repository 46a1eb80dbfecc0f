// Codec front door: the reference's numcodecs contract as thread-safe C entry points (include/cae_hip.h, "codec front
// door").  dask's threaded scheduler calls Codec.encode / Codec.decode one zarr chunk at a time from a thread pool on
// ONE shared codec (src/compress.py:121-128, src/decompress.py:51-58; bodies _autoencoders.py:539-555, :557-584, batch
// of one at :544).  Here every such call
//   * stages its chunk in pinned memory and range-codes in the CALLING thread (no lock, no GIL: the Python binding is
//     one ctypes call), and
//   * has its GPU part coalesced with the calls that arrived meanwhile: a dispatcher thread launches the analysis /
//     synthesis for whatever is queued (chunks of one shape, at most max_batch) as soon as one of `inflight` flights is
//     free -- requests pile up exactly while the device is busy, nothing waits on a timer, a lone caller is served at
//     once; a completer thread waits for the flight's event, pulls the batch over the DMA engines into one pinned
//     buffer and wakes the callers.
// The kernels work on every tile by itself: results do not depend on the grouping (tests/test_frontdoor.py).
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace cae;

namespace {

using Clock = std::chrono::steady_clock;
inline double seconds(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

enum { kEncode = 0, kDecode = 1 };

struct Pinned {  // one hipHostMalloc'd block
    void *ptr = nullptr;
    size_t bytes = 0;
};

// free list of pinned blocks (hipHostMalloc costs milliseconds); blocks are handed out best-fit within 4x
struct PinnedPool {
    std::mutex mu;
    std::vector<Pinned> free_;
    size_t keep;
    explicit PinnedPool(size_t keep_) : keep(keep_) {}
    int get(size_t bytes, Pinned *out) {
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (int i = 0; i < (int)free_.size(); ++i)
                if (free_[i].bytes >= bytes && free_[i].bytes <= 4 * bytes && (best < 0 || free_[i].bytes < free_[best].bytes))
                    best = i;
            if (best >= 0) {
                *out = free_[best];
                free_.erase(free_.begin() + best);
                return CAE_OK;
            }
        }
        size_t cap = 1 << 16;
        while (cap < bytes) cap <<= 1;  // power-of-two classes: the batch size varies from flight to flight
        if (cap > bytes + (bytes >> 1)) cap = (bytes + (1u << 16)) & ~(size_t)((1u << 16) - 1);
        HIP_TRY(hipHostMalloc(&out->ptr, cap, hipHostMallocDefault));
        out->bytes = cap;
        return CAE_OK;
    }
    void put(Pinned p) {
        if (!p.ptr) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (free_.size() < keep) {
                free_.push_back(p);
                return;
            }
        }
        (void)hipHostFree(p.ptr);
    }
    ~PinnedPool() {
        for (auto &p : free_) (void)hipHostFree(p.ptr);
    }
};

struct OutBuf {  // pinned result of one flight, shared by its callers
    Pinned mem;
    std::atomic<int> users{0};
};

struct Request {
    int kind = kEncode;
    int h = 0, w = 0;        // encode: tile size; decode: latent size
    size_t in_bytes = 0, out_bytes = 0;
    Pinned staged;           // this call's input in pinned memory
    // answer
    const uint8_t *result = nullptr;
    OutBuf *out = nullptr;
    int rc = CAE_OK;
    std::string err;
    bool done = false;
    std::mutex mu;
    std::condition_variable cv;
    Clock::time_point t_submit, t_picked, t_launched, t_ready, t_pulled;
    bool same_key(const Request &o) const { return kind == o.kind && h == o.h && w == o.w; }
};

struct Flight {  // device side of one batch in flight
    void *d_in = nullptr, *d_out = nullptr;
    size_t d_in_bytes = 0, d_out_bytes = 0;
    hipEvent_t landed = nullptr, ready = nullptr;
    bool busy = false;
    std::vector<Request *> batch;
    int64_t ticket = 0;
};

}  // namespace

struct cae_door {
    Model *enc = nullptr, *dec = nullptr;
    int dev = 0, max_batch = 32;
    hipStream_t main = nullptr, up = nullptr;
    std::mutex mu;  // queues, flights, stop
    std::condition_variable cv_work, cv_flight, cv_done;
    std::deque<Request *> q;
    std::deque<Flight *> cq;
    std::vector<std::unique_ptr<Flight>> flights;
    bool stop = false, dispatcher_gone = false, hold = false;
    std::thread dispatcher, completer;
    PinnedPool staging{64}, results{8};
    // statistics (seconds summed over chunks; guarded by stat_mu)
    std::mutex stat_mu;
    double st[CAE_DOOR_STATS] = {0};

    void add_stat(int k, double v) {
        std::lock_guard<std::mutex> lk(stat_mu);
        st[k] += v;
    }
    int launch(Flight *f);
    int complete(Flight *f);
    void finish(Flight *f, int rc);
    void dispatch_loop();
    void complete_loop();
    int submit(Request *r);
    void release(Request *r);
};

namespace {

int grow_device(void **p, size_t *have, size_t need) {
    if (*have >= need) return CAE_OK;
    if (*p) {
        (void)hipFree(*p);  // (the flight is idle: its last batch was pulled before it was freed)
        *p = nullptr;
        *have = 0;
    }
    need = (need + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    HIP_TRY(hipMalloc(p, need));
    *have = need;
    return CAE_OK;
}

inline int latent_dim(int x, int levels) {
    for (int i = 0; i < levels; ++i) x = (x + 1) / 2;
    return x;
}

}  // namespace

// launches one batch on `main`: H2D of the callers' pinned chunks on the side stream, then the fused track
int cae_door::launch(Flight *f) {
    const int n = (int)f->batch.size();
    Request *r0 = f->batch[0];
    int rc;
    if ((rc = grow_device(&f->d_in, &f->d_in_bytes, (size_t)n * r0->in_bytes))) return rc;
    if ((rc = grow_device(&f->d_out, &f->d_out_bytes, (size_t)n * r0->out_bytes))) return rc;
    for (int i = 0; i < n; ++i)
        HIP_TRY(hipMemcpyAsync((char *)f->d_in + (size_t)i * r0->in_bytes, f->batch[i]->staged.ptr, r0->in_bytes,
                               hipMemcpyHostToDevice, up));
    HIP_TRY(hipEventRecord(f->landed, up));
    HIP_TRY(hipStreamWaitEvent(main, f->landed, 0));
    if (r0->kind == kEncode)
        rc = cae_analysis_symbols(reinterpret_cast<cae_model_t *>(enc), f->d_in, CAE_FMT_U8_HWC, n, r0->h, r0->w,
                                  (int32_t *)f->d_out, main);
    else
        rc = cae_synthesis_symbols(reinterpret_cast<cae_model_t *>(dec), (const int32_t *)f->d_in, n, r0->h, r0->w,
                                   f->d_out, CAE_FMT_U8_HWC, main);
    if (rc) return rc;
    f->ticket = cae_last_range_ticket();
    HIP_TRY(hipEventRecord(f->ready, main));
    return CAE_OK;
}

// waits for the flight, repeats it on the fp32 kernels if a value left the f16 range, pulls the results
int cae_door::complete(Flight *f) {
    const int n = (int)f->batch.size();
    Request *r0 = f->batch[0];
    HIP_TRY(hipEventSynchronize(f->ready));
    const auto t_ready = Clock::now();
    if (f->ticket) {
        Model *m = r0->kind == kEncode ? enc : dec;
        int over = 0, rc = cae_range_check(reinterpret_cast<cae_model_t *>(m), f->ticket, &over);
        if (rc) return rc;
        if (over) {  // f16x3 range guard (include/cae_hip.h): the batch again on the exact-fp32 kernels
            cae_thread_force_fp32(1);
            if (r0->kind == kEncode)
                rc = cae_analysis_symbols(reinterpret_cast<cae_model_t *>(enc), f->d_in, CAE_FMT_U8_HWC, n, r0->h, r0->w,
                                          (int32_t *)f->d_out, main);
            else
                rc = cae_synthesis_symbols(reinterpret_cast<cae_model_t *>(dec), (const int32_t *)f->d_in, n, r0->h,
                                           r0->w, f->d_out, CAE_FMT_U8_HWC, main);
            cae_thread_force_fp32(0);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(f->ready, main));
            HIP_TRY(hipEventSynchronize(f->ready));
            add_stat(CAE_DOOR_STAT_FP32_REPEATS, 1.0);
        }
    }
    auto *ob = new OutBuf;
    int rc = results.get((size_t)n * r0->out_bytes, &ob->mem);
    if (rc == CAE_OK) rc = cae_copy_to_host(ob->mem.ptr, f->d_out, (size_t)n * r0->out_bytes);
    if (rc) {
        results.put(ob->mem);
        delete ob;
        return rc;
    }
    const auto t_pulled = Clock::now();
    ob->users.store(n);
    for (int i = 0; i < n; ++i) {
        Request *r = f->batch[i];
        r->t_ready = t_ready;
        r->t_pulled = t_pulled;
        r->result = (const uint8_t *)ob->mem.ptr + (size_t)i * r0->out_bytes;
        r->out = ob;
    }
    return CAE_OK;
}

// hands the flight's answers (or its error) to the callers and frees the flight
void cae_door::finish(Flight *f, int rc) {
    const std::string err = rc ? cae_last_error() : "";
    std::vector<Request *> batch;
    batch.swap(f->batch);
    {
        std::lock_guard<std::mutex> lk(mu);
        f->busy = false;
    }
    cv_flight.notify_one();
    for (Request *r : batch) {
        // notified under the request's lock: the request lives on its caller's stack and is gone once the caller
        // has seen `done`, which it cannot before this lock is released
        std::lock_guard<std::mutex> lk(r->mu);
        r->rc = rc;
        r->err = err;
        r->done = true;
        r->cv.notify_one();
    }
}

void cae_door::dispatch_loop() {
    (void)hipSetDevice(dev);
    for (;;) {
        Flight *f = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [&] { return stop || (!q.empty() && !hold); });
            if (q.empty()) break;  // stop, nothing pending
            // requests keep piling up while every flight is on the device
            cv_flight.wait(lk, [&] {
                for (auto &fl : flights)
                    if (!fl->busy) return true;
                return false;
            });
            for (auto &fl : flights)
                if (!fl->busy) {
                    f = fl.get();
                    break;
                }
            f->busy = true;
            Request *first = q.front();
            while (!q.empty() && (int)f->batch.size() < max_batch && q.front()->same_key(*first)) {
                f->batch.push_back(q.front());  // another shape / direction: next flight
                q.pop_front();
            }
        }
        const auto t0 = Clock::now();
        for (Request *r : f->batch) r->t_picked = t0;
        const int rc = launch(f);
        const auto t1 = Clock::now();
        for (Request *r : f->batch) r->t_launched = t1;
        {
            std::lock_guard<std::mutex> lk(stat_mu);
            st[CAE_DOOR_STAT_BATCHES] += 1.0;
            st[CAE_DOOR_STAT_CHUNKS] += (double)f->batch.size();
        }
        if (rc) {
            (void)hipStreamSynchronize(main);  // whatever was queued of this flight is done before its buffers go on
            finish(f, rc);
            continue;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            cq.push_back(f);
        }
        cv_done.notify_one();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        dispatcher_gone = true;
    }
    cv_done.notify_one();
}

void cae_door::complete_loop() {
    (void)hipSetDevice(dev);
    for (;;) {
        Flight *f = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return dispatcher_gone || !cq.empty(); });
            if (cq.empty()) break;
            f = cq.front();
            cq.pop_front();
        }
        const int rc = complete(f);
        if (rc) (void)hipStreamSynchronize(main);  // nothing of this flight is running when its buffers are reused
        finish(f, rc);
    }
}

int cae_door::submit(Request *r) {
    r->t_submit = Clock::now();
    {
        std::lock_guard<std::mutex> lk(mu);
        if (stop) return fail(CAE_ERR_ARG, "codec front door is closed");
        q.push_back(r);
    }
    cv_work.notify_one();
    {
        std::unique_lock<std::mutex> lk(r->mu);
        r->cv.wait(lk, [&] { return r->done; });
    }
    if (r->rc) return fail(r->rc, "%s", r->err.c_str());
    const auto now = Clock::now();
    std::lock_guard<std::mutex> lk(stat_mu);
    st[CAE_DOOR_STAT_T_QUEUE] += seconds(r->t_submit, r->t_picked);
    st[CAE_DOOR_STAT_T_LAUNCH] += seconds(r->t_picked, r->t_launched);
    st[CAE_DOOR_STAT_T_DEVICE] += seconds(r->t_launched, r->t_ready);
    st[CAE_DOOR_STAT_T_PULL] += seconds(r->t_ready, r->t_pulled);
    st[CAE_DOOR_STAT_T_WAKE] += seconds(r->t_pulled, now);
    return CAE_OK;
}

void cae_door::release(Request *r) {
    if (r->out) {
        if (r->out->users.fetch_sub(1) == 1) {
            results.put(r->out->mem);
            delete r->out;
        }
        r->out = nullptr;
    }
    staging.put(r->staged);
    r->staged = Pinned{};
}

extern "C" {

int cae_door_create(cae_model_t *analysis, cae_model_t *synthesis, int max_batch, int inflight, cae_door_t **out) {
    if (!analysis || !synthesis || !out) return fail(CAE_ERR_ARG, "NULL argument");
    Model *e = reinterpret_cast<Model *>(analysis), *d = reinterpret_cast<Model *>(synthesis);
    if (e->L != d->L || e->c_org != d->c_org || e->c_bn != d->c_bn)
        return fail(CAE_ERR_ARG, "analysis and synthesis handles describe different models");
    if (e->L < 1 || e->L > 16) return fail(CAE_ERR_ARG, "bad compression level %d", e->L);
    auto door = std::make_unique<cae_door>();
    door->enc = e;
    door->dec = d;
    door->max_batch = std::max(1, std::min(max_batch > 0 ? max_batch : 32, 1024));
    inflight = std::max(1, std::min(inflight > 0 ? inflight : 3, 16));
    HIP_TRY(hipGetDevice(&door->dev));
    HIP_TRY(hipStreamCreateWithFlags(&door->main, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&door->up, hipStreamNonBlocking));
    for (int i = 0; i < inflight; ++i) {
        auto f = std::make_unique<Flight>();
        HIP_TRY(hipEventCreateWithFlags(&f->landed, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&f->ready, hipEventDisableTiming | hipEventBlockingSync));
        door->flights.push_back(std::move(f));
    }
    cae_door *p = door.release();
    p->dispatcher = std::thread([p] { p->dispatch_loop(); });
    p->completer = std::thread([p] { p->complete_loop(); });
    *out = p;
    return CAE_OK;
}

void cae_door_destroy(cae_door_t *door) {
    if (!door) return;
    {
        std::lock_guard<std::mutex> lk(door->mu);
        door->stop = true;
    }
    door->cv_work.notify_all();
    if (door->dispatcher.joinable()) door->dispatcher.join();  // serves what is queued, then leaves
    if (door->completer.joinable()) door->completer.join();
    (void)hipSetDevice(door->dev);
    (void)hipStreamSynchronize(door->main);
    (void)hipStreamSynchronize(door->up);
    for (Model *m : {door->enc, door->dec}) {  // the handles outlive the door: forget its stream
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->last_stream == (void *)door->main) m->last_stream_set = false;
    }
    for (auto &f : door->flights) {
        if (f->d_in) (void)hipFree(f->d_in);
        if (f->d_out) (void)hipFree(f->d_out);
        (void)hipEventDestroy(f->landed);
        (void)hipEventDestroy(f->ready);
    }
    (void)hipStreamDestroy(door->main);
    (void)hipStreamDestroy(door->up);
    delete door;
}

int cae_door_encode(cae_door_t *door, const uint8_t *tile, int h, int w, int c, uint8_t **out, size_t *out_len) {
    if (!door || !tile || !out || !out_len) return fail(CAE_ERR_ARG, "NULL argument");
    *out = nullptr;
    *out_len = 0;
    Model *m = door->enc;
    if (c != m->c_org) return fail(CAE_ERR_ARG, "expected uint8 (h,w,%d), got %d channels", m->c_org, c);
    if (h < 2 || w < 2) return fail(CAE_ERR_ARG, "bad tile %dx%dx%d", h, w, c);
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    HIP_TRY(hipSetDevice(door->dev));
    const auto t0 = Clock::now();
    Request r;
    r.kind = kEncode;
    r.h = h;
    r.w = w;
    const int lh = latent_dim(h, m->L), lw = latent_dim(w, m->L);
    r.in_bytes = (size_t)h * w * c;
    r.out_bytes = (size_t)m->c_bn * lh * lw * sizeof(int32_t);
    int rc = door->staging.get(r.in_bytes, &r.staged);
    if (rc) return rc;
    memcpy(r.staged.ptr, tile, r.in_bytes);  // into pinned memory, in the caller's thread
    const auto t1 = Clock::now();
    rc = door->submit(&r);
    const auto t2 = Clock::now();
    if (rc == CAE_OK) {
        // range coder in the caller's thread, straight from the batch's pinned buffer; 16 bytes of headroom = the header
        rc = rans_encode_chunk(m->ent, (const int32_t *)r.result, lh * lw, 16, out, out_len);
        if (rc == CAE_OK) {
            const uint64_t hw[2] = {(uint64_t)h, (uint64_t)w};  // struct.pack('>QQ', h, w)  (_autoencoders.py:553)
            for (int k = 0; k < 2; ++k)
                for (int b = 0; b < 8; ++b) (*out)[8 * k + b] = (uint8_t)(hw[k] >> (56 - 8 * b));
        }
    }
    door->release(&r);
    const auto t3 = Clock::now();
    std::lock_guard<std::mutex> lk(door->stat_mu);
    door->st[CAE_DOOR_STAT_T_STAGE] += seconds(t0, t1);
    door->st[CAE_DOOR_STAT_T_WAIT] += seconds(t1, t2);
    door->st[CAE_DOOR_STAT_T_CODE] += seconds(t2, t3);
    return rc;
}

int cae_door_decode_shape(cae_door_t *door, const uint8_t *chunk, size_t len, int *h, int *w, int *c) {
    if (!door || !chunk || !h || !w || !c) return fail(CAE_ERR_ARG, "NULL argument");
    if (len < 16) return fail(CAE_ERR_CORRUPT, "chunk shorter than its 16-byte header");
    uint64_t hw[2] = {0, 0};
    for (int k = 0; k < 2; ++k)
        for (int b = 0; b < 8; ++b) hw[k] = (hw[k] << 8) | chunk[8 * k + b];
    const int L = door->dec->L;
    // latent size as the reference's decode derives it (floor, _autoencoders.py:565)
    const uint64_t lh = hw[0] >> L, lw = hw[1] >> L;
    if (lh < 1 || lw < 1 || hw[0] > (1u << 20) || hw[1] > (1u << 20))
        return fail(CAE_ERR_CORRUPT, "chunk header names a %llu x %llu tile", (unsigned long long)hw[0],
                    (unsigned long long)hw[1]);
    *h = (int)(lh << L);
    *w = (int)(lw << L);
    *c = door->dec->c_org;
    return CAE_OK;
}

int cae_door_decode(cae_door_t *door, const uint8_t *chunk, size_t len, uint8_t *out, size_t out_capacity) {
    int H, W, C;
    int rc = cae_door_decode_shape(door, chunk, len, &H, &W, &C);
    if (rc) return rc;
    if (!out) return fail(CAE_ERR_ARG, "NULL argument");
    Model *m = door->dec;
    if (m->ent.channels == 0) return fail(CAE_ERR_ARG, "entropy model not set");
    if (out_capacity < (size_t)H * W * C)
        return fail(CAE_ERR_ARG, "output buffer of %zu bytes for a %dx%dx%d tile", out_capacity, H, W, C);
    HIP_TRY(hipSetDevice(door->dev));
    const auto t0 = Clock::now();
    Request r;
    r.kind = kDecode;
    r.h = H >> m->L;
    r.w = W >> m->L;
    r.in_bytes = (size_t)m->c_bn * r.h * r.w * sizeof(int32_t);
    r.out_bytes = (size_t)H * W * C;
    rc = door->staging.get(r.in_bytes, &r.staged);
    if (rc) return rc;
    // range decoder in the caller's thread, straight into pinned memory
    rc = rans_decode_chunk(m->ent, chunk + 16, len - 16, r.h * r.w, (int32_t *)r.staged.ptr);
    const auto t1 = Clock::now();
    if (rc == CAE_OK) rc = door->submit(&r);
    const auto t2 = Clock::now();
    if (rc == CAE_OK) memcpy(out, r.result, r.out_bytes);
    door->release(&r);
    const auto t3 = Clock::now();
    std::lock_guard<std::mutex> lk(door->stat_mu);
    door->st[CAE_DOOR_STAT_T_STAGE] += seconds(t0, t1);
    door->st[CAE_DOOR_STAT_T_WAIT] += seconds(t1, t2);
    door->st[CAE_DOOR_STAT_T_CODE] += seconds(t2, t3);
    return rc;
}

int cae_door_hold(cae_door_t *door, int on) {
    if (!door) return fail(CAE_ERR_ARG, "NULL argument");
    {
        std::lock_guard<std::mutex> lk(door->mu);
        door->hold = on != 0;
    }
    door->cv_work.notify_all();
    return CAE_OK;
}

int cae_door_stats(cae_door_t *door, double *stats, int n, int reset) {
    if (!door || !stats || n < 0) return fail(CAE_ERR_ARG, "NULL argument");
    size_t queued;
    {
        std::lock_guard<std::mutex> lk(door->mu);
        queued = door->q.size();
    }
    std::lock_guard<std::mutex> lk(door->stat_mu);
    door->st[CAE_DOOR_STAT_QUEUED] = (double)queued;
    for (int i = 0; i < n; ++i) stats[i] = i < CAE_DOOR_STATS ? door->st[i] : 0.0;
    if (reset)
        for (double &v : door->st) v = 0.0;
    return CAE_OK;
}

}  // extern "C"
