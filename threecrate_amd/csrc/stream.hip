// LiDAR frame streaming: a bounded queue of host frames feeding the per-frame pipeline
//   voxel_grid_filter -> estimate_normals(previous frame) -> icp_point_to_plane(current -> previous)
// on one GPU, with the host->device copy of frame i+1 overlapped with the compute of frame i.
//
// Models threecrate_algorithms::streaming::RealtimePipeline (threecrate-algorithms/src/streaming.rs:
// 540-646): `send` blocks the producer when `max_queue_depth` frames are waiting (backpressure),
// `try_send` drops the frame instead and counts it, `finish` closes the input, drains the queue, joins
// the worker and returns the output (here: one registration result per consecutive frame pair) and
// the metrics.  The KITTI reader follows VelodyneKittiBinReader::read (threecrate-io/src/lidar.rs:
// 310-343): 16-byte little-endian records x, y, z, intensity; the intensity is dropped.
//
// Threads: the producer (caller) and one worker.  The worker is the only user of the tc_context while
// the stream exists.  Frames are copied into pinned slots by the producer, so the caller's buffer is
// free as soon as send returns.  Two HIP streams: the context's (compute) and a copy stream; an event
// orders "frame landed" before the first kernel that reads it.
#include "tc_internal.h"

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

struct tc_frame_stream {
    tc_context *ctx = nullptr;
    tc_frame_stream_config cfg{};
    // slots: pinned host buffers
    struct Slot { float *host = nullptr; size_t n = 0; };
    std::vector<Slot> slots;
    std::deque<int> free_slots, ready;
    std::mutex mu;
    std::condition_variable cv_free, cv_ready;
    bool closed = false;
    // worker
    std::thread worker;
    hipStream_t copy_stream = nullptr;
    hipEvent_t landed[2] = {nullptr, nullptr};
    float *d_raw[2] = {nullptr, nullptr};       // frames as received
    float *d_frame[2] = {nullptr, nullptr};     // filtered: current / previous
    tc_cloud *prev_h = nullptr;                 // the previous frame: points + index + cell-sorted normals
    std::vector<tc_frame_result> results;
    tc_frame_stream_metrics metrics{};
    tc_status worker_status = TC_OK;
};

namespace {

void release_slot(tc_frame_stream *s, int slot) {
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->free_slots.push_back(slot);
    }
    s->cv_free.notify_one();
}

void worker_body(tc_frame_stream *s);

// The thread body: an exception that escaped it would be std::terminate for the whole process (the caller may be a Rust or Python
// host).  The stream fails instead: status TC_GPU for tc_frame_stream_finish, the queue closed, the queued frames dropped
// and every sender that waits for a slot woken (enqueue returns TC_INVALID_DATA on a closed stream).
void worker_main(tc_frame_stream *s) {
    try {
        tc::fault_point("stream_worker");
        worker_body(s);
    } catch (...) {
        s->worker_status = TC_GPU;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->closed = true;
            s->ready.clear();          // (noexcept; the queued frames are dropped with the stream)
        }
        s->cv_free.notify_all();
    }
}

// registration of consecutive frames; runs until the input is closed and drained
void worker_body(tc_frame_stream *s) {
    tc_context *ctx = s->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) { s->worker_status = TC_GPU; return; }
    const tc_frame_stream_config &c = s->cfg;
    int cur = 0;                    // index into d_frame: current; the other one holds the previous frame
    bool have_prev = false;
    int pre_slot = -1, pre_buf = -1;            // a frame whose copy has already been issued
    int raw = 0;
    for (;;) {
        int slot = -1, buf = -1;
        if (pre_slot >= 0) {
            slot = pre_slot; buf = pre_buf; pre_slot = -1;
        } else {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv_ready.wait(lk, [&] { return !s->ready.empty() || s->closed; });
            if (s->ready.empty()) break;                       // closed and drained
            slot = s->ready.front(); s->ready.pop_front();
            lk.unlock();
            buf = raw; raw ^= 1;
            (void)hipMemcpyAsync(s->d_raw[buf], s->slots[slot].host, s->slots[slot].n * 3 * sizeof(float), hipMemcpyHostToDevice,
                                 s->copy_stream);
            (void)hipEventRecord(s->landed[buf], s->copy_stream);
        }
        const size_t n_raw = s->slots[slot].n;
        // start the next frame's copy before computing on this one (it lands while the kernels run)
        {
            std::unique_lock<std::mutex> lk(s->mu);
            if (!s->ready.empty()) {
                pre_slot = s->ready.front(); s->ready.pop_front();
                lk.unlock();
                pre_buf = raw; raw ^= 1;
                (void)hipMemcpyAsync(s->d_raw[pre_buf], s->slots[pre_slot].host, s->slots[pre_slot].n * 3 * sizeof(float),
                                     hipMemcpyHostToDevice, s->copy_stream);
                (void)hipEventRecord(s->landed[pre_buf], s->copy_stream);
            }
        }
        (void)hipStreamWaitEvent(ctx->stream, s->landed[buf], 0);
        tc_frame_result r{};
        r.status = TC_OK;
        r.n_points_in = n_raw;
        size_t n_cur = n_raw;
        tc_status st = TC_OK;
        if (c.voxel_size > 0.0f) {
            st = tc_voxel_grid_filter_device(ctx, s->d_raw[buf], n_raw, c.voxel_size, s->d_frame[cur], &n_cur);
        } else if (hipMemcpyAsync(s->d_frame[cur], s->d_raw[buf], n_raw * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
            st = TC_GPU;
        }
        // the raw frame has been consumed (the filter call is blocking; the plain copy is stream ordered
        // before anything that could overwrite d_raw[buf]: its next writer waits for this slot's release)
        if (c.voxel_size <= 0.0f) (void)hipStreamSynchronize(ctx->stream);
        release_slot(s, slot);
        r.n_points = n_cur;
        // Every frame lives in a cloud handle (tc_cloud_*): it is indexed ONCE, when its normals are estimated, and that index +
        // the cell-sorted normals are what the NEXT frame registers against (the handle-free calls would index it twice).
        tc_cloud *cur_h = nullptr;
        if (st == TC_OK) st = tc_cloud_upload_device(ctx, s->d_frame[cur], n_cur, &cur_h);
        if (st == TC_OK) {
            // its normals first (the next registration needs them): the index built for them also orders this frame as the
            // SOURCE of the registration below
            tc_normal_config nc;
            tc_normal_config_default(&nc);
            nc.k_neighbors = c.k_neighbors;
            st = tc_cloud_estimate_normals_device(cur_h, &nc, nullptr);
        }
        if (st == TC_OK && have_prev) {
            tc_icp_result ir{};
            const float ident[7] = {0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f};
            st = tc_cloud_icp_point_to_plane(cur_h, s->prev_h, ident, c.max_iterations, c.max_correspondence_distance, c.convergence_threshold, &ir);
            std::memcpy(r.transformation, ir.transformation, sizeof(r.transformation));
            r.mse = ir.mse; r.iterations = ir.iterations; r.converged = ir.converged;
        }
        if (st == TC_OK) {
            tc_cloud_destroy(s->prev_h);              // (its blocks go back to the context's pool: no allocation per frame)
            s->prev_h = cur_h;
        } else {
            tc_cloud_destroy(cur_h);
        }
        r.status = st;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            if (have_prev || st != TC_OK) s->results.push_back(r);
            s->metrics.items_processed += 1;
        }
        if (st == TC_OK) { have_prev = true; cur ^= 1; }
    }
}

tc_status enqueue(tc_frame_stream *s, const float *frame, size_t n, size_t stride, bool block, int *accepted) {
    if (!s || !frame || (stride != 3 && stride != 4)) return TC_INVALID_DATA;
    // plain status codes here: the context's error string belongs to the worker thread while the stream runs
    if (n == 0 || n > s->cfg.max_points) return TC_INVALID_DATA;          // frame is empty or larger than max_points
    int slot = -1;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        if (s->closed) return TC_INVALID_DATA;                            // pipeline already finished (streaming.rs:583-588)
        if (s->free_slots.empty()) {
            if (!block) {
                s->metrics.items_dropped += 1;
                if (accepted) *accepted = 0;
                return TC_OK;
            }
            s->cv_free.wait(lk, [&] { return !s->free_slots.empty() || s->closed; });
            // closed while waiting (finished, or the worker failed): no worker will look at the queue again, so a frame must not be
            // accepted even when a slot happens to be free at this moment
            if (s->closed || s->free_slots.empty()) return TC_INVALID_DATA;
        }
        slot = s->free_slots.front(); s->free_slots.pop_front();
    }
    float *dst = s->slots[slot].host;
    if (stride == 3) {
        std::memcpy(dst, frame, n * 3 * sizeof(float));
    } else {
        for (size_t i = 0; i < n; ++i) { dst[3 * i] = frame[4 * i]; dst[3 * i + 1] = frame[4 * i + 1]; dst[3 * i + 2] = frame[4 * i + 2]; }
    }
    s->slots[slot].n = n;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->ready.push_back(slot);
        s->metrics.items_queued += 1;
        s->metrics.max_depth_seen = std::max<uint64_t>(s->metrics.max_depth_seen, s->ready.size());
    }
    s->cv_ready.notify_one();
    if (accepted) *accepted = 1;
    return TC_OK;
}

}   // namespace

extern "C" {

tc_status tc_frame_stream_create(tc_context *ctx, const tc_frame_stream_config *cfg, tc_frame_stream **out) try {
    if (!ctx || !cfg || !out) return TC_INVALID_DATA;
    *out = nullptr;
    if (cfg->max_points == 0 || cfg->max_queue_depth == 0) return tc::fail(ctx, TC_INVALID_DATA, "max_points and max_queue_depth must be >= 1");
    if (cfg->max_iterations == 0) return tc::fail(ctx, TC_INVALID_DATA, "max_iterations must be > 0");
    if (cfg->k_neighbors < 3) return tc::fail(ctx, TC_INVALID_DATA, "k_neighbors must be >= 3");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    tc_frame_stream *s = new tc_frame_stream();
    s->ctx = ctx;
    s->cfg = *cfg;
    // ONE owner of the half-built stream: bail destroys it and forgets it before it builds the message -- tc::fail can throw
    // (std::string), and the catch below must not destroy the stream a second time (use after free + double hipHostFree)
    auto bail = [&](const char *what) {
        tc_frame_stream *dead = s;
        s = nullptr;
        tc_frame_stream_destroy(dead);
        return tc::fail(ctx, TC_GPU, std::string("tc_frame_stream_create: ") + what);
    };
    const size_t bytes = cfg->max_points * 3 * sizeof(float);
    try {          // (vector / deque growth and the thread's start can throw: the half-built stream must not leak its pinned slots)
    s->slots.resize(cfg->max_queue_depth);
    for (size_t i = 0; i < s->slots.size(); ++i) {
        if (hipHostMalloc((void **)&s->slots[i].host, bytes, hipHostMallocDefault) != hipSuccess) return bail("pinned frame slot");
        s->free_slots.push_back((int)i);
    }
    if (hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) != hipSuccess) return bail("copy stream");
    for (int b = 0; b < 2; ++b) {
        if (hipEventCreateWithFlags(&s->landed[b], hipEventDisableTiming) != hipSuccess) return bail("event");
        if (hipMalloc((void **)&s->d_raw[b], bytes) != hipSuccess) return bail("device frame");
        if (hipMalloc((void **)&s->d_frame[b], bytes) != hipSuccess) return bail("device frame");
    }
    s->worker = std::thread(worker_main, s);
    } catch (...) { if (s) tc_frame_stream_destroy(s); throw; }
    *out = s;
    return TC_OK;
} TC_CATCH_STATUS(ctx)

tc_status tc_frame_stream_send(tc_frame_stream *s, const float *frame, size_t n, size_t stride_floats) try {
    return enqueue(s, frame, n, stride_floats, true, nullptr);
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_frame_stream_try_send(tc_frame_stream *s, const float *frame, size_t n, size_t stride_floats, int *accepted) try {
    return enqueue(s, frame, n, stride_floats, false, accepted);
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_frame_stream_finish(tc_frame_stream *s, tc_frame_result *results, size_t capacity, size_t *n_results,
                                 tc_frame_stream_metrics *metrics) try {
    if (!s) return TC_INVALID_DATA;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->closed = true;
    }
    s->cv_ready.notify_all();
    if (s->worker.joinable()) s->worker.join();
    const size_t n = std::min(capacity, s->results.size());
    if (results) std::memcpy(results, s->results.data(), n * sizeof(tc_frame_result));
    if (n_results) *n_results = s->results.size();
    if (metrics) *metrics = s->metrics;
    return s->worker_status;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

void tc_frame_stream_destroy(tc_frame_stream *s) try {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->closed = true;
    }
    s->cv_ready.notify_all();
    if (s->worker.joinable()) s->worker.join();
    (void)hipSetDevice(s->ctx->device);
    for (auto &sl : s->slots)
        if (sl.host) (void)hipHostFree(sl.host);
    for (int b = 0; b < 2; ++b) {
        if (s->landed[b]) (void)hipEventDestroy(s->landed[b]);
        if (s->d_raw[b]) (void)hipFree(s->d_raw[b]);
        if (s->d_frame[b]) (void)hipFree(s->d_frame[b]);
    }
    if (s->prev_h) tc_cloud_destroy(s->prev_h);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    delete s;
} TC_CATCH_VOID

tc_status tc_read_kitti_bin(const char *path, float *out_xyz, size_t capacity_points, size_t *n_points) try {
    if (!path || !n_points) return TC_INVALID_DATA;
    *n_points = 0;
    tc::fault_point("kitti");
    FILE *f = std::fopen(path, "rb");
    if (!f) return TC_INVALID_DATA;
    if (std::fseek(f, 0, SEEK_END) != 0) { std::fclose(f); return TC_INVALID_DATA; }
    const long size = std::ftell(f);
    std::rewind(f);
    if (size < 0 || size % 16 != 0) { std::fclose(f); return TC_INVALID_DATA; }      // lidar.rs:321-326
    const size_t n = (size_t)size / 16;
    *n_points = n;
    if (!out_xyz || capacity_points < n) { std::fclose(f); return out_xyz ? TC_INVALID_DATA : TC_OK; }   // size query
    float rec[4 * 256];
    size_t done = 0;
    while (done < n) {
        const size_t want = std::min<size_t>(256, n - done);
        if (std::fread(rec, 16, want, f) != want) { std::fclose(f); return TC_INVALID_DATA; }
        for (size_t i = 0; i < want; ++i) {               // x86 / gfx hosts are little endian, like the file
            out_xyz[3 * (done + i)] = rec[4 * i]; out_xyz[3 * (done + i) + 1] = rec[4 * i + 1]; out_xyz[3 * (done + i) + 2] = rec[4 * i + 2];
        }
        done += want;
    }
    std::fclose(f);
    return TC_OK;
} TC_CATCH_STATUS(nullptr)

}   // extern "C"
