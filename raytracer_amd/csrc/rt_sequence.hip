// rt_sequence.hip -- the frame loop of driver.c:28-203 as one call: for every frame of the orbit
// the per-frame set-up (driver.c:34-107), the vertex transform (:110), the pixel loop (:126-193)
// and, optionally, the picture written as pic%04d.bmp (:196-199) -- pipelined: while frame f is
// on the GPU, frame f-1 is on its way to the host on a copy stream and earlier frames are being
// encoded and written by host threads.
//
//   GPU (library stream) : transform, sweeps, shading, placement of frame f into screen[b]
//   copy stream          : screen[b] -> pinned host[b]                       (frame f-1)
//   writer threads       : host[b] -> BMP file                               (frames f-2, ...)
//
// A ring of RING buffers decouples the three; a buffer returns to the GPU side when its file
// is closed.  Frames can be taken with a stride, which is how a sequence is split over ranks
// without any exchange (rank r of n renders frames first + r, first + r + n, ...).
#include "rt_internal.h"

#include <condition_variable>
#include <chrono>
#include <deque>
#include <mutex>
#include <thread>

namespace {

constexpr int RING = 6;
constexpr int WRITERS = 4;

struct Job {
    int buf, frame;
};

struct Pipeline {
    std::mutex m;
    std::condition_variable cv_job, cv_free;
    std::deque<Job> jobs;
    bool free_buf[RING];
    bool stop = false;
    int rc = RT_OK;
    std::string error;
    double write_seconds = 0;
};

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" int rt_render_sequence(rt_scene* sc, int first_frame, int count, int frame_step,
                                  int frames_per_turn, int W, int H, double tan_half,
                                  const char* bmp_pattern, rt_sequence_stats* stats)
{
    RT_ARG(sc, "null scene");
    RT_ARG(count >= 0 && frame_step >= 1 && frames_per_turn >= 1, "bad frame range");
    RT_ARG(W > 0 && H > 0 && tan_half > 0, "bad image size");
    if (stats) *stats = rt_sequence_stats{};
    if (count == 0) return RT_OK;
    if (rt_ensure_init() != RT_OK) return RT_E_NODEVICE;

    const size_t bytes = (size_t)W * H * 3;
    const bool write = bmp_pattern != nullptr;
    const int ring = write ? RING : 1;
    uint8_t* rgb8 = nullptr;
    uint8_t* screen[RING] = {};
    uint8_t* host[RING] = {};
    hipEvent_t copied[RING] = {};
    hipStream_t copy_stream = nullptr;
    Pipeline pl;
    std::vector<std::thread> writers;
    int rc = RT_OK;
    auto fail = [&](int code, const char* what) {
        rt_set_error("rt_render_sequence: %s", what);
        rc = code;
    };

    if (hipMalloc((void**)&rgb8, bytes) != hipSuccess) fail(RT_E_HIP, "device allocation failed");
    for (int b = 0; b < ring && rc == RT_OK; ++b) {
        if (hipMalloc((void**)&screen[b], bytes) != hipSuccess) fail(RT_E_HIP, "device allocation failed");
        if (write && rc == RT_OK) {
            if (hipHostMalloc((void**)&host[b], bytes, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&copied[b], hipEventDisableTiming) != hipSuccess)
                fail(RT_E_HIP, "pinned allocation failed");
        }
        pl.free_buf[b] = true;
    }
    if (write && rc == RT_OK &&
        hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess)
        fail(RT_E_HIP, "stream creation failed");

    if (write && rc == RT_OK) {
        const std::string pattern(bmp_pattern);
        for (int t = 0; t < WRITERS; ++t)
            writers.emplace_back([&, pattern] {
                for (;;) {
                    Job j;
                    {
                        std::unique_lock<std::mutex> lk(pl.m);
                        pl.cv_job.wait(lk, [&] { return pl.stop || !pl.jobs.empty(); });
                        if (pl.jobs.empty()) return;
                        j = pl.jobs.front();
                        pl.jobs.pop_front();
                    }
                    const double t0 = now();
                    int wrc = RT_OK;
                    std::string msg;
                    if (hipEventSynchronize(copied[j.buf]) != hipSuccess) {
                        wrc = RT_E_HIP;
                        msg = "copy to host failed";
                    } else {
                        char path[4096];
                        snprintf(path, sizeof path, pattern.c_str(), j.frame);
                        wrc = rt_save_bmp(path, host[j.buf], W, H);
                        if (wrc != RT_OK) msg = rt_last_error();      // this thread's message
                    }
                    std::lock_guard<std::mutex> lk(pl.m);
                    pl.write_seconds += now() - t0;
                    if (wrc != RT_OK && pl.rc == RT_OK) { pl.rc = wrc; pl.error = msg; }
                    pl.free_buf[j.buf] = true;
                    pl.cv_free.notify_all();
                }
            });
    }

    const double t_start = now();
    double t_render = 0, t_wait = 0;
    int done = 0;
    for (int k = 0; k < count && rc == RT_OK; ++k) {
        const int frame = first_frame + k * frame_step;
        // a free buffer of the ring
        int b = 0;
        if (write) {
            const double t0 = now();
            std::unique_lock<std::mutex> lk(pl.m);
            pl.cv_free.wait(lk, [&] {
                for (int i = 0; i < ring; ++i)
                    if (pl.free_buf[i]) return true;
                return false;
            });
            for (int i = 0; i < ring; ++i)
                if (pl.free_buf[i]) { b = i; break; }
            pl.free_buf[b] = false;
            if (pl.rc != RT_OK) { rc = pl.rc; rt_set_error("%s", pl.error.c_str()); }
            t_wait += now() - t0;
            if (rc != RT_OK) break;
        }
        const double t0 = now();
        rt_frame f;
        if ((rc = rt_driver_frame_setup(frame, frames_per_turn, &f)) != RT_OK) break;      // driver.c:34-107
        if ((rc = rt_scene_transform(sc, f.obmat)) != RT_OK) break;                         // driver.c:110
        if (hipMemsetAsync(screen[b], 0, bytes, rt_g().stream) != hipSuccess) {             // driver.c:31-32
            fail(RT_E_HIP, "clear failed");
            break;
        }
        if ((rc = rt_render_rows(sc, &f, W, H, tan_half, 0, H, nullptr, nullptr, nullptr, rgb8)) != RT_OK) break;
        if ((rc = rt_place_rows(rgb8, W, H, 0, H, screen[b])) != RT_OK) break;              // blocks until done
        t_render += now() - t0;
        ++done;
        if (write) {
            if (hipMemcpyAsync(host[b], screen[b], bytes, hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
                hipEventRecord(copied[b], copy_stream) != hipSuccess) {
                fail(RT_E_HIP, "copy to host failed");
                break;
            }
            std::lock_guard<std::mutex> lk(pl.m);
            pl.jobs.push_back(Job{b, frame});
            pl.cv_job.notify_one();
        }
    }

    // drain
    if (!writers.empty()) {
        {
            std::unique_lock<std::mutex> lk(pl.m);
            pl.cv_free.wait(lk, [&] { return pl.jobs.empty(); });
            pl.stop = true;
        }
        pl.cv_job.notify_all();
        for (std::thread& t : writers) t.join();
        if (rc == RT_OK && pl.rc != RT_OK) { rc = pl.rc; rt_set_error("%s", pl.error.c_str()); }
    }
    const double t_total = now() - t_start;
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    for (int b = 0; b < RING; ++b) {
        if (copied[b]) (void)hipEventDestroy(copied[b]);
        if (host[b]) (void)hipHostFree(host[b]);
        if (screen[b]) (void)hipFree(screen[b]);
    }
    if (rgb8) (void)hipFree(rgb8);
    if (stats) {
        stats->frames = done;
        stats->seconds_total = t_total;
        stats->seconds_render = t_render;
        stats->seconds_waiting_for_writers = t_wait;
        stats->seconds_writing = pl.write_seconds;
    }
    return rc;
}
