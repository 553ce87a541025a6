// rt_multi.hip -- one frame on several GPUs of one node from a single process.
//
// The reference is a single-GPU program (triangle_intersection.cu:98-190 talks to one device);
// its pixel loop (driver.c:126-193) shards by rows because pixels are independent.  Here every
// device holds the full mesh in a scene of its own and renders one interleaved row part
// (rt_rows in the public header); the finished 8-bit rows travel to the first device over RCCL
// (grouped ncclSend / ncclRecv, i.e. xGMI point to point: every sender has its own link to the
// root) and are placed there with the reference's placement rule (driver.c:186-190).
//
// The library's calls block and wait for the device inside, so the devices are driven by one
// host thread each for the duration of a call; the interface stays blocking and single-caller.
// RCCL is loaded at run time (dlopen) when a set of more than one distinct device is created:
// programs that use one GPU do not depend on it, and a process that already holds an RCCL (a
// Python process with torch) shares that copy.
#include "rt_internal.h"

#include <dlfcn.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

namespace {

// ---- the part of RCCL's C API used here (rccl.h:236,260,339,700,722 and the group calls)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;           // ncclSuccess = 0
constexpr int kNcclUint8 = 1;       // ncclUint8 / ncclChar family: ncclInt8 = 0, ncclUint8 = 1
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
};

int load_rccl(Rccl& r)
{
    if (r.handle) return RT_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        rt_set_error("rt_multi: cannot load RCCL (librccl.so.1): %s", dlerror());
        return RT_E_STATE;
    }
    auto sym = [&](const char* name) { return dlsym(r.handle, name); };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GetErrorString || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) {
        rt_set_error("rt_multi: the RCCL library lacks a symbol this code needs");
        return RT_E_STATE;
    }
    return RT_OK;
}

// The host thread that drives one device for as long as the rt_multi lives: it makes its device
// current once and then runs the jobs the calls hand it (a thread per device and call cost its
// creation on every frame).
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;       // set by the caller, cleared by the worker
    bool has_job = false, done = false, quit = false;
};

struct Device {
    int ordinal = 0;
    rt_scene* scene = nullptr;
    DevBuf rgb8;                    // this device's rows, [rows of its part][W][3]
    hipStream_t comm_stream = nullptr;
    std::string error;              // message of a failed call on this device's thread
    int rc = RT_OK;
    Worker* worker = nullptr;
};

}  // namespace

struct rt_multi {
    std::vector<Device> dev;
    Rccl rccl;
    std::vector<ncclComm_t> comms;  // one per device when the exchange goes over RCCL
    bool use_rccl = false;
    std::string rccl_failure;       // why a set of distinct devices exchanges by peer copies after all
    std::string exchange;           // rt_multi_exchange's answer
    DevBuf gathered;                // on device 0: the rows of devices 1.. back to back
};

namespace {

void worker_main(Device* d)
{
    Worker& w = *d->worker;
    const bool device_ok = hipSetDevice(d->ordinal) == hipSuccess;
    for (;;) {
        std::function<int()> job;
        {
            std::unique_lock<std::mutex> lock(w.mu);
            w.cv.wait(lock, [&] { return w.has_job || w.quit; });
            if (w.quit) return;
            job = std::move(w.job);
        }
        int rc = RT_OK;
        std::string error;
        if (!device_ok) {
            rc = RT_E_HIP;
            error = "hipSetDevice failed";
        } else {
            rc = job();
            if (rc != RT_OK) error = rt_last_error();
        }
        {
            std::lock_guard<std::mutex> lock(w.mu);
            d->rc = rc;
            d->error = error;
            w.has_job = false;
            w.done = true;
        }
        w.cv.notify_all();
    }
}

int start_workers(rt_multi* m)
{
    for (Device& d : m->dev) {
        d.worker = new Worker();
        try {
            d.worker->th = std::thread(worker_main, &d);
        } catch (const std::system_error& e) {
            delete d.worker;
            d.worker = nullptr;
            rt_set_error("rt_multi: cannot start the thread of device %d: %s", d.ordinal, e.what());
            return RT_E_STATE;
        }
    }
    return RT_OK;
}

void stop_workers(rt_multi* m)
{
    for (Device& d : m->dev) {
        if (!d.worker) continue;
        {
            std::lock_guard<std::mutex> lock(d.worker->mu);
            d.worker->quit = true;
        }
        d.worker->cv.notify_all();
        if (d.worker->th.joinable()) d.worker->th.join();
        delete d.worker;
        d.worker = nullptr;
    }
}

// Runs fn(i) for every device on that device's thread, with the device current; a failure's
// message (thread-local in the library) is carried back to the caller's thread.
template <class F>
int on_every_device(rt_multi* m, F&& fn)
{
    for (size_t i = 0; i < m->dev.size(); ++i) {
        Device& d = m->dev[i];
        if (!d.worker) {
            rt_set_error("rt_multi: device %d has no thread", d.ordinal);
            return RT_E_STATE;
        }
        {
            std::lock_guard<std::mutex> lock(d.worker->mu);
            d.worker->job = [&fn, i] { return fn((int)i); };
            d.worker->has_job = true;
            d.worker->done = false;
        }
        d.worker->cv.notify_all();
    }
    for (Device& d : m->dev) {
        std::unique_lock<std::mutex> lock(d.worker->mu);
        d.worker->cv.wait(lock, [&] { return d.worker->done; });
    }
    for (size_t i = 0; i < m->dev.size(); ++i)
        if (m->dev[i].rc != RT_OK) {
            rt_set_error("device %d (part %zu): %s", m->dev[i].ordinal, i, m->dev[i].error.c_str());
            return m->dev[i].rc;
        }
    return RT_OK;
}

struct DeviceGuard {                // restores the caller's current device
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

int check_nccl(const rt_multi* m, ncclResult_t r, const char* what)
{
    if (r == 0) return RT_OK;
    rt_set_error("rt_multi: %s failed: %s", what, m->rccl.GetErrorString(r));
    return RT_E_HIP;
}

}  // namespace

extern "C" int rt_multi_create(rt_multi** out, int n_devices, const int* devices)
{
    RT_ARG(out, "null output");
    *out = nullptr;
    const int visible = rt_device_count();
    if (visible <= 0) {
        rt_set_error("rt_multi_create: no HIP device visible (this library has no CPU path)");
        return RT_E_NODEVICE;
    }
    RT_ARG(n_devices >= 1 && n_devices <= 64, "device count outside [1,64]");
    if (rt_g().stream) {
        rt_set_error("rt_multi_create: a user stream is set (rt_set_stream); the multi-device calls use "
                     "every device's default stream");
        return RT_E_STATE;
    }
    DeviceGuard guard;
    rt_multi* m = new rt_multi();
    m->dev.resize((size_t)n_devices);
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= visible) {
            rt_set_error("rt_multi_create: device %d out of range [0,%d)", d, visible);
            delete m;
            return RT_E_ARG;
        }
        m->dev[(size_t)i].ordinal = d;
        for (int k = 0; k < i; ++k) distinct = distinct && m->dev[(size_t)k].ordinal != d;
    }
    // every device: architecture check and static workspace (rt_init), a scene, a stream for the exchange
    const int first_device = rt_g().inited ? rt_g().device : m->dev[0].ordinal;
    int rc = RT_OK;
    for (Device& d : m->dev) {
        rc = rt_init(d.ordinal);
        if (rc) break;
        rc = rt_scene_create(&d.scene);
        if (rc) break;
        if (hipStreamCreateWithFlags(&d.comm_stream, hipStreamNonBlocking) != hipSuccess) {
            rt_set_error("rt_multi_create: stream creation failed on device %d", d.ordinal);
            rc = RT_E_HIP;
            break;
        }
    }
    if (rc == RT_OK) rc = rt_init(first_device);      // the single-device calls keep their device
    if (rc == RT_OK) rc = start_workers(m);
    // RT_AMD_MULTI_EXCHANGE=copy: peer copies instead of RCCL (also what a set that names a device
    // twice falls back to: RCCL refuses duplicate devices); =rccl: RCCL even for one device (its
    // part then goes through a send to itself), which rehearses the exchange on a one-GPU machine
    const char* mode = getenv("RT_AMD_MULTI_EXCHANGE");
    const bool force_copy = mode && strcmp(mode, "copy") == 0, force_rccl = mode && strcmp(mode, "rccl") == 0;
    m->use_rccl = !force_copy && distinct && (n_devices > 1 || force_rccl);
    if (rc == RT_OK && m->use_rccl) {
        rc = load_rccl(m->rccl);
        if (rc == RT_OK) {
            std::vector<int> list;
            for (const Device& d : m->dev) list.push_back(d.ordinal);
            m->comms.resize((size_t)n_devices);
            rc = check_nccl(m, m->rccl.CommInitAll(m->comms.data(), n_devices, list.data()), "ncclCommInitAll");
            if (rc) m->comms.clear();
        }
        // a machine without a usable RCCL still renders: the rows then travel by peer copies
        // (the reason stays readable through rt_multi_exchange); RT_AMD_MULTI_EXCHANGE=rccl insists
        if (rc != RT_OK && !force_rccl) {
            m->rccl_failure = rt_last_error();
            m->use_rccl = false;
            rc = RT_OK;
        }
    }
    if (rc) {
        std::string keep = rt_last_error();
        rt_multi_destroy(m);
        rt_set_error("%s", keep.c_str());
        return rc;
    }
    m->exchange = m->use_rccl ? "rccl" : (m->rccl_failure.empty() ? "copy" : "copy (" + m->rccl_failure + ")");
    *out = m;
    return RT_OK;
}

extern "C" const char* rt_multi_exchange(const rt_multi* m) { return m ? m->exchange.c_str() : ""; }

extern "C" int rt_multi_destroy(rt_multi* m)
{
    if (!m) return RT_OK;
    DeviceGuard guard;
    stop_workers(m);
    for (ncclComm_t c : m->comms)
        if (c) (void)m->rccl.CommDestroy(c);
    for (Device& d : m->dev) {
        (void)hipSetDevice(d.ordinal);
        if (d.comm_stream) (void)hipStreamDestroy(d.comm_stream);
        if (d.scene) (void)rt_scene_destroy(d.scene);
        d.rgb8.release();
    }
    if (!m->dev.empty()) {
        (void)hipSetDevice(m->dev[0].ordinal);
        m->gathered.release();
    }
    delete m;
    return RT_OK;
}

extern "C" int rt_multi_num_devices(const rt_multi* m) { return m ? (int)m->dev.size() : 0; }

extern "C" int rt_multi_for_each_scene(rt_multi* m, int (*fn)(rt_scene* scene, int part, void* user), void* user)
{
    RT_ARG(m && fn, "null pointer");
    DeviceGuard guard;
    // one after the other: the callback is the caller's code and need not be thread-safe
    for (size_t i = 0; i < m->dev.size(); ++i) {
        RT_HIP(hipSetDevice(m->dev[i].ordinal));
        const int rc = fn(m->dev[i].scene, (int)i, user);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

extern "C" int rt_multi_set_obj(rt_multi* m, const rt_obj* o)
{
    RT_ARG(m && o, "null pointer");
    DeviceGuard guard;
    return on_every_device(m, [&](int i) { return rt_scene_set_obj(m->dev[(size_t)i].scene, o); });
}

extern "C" int rt_multi_transform(rt_multi* m, const double obmat[4][4])
{
    RT_ARG(m && obmat, "null pointer");
    DeviceGuard guard;
    return on_every_device(m, [&](int i) { return rt_scene_transform(m->dev[(size_t)i].scene, obmat); });
}

extern "C" int rt_render_frame_multi(rt_multi* m, const rt_frame* f, int W, int H, double tan_half,
                                     uint8_t* screen)
{
    RT_ARG(m && f && screen, "null pointer");
    RT_ARG(W > 0 && H > 0, "bad image size");
    if (rt_g().stream) {
        rt_set_error("rt_render_frame_multi: a user stream is set (rt_set_stream)");
        return RT_E_STATE;
    }
    DeviceGuard guard;
    const int n = (int)m->dev.size();
    const size_t row_bytes = (size_t)W * 3;
    std::vector<rt_rows> rows((size_t)n);
    std::vector<size_t> bytes((size_t)n), offset((size_t)n, 0);
    size_t gathered_bytes = 0;
    for (int i = 0; i < n; ++i) {
        rows[(size_t)i] = rt_rows{0, H, RT_TILE_H, n, i};
        const int r = rt_rows_count(&rows[(size_t)i]);
        if (r < 0) return r;
        bytes[(size_t)i] = (size_t)r * row_bytes;
        if (i > 0 || m->use_rccl) {             // device 0's own rows are placed where they are, unless rehearsing
            offset[(size_t)i] = gathered_bytes;
            gathered_bytes += (bytes[(size_t)i] + 255) & ~(size_t)255;
        }
    }
    const bool profiling = rt_g().profiling;
    rt_g().profiling = false;       // the profile record is one per process; the devices' threads would race on it
    // 1. every device renders its part (driver.c:126-193 for its rows)
    int rc = on_every_device(m, [&](int i) {
        Device& d = m->dev[(size_t)i];
        if (bytes[(size_t)i] == 0) return (int)RT_OK;
        RT_TRY(d.rgb8.reserve(bytes[(size_t)i]));
        return rt_render_rows_part(d.scene, f, W, H, tan_half, &rows[(size_t)i], nullptr, nullptr, nullptr,
                                   d.rgb8.as<uint8_t>());
    });
    rt_g().profiling = profiling;
    if (rc) return rc;
    // 2. the rows travel to device 0
    RT_HIP(hipSetDevice(m->dev[0].ordinal));
    RT_TRY(m->gathered.reserve(gathered_bytes ? gathered_bytes : 256));
    uint8_t* gathered = m->gathered.as<uint8_t>();
    if (m->use_rccl) {
        // one group: every sender's ncclSend on its own communicator and stream, the root's
        // matching ncclRecv (rccl.h:700,722); the sizes differ from part to part
        RT_TRY(check_nccl(m, m->rccl.GroupStart(), "ncclGroupStart"));
        for (int i = 0; i < n && rc == RT_OK; ++i) {
            if (bytes[(size_t)i] == 0) continue;
            rc = check_nccl(m, m->rccl.Send(m->dev[(size_t)i].rgb8.p, bytes[(size_t)i], kNcclUint8, 0,
                                            m->comms[(size_t)i], m->dev[(size_t)i].comm_stream), "ncclSend");
            if (rc == RT_OK)
                rc = check_nccl(m, m->rccl.Recv(gathered + offset[(size_t)i], bytes[(size_t)i], kNcclUint8, i,
                                                m->comms[0], m->dev[0].comm_stream), "ncclRecv");
        }
        const int rc_end = check_nccl(m, m->rccl.GroupEnd(), "ncclGroupEnd");
        if (rc) return rc;
        if (rc_end) return rc_end;
        for (int i = 0; i < n; ++i) {
            RT_HIP(hipSetDevice(m->dev[(size_t)i].ordinal));
            RT_HIP(hipStreamSynchronize(m->dev[(size_t)i].comm_stream));
        }
        RT_HIP(hipSetDevice(m->dev[0].ordinal));
    } else {
        for (int i = 1; i < n; ++i) {
            if (bytes[(size_t)i] == 0) continue;
            RT_HIP(hipMemcpyPeerAsync(gathered + offset[(size_t)i], m->dev[0].ordinal, m->dev[(size_t)i].rgb8.p,
                                      m->dev[(size_t)i].ordinal, bytes[(size_t)i], m->dev[0].comm_stream));
        }
        RT_HIP(hipStreamSynchronize(m->dev[0].comm_stream));
    }
    // 3. placement on device 0 (driver.c:186-190; row 0 keeps the clear colour of driver.c:31-32)
    RT_HIP(hipMemsetAsync(screen, 0, (size_t)H * row_bytes, rt_g().stream));
    for (int i = 0; i < n; ++i) {
        if (bytes[(size_t)i] == 0) continue;
        const uint8_t* src = (i == 0 && !m->use_rccl) ? m->dev[0].rgb8.as<uint8_t>() : gathered + offset[(size_t)i];
        RT_TRY(rt_place_rows_part(src, W, H, &rows[(size_t)i], screen));
    }
    return RT_OK;
}
