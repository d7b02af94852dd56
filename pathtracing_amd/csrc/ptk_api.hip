// C ABI of the GPU library (include/ptk.h): scene flattening, path-pool management, the bounce
// loop that drives the wavefront kernels, film accumulation, denoise entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "device/kernels.h"
#include "host/bvh_builder.h"
#include "ptk.h"

using namespace ptk;

namespace {

thread_local std::string g_error;

// Scratch space of ptk_denoise_device_static, the entry point that allocates nothing: room for one float64 [H][W][3]
// image of up to STATIC_SCRATCH_PIXELS pixels (48 MiB).  Every device the library runs on holds its own copy of the
// array at its own address, from the moment the code object is loaded until the process ends.  In host code the
// array's name is only a handle for hipGetSymbolAddress: the address kernels and copies use is looked up once per
// device in ptk_init (DeviceContext::static_scratch) and nothing else may touch `scratch` by name.
constexpr int64_t STATIC_SCRATCH_PIXELS = int64_t(1) << 21;
constexpr size_t STATIC_SCRATCH_BYTES = static_cast<size_t>(STATIC_SCRATCH_PIXELS) * 24;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

// One context per device the process has initialised (ptk_init): the library's stream on that device and what the
// launchers need to know about it.  Scenes and films remember the context they were created in, so one process can
// drive several GPUs, one host thread per GPU; calls without a scene or film argument use the device the calling thread
// selected with its last ptk_init (or, for a thread that never called it, the process's most recent one).
struct DeviceContext {
    int device = -1;
    hipStream_t stream = nullptr;
    uint32_t compute_units = 256;
    double* static_scratch = nullptr; // this device's copy of `scratch` (hipGetSymbolAddress, with the device current)
};
constexpr int MAX_DEVICES = 64;
DeviceContext g_contexts[MAX_DEVICES];
std::mutex g_context_mutex;
std::atomic<int> g_default_device{-1};
thread_local int t_device = -1;

// Process-wide tuning knobs (ptk_set_option, ptk_set_profiling): read at every batch; results do not depend on them.
std::atomic<bool> g_profiling{false};
constexpr uint32_t TAIL_RECORDS_LIMIT = 1u << 24; // upper limit of the tail_records option (no batch is larger)
std::atomic<uint32_t> g_tail_records{TAIL_RECORDS};
// Bounces the host queues beyond the newest record count it has read back (option launch_ahead).
std::atomic<int> g_launch_ahead{7};
// Batches of a large job in flight at a time (option batch_lanes): 1, 2, or 0 = two for scenes with meshes, else one.
std::atomic<int> g_batch_lanes{0};
std::atomic<bool> g_specialised{true};
std::atomic<bool> g_quad_traverse{false};
std::atomic<bool> g_gpu_bvh_build{false};
// The closest-hit loop takes spheres and rectangles along the coordinate axes in runs of their own (option scan_runs,
// read when a scene is created; 0: every object in table order through the general loop).
std::atomic<bool> g_scan_runs{true};
// Most device memory a hierarchy build on the GPU may allocate, 0 for no limit (option gpu_bvh_memory_limit): a
// build that needs more falls back to the host, as it does when the device itself has no room.
std::atomic<int64_t> g_gpu_bvh_memory_limit{0};

int fail(int code, const std::string& message) {
    g_error = message;
    return code;
}

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(PTK_ERR_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(e_));          \
    } while (0)

// The calling thread's context, made current; nullptr (and the error message set) before any ptk_init.
DeviceContext* current_context() {
    const int device = t_device >= 0 ? t_device : g_default_device.load();
    if (device < 0) {
        fail(PTK_ERR_DEVICE, "ptk_init() has not been called (or no GPU is available)");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        fail(PTK_ERR_DEVICE, "hipSetDevice failed for the device selected by ptk_init");
        return nullptr;
    }
    return &g_contexts[device];
}
// Makes the device of a scene's or film's context current for the calling thread.
int use(const DeviceContext* ctx) {
    if (!ctx) return fail(PTK_ERR_INVALID, "object without a device context");
    HIP_TRY(hipSetDevice(ctx->device));
    return PTK_OK;
}
LaunchContext launch_context(const DeviceContext* ctx) {
    return LaunchContext{ctx->stream, ctx->compute_units, g_quad_traverse.load(), g_specialised.load()};
}

// Exceptions never cross the C boundary: the bodies of the entry points that allocate or parse run inside these.
#define PTK_GUARD_BEGIN try {
#define PTK_GUARD_END                                                                    \
    }                                                                                    \
    catch (const std::bad_alloc&) { return fail(PTK_ERR_MEMORY, "out of host memory"); } \
    catch (const std::exception& e) { return fail(PTK_ERR_INVALID, std::string("unexpected error: ") + e.what()); } \
    catch (...) { return fail(PTK_ERR_INVALID, "unexpected error"); }

// ---- host double[3] helpers in the reference's operation order ---------------------------------
struct H3 {
    double x, y, z;
};
H3 h3(const double* p) { return H3{p[0], p[1], p[2]}; }
H3 sub(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
double hdot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
H3 hcross(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
H3 hnormalize(H3 a) {
    double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return H3{a.x / l, a.y / l, a.z / l};
}
// The coordinate axis a vector is a unit vector of, up to sign (every other component a zero of either sign); -1 for
// any other vector.
int unit_axis(double x, double y, double z) {
    const double v[3] = {x, y, z};
    int axis = -1;
    for (int k = 0; k < 3; k++) {
        if (std::fabs(v[k]) == 1.0) {
            if (axis >= 0) return -1;
            axis = k;
        }
        else if (v[k] != 0.0) return -1;
    }
    return axis;
}
void put(double* dst, H3 v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}
H3 rotate_yzx(H3 p, double alpha, double beta, double gamma) { // src/utils.cpp:143-173
    double a = alpha * M_PI / 180, b = beta * M_PI / 180, g = gamma * M_PI / 180;
    double ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cg = std::cos(g), sg = std::sin(g);
    return H3{p.x * (ca * cb) + p.y * (sa * sg - ca * cg * sb) + p.z * (cg * sa + ca * sb * sg),
              p.x * (sb) + p.y * (cb * cg) + p.z * (-cb * sg),
              p.x * (-cb * sa) + p.y * (ca * sg + cg * sa * sb) + p.z * (ca * cg - sa * sb * sg)};
}

// Triangle::Triangle (src/objects.cpp:269-299) + set_vertex_UV / set_vertex_normals.
void make_triangle(const double* p, const double* uv, const double* nrm, uint8_t flags, int32_t face, TriHot& hot, TriCold& cold) {
    H3 p1 = h3(p), p2 = h3(p + 3), p3 = h3(p + 6);
    put(hot.p1, p1);
    put(hot.p2, p2);
    put(hot.p3, p3);
    H3 v1 = sub(p2, p1), v2 = sub(p3, p1);
    H3 normal = hnormalize(hcross(v1, v2));
    v1 = hnormalize(v1);
    v2 = hnormalize(hcross(normal, v1));
    put(cold.normal, normal);
    put(cold.v1, v1);
    put(cold.v2, v2);
    cold.x1 = hdot(p1, v1);
    cold.y1 = hdot(p1, v2);
    cold.x2 = hdot(p2, v1);
    cold.y2 = hdot(p2, v2);
    cold.x3 = hdot(p3, v1);
    cold.y3 = hdot(p3, v2);
    cold.det_T = (cold.y2 - cold.y3) * (cold.x1 - cold.x3) + (cold.x3 - cold.x2) * (cold.y1 - cold.y3);
    cold.area = 0.5 * std::abs(cold.x1 * (cold.y2 - cold.y3) + cold.x2 * (cold.y3 - cold.y1) + cold.x3 * (cold.y1 - cold.y2));
    for (int i = 0; i < 2; i++) {
        cold.uv1[i] = (flags & PTK_TRI_HAS_UV) ? uv[i] : 0.0;
        cold.uv2[i] = (flags & PTK_TRI_HAS_UV) ? uv[2 + i] : 0.0;
        cold.uv3[i] = (flags & PTK_TRI_HAS_UV) ? uv[4 + i] : 0.0;
    }
    if (flags & PTK_TRI_SMOOTH) {
        put(cold.n1, h3(nrm));
        put(cold.n2, h3(nrm + 3));
        put(cold.n3, h3(nrm + 6));
        cold.smooth = 1;
    }
    else {
        put(cold.n1, normal);
        put(cold.n2, normal);
        put(cold.n3, normal);
        cold.smooth = 0;
    }
    cold.face_index = face;
}

struct DeviceBuffers {
    std::vector<void*> allocations;
    std::vector<size_t> bytes; // of each allocation
    DeviceBuffers() = default;
    DeviceBuffers(const DeviceBuffers&) = delete;
    DeviceBuffers& operator=(const DeviceBuffers&) = delete;
    ~DeviceBuffers() { free_all(); }
    void free_all() {
        for (void* p : allocations) (void) hipFree(p);
        allocations.clear();
        bytes.clear();
    }
    template<typename T>
    hipError_t alloc(T** out, size_t count) {
        void* p = nullptr;
        const size_t size = std::max<size_t>(count, 1) * sizeof(T);
        hipError_t e = hipMalloc(&p, size);
        if (e == hipSuccess) {
            allocations.push_back(p);
            bytes.push_back(size);
        }
        *out = static_cast<T*>(p);
        return e;
    }
    template<typename T>
    hipError_t upload(const T** out, const std::vector<T>& host) {
        T* p = nullptr;
        hipError_t e = alloc(&p, host.size());
        if (e == hipSuccess && !host.empty()) e = hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
        *out = p;
        return e;
    }
    void release(void* p) {
        auto it = std::find(allocations.begin(), allocations.end(), p);
        if (it != allocations.end()) {
            (void) hipFree(p);
            bytes.erase(bytes.begin() + (it - allocations.begin()));
            allocations.erase(it);
        }
    }
};

// Paths of one batch: the kernels address a record by a 32-bit byte offset into its pool (160 bytes per record).
constexpr int64_t BATCH_MAX_PATHS = int64_t(1) << 24;
static_assert(BATCH_MAX_PATHS * 160 <= (int64_t(1) << 32), "record offsets are 32-bit");
constexpr int64_t DEFAULT_MAX_PATHS = 3 * BATCH_MAX_PATHS; // paths in flight of a scene created without a limit of its own: three full batches
                                                           // (buffers are sized by the batches a scene has really traced)
constexpr int MAX_LANES = 4;                              // batches in flight at a time, at most
constexpr int64_t MAX_PATHS_LIMIT = MAX_LANES * BATCH_MAX_PATHS; // a batch of the largest size in every lane
constexpr int INFO_RING = 8;

// What one batch in flight needs.  A scene has two lanes: a large job is cut into batches of half the paths-in-flight
// limit that are traced two at a time, each on a stream and a host thread of its own, so that while one batch runs
// its last bounces (few paths left, kernels that do not fill the device, single long walks) the other's take the
// idle compute units.  Lane 0 runs on the stream of the scene's device context and in the calling thread.
struct BatchLane {
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    DeviceBuffers buffers;      // path state, queues, results: sized by the largest batch so far
    DeviceBuffers fixed;        // summaries and statistics
    int64_t capacity = 0;
    PathPool pools[2] = {};     // path records, ping-pong
    SampleResults results{};
    uint32_t* list = nullptr;   // class-sorted record indices of the current bounce
    uint32_t* block_counts = nullptr;        // [N_CLASSES][count_stride]
    uint32_t count_stride = 0;
    BinInfo* d_info = nullptr;  // [2]: summaries of the previous and the current bounce's lists
    BinInfo* h_info = nullptr;  // pinned ring of INFO_RING summaries read back asynchronously
    hipEvent_t info_events[INFO_RING] = {};
    unsigned long long* h_rays = nullptr; // pinned [2][N_STATISTICS], then the stack-overflow counter (32 bits used)
    unsigned long long* d_statistics = nullptr; // STATISTICS_SLOTS per-block slots, then the totals
    DeferQueues queues{};       // scenes with meshes: rays and vertices waiting for k_traverse / k_bounce_resume
    PtkStats stats{};           // of the job in progress; added to the scene's when the job ends
    std::vector<hipEvent_t> events; // kernel-family timers while profiling is on
    hipEvent_t fence = nullptr; // orders this lane's stream and the context's at the ends of a job
    bool ready = false;
    BatchLane() = default;
    BatchLane(const BatchLane&) = delete;
    BatchLane& operator=(const BatchLane&) = delete;
    ~BatchLane() { // the device buffers go with their DeviceBuffers members
        if (h_info) (void) hipHostFree(h_info);
        if (h_rays) (void) hipHostFree(h_rays);
        for (hipEvent_t e : info_events)
            if (e) (void) hipEventDestroy(e);
        for (hipEvent_t e : events) (void) hipEventDestroy(e);
        if (fence) (void) hipEventDestroy(fence);
        if (owns_stream && stream) (void) hipStreamDestroy(stream);
    }
};

} // namespace

struct PtkFilm {
    DeviceContext* ctx = nullptr;
    int32_t width = 0, height = 0;
    bool owns_buffers = true;
    double *image = nullptr, *position = nullptr, *normal = nullptr;
    unsigned long long* counters = nullptr;
    // ptk_film_merge from a film of another device: that film's buffers are copied here, device to device, first
    double *stage_image = nullptr, *stage_position = nullptr, *stage_normal = nullptr;
    unsigned long long* stage_counters = nullptr;
};

// The host thread of a scene's second batch lane: started with the first two-lane job and kept until the scene goes,
// instead of one thread created and joined per job.
class LaneWorker {
public:
    LaneWorker() : thread_([this] { loop(); }) {}
    ~LaneWorker() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_all();
        thread_.join();
    }
    LaneWorker(const LaneWorker&) = delete;
    LaneWorker& operator=(const LaneWorker&) = delete;
    void start(std::function<void()> task) { // one task at a time: wait() before the next
        {
            std::lock_guard<std::mutex> lock(mutex_);
            task_ = std::move(task);
            busy_ = true;
        }
        wake_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lock(mutex_);
        done_.wait(lock, [this] { return !busy_; });
    }

private:
    void loop() {
        std::unique_lock<std::mutex> lock(mutex_);
        for (;;) {
            wake_.wait(lock, [this] { return stop_ || (busy_ && task_); });
            if (stop_) return;
            std::function<void()> task = std::move(task_);
            task_ = nullptr;
            lock.unlock();
            task(); // (the tasks catch their own exceptions)
            lock.lock();
            busy_ = false;
            done_.notify_all();
        }
    }
    std::mutex mutex_;
    std::condition_variable wake_, done_;
    std::function<void()> task_;
    bool busy_ = false, stop_ = false;
    std::thread thread_; // last: the thread starts when every other member exists
};

struct PtkScene {
    DeviceContext* ctx = nullptr;
    DevScene dev{};
    PtkSettings settings{};
    DeviceBuffers scene_buffers; // tables, triangles, nodes
    int64_t max_paths = DEFAULT_MAX_PATHS;
    BatchLane lanes[MAX_LANES];
    std::vector<int> classes; // queue classes present in the scene
    PtkFilm* host_film = nullptr; // scratch film for ptk_render_host
    std::unique_ptr<LaneWorker> workers[MAX_LANES - 1]; // the host threads of the lanes after the first, each from the first job that uses it on
    PtkStats stats{};
    double bvh_build_ms = 0;      // hierarchy builds of all meshes: device time (GPU build) or wall time (host build)
    double bvh_build_wall_ms = 0; // the same builds by the host's clock: allocations, copies, waits and the stack bound included
    int32_t bvh_meshes_on_gpu = 0, bvh_meshes_on_host = 0;
    int64_t bvh_nodes = 0;
    PtkScene() = default;
    PtkScene(const PtkScene&) = delete;
    PtkScene& operator=(const PtkScene&) = delete;
    ~PtkScene() { // the device buffers go with their DeviceBuffers members
        if (host_film) ptk_film_destroy(host_film);
    }
};

namespace {

int ensure_capacity(const PtkScene* s, BatchLane& lane, int64_t paths) {
    if (paths > BATCH_MAX_PATHS) return fail(PTK_ERR_MEMORY, "a batch of more paths than record offsets can address"); // (for_each_batch never asks)
    if (paths <= lane.capacity) return PTK_OK;
    lane.buffers.free_all(); // the previous, smaller arrays
    lane.capacity = 0;
    const size_t n = (static_cast<size_t>(paths) + 255) & ~static_cast<size_t>(255); // whole record tiles
    DeviceBuffers& b = lane.buffers;
    for (PathPool& pool : lane.pools) {
        HIP_TRY(b.alloc(&pool.base, n * N_PLANES * 16));
        HIP_TRY(b.alloc(&pool.cls, n));
    }
    HIP_TRY(b.alloc(&lane.results.color, n * 4));
    lane.results.plane_doubles = n * 2;
    HIP_TRY(b.alloc(&lane.results.surface, n * 6));
    HIP_TRY(b.alloc(&lane.list, n));
    if (s->dev.has_meshes) {
        HIP_TRY(b.alloc(&lane.queues.rays, 2 * n));
        HIP_TRY(b.alloc(&lane.queues.results, 2 * n));
        HIP_TRY(b.alloc(&lane.queues.vertices, n));
        HIP_TRY(b.alloc(&lane.queues.counters, 4));
        HIP_TRY(b.alloc(&lane.queues.pending, n * PENDING_PLANES * 2));
        lane.queues.capacity = n;
    }
    lane.count_stride = (chunk_capacity(static_cast<uint32_t>(n)) + 3u) & ~3u; // one entry per chunk; rows stay 16-byte aligned
    HIP_TRY(b.alloc(&lane.block_counts, static_cast<size_t>(lane.count_stride) * N_CLASSES));
    lane.capacity = paths;
    return PTK_OK;
}

// Bytes the kernels are specified to move per item (16-byte planes of the path records, see
// scene_device.h).
constexpr uint64_t RECORD_BYTES = N_PLANES * 16;
constexpr uint64_t BOUNCE_BYTES_PER_VERTEX = 4 + RECORD_BYTES;      // list entry, record in
constexpr uint64_t BOUNCE_BYTES_PER_CONTINUED = RECORD_BYTES + 1;   // record, class out
constexpr uint64_t BOUNCE_BYTES_PER_ENDED = 32 + 1;                 // radiance flushed to the results, class out
constexpr uint64_t BOUNCE_BYTES_PER_PATH = 3 * 16;                  // first-surface position and normal, once
constexpr uint64_t BIN_BYTES_PER_RECORD = 1;                        // class in
constexpr uint64_t BIN_BYTES_PER_VERTEX = 4;                        // list entry out
constexpr uint64_t GENERATE_BYTES_PER_PATH = RECORD_BYTES + 1;          // record, class out
constexpr uint64_t ACCUMULATE_BYTES_PER_PATH = 32 + 3 * 16;         // results in
constexpr uint64_t TRAVERSE_BYTES_PER_RAY = sizeof(RayRequest) + sizeof(RayResult);
constexpr uint64_t NODE_BYTES = BVH_NODE_BYTES; // without the padding
// two sets of statistics slots (k_bounce and k_bounce_resume; k_traverse), then their totals
constexpr size_t STATISTICS_SET = static_cast<size_t>(N_STATISTICS) * STATISTICS_SLOTS;
constexpr size_t STATISTICS_VALUES = 2 * (STATISTICS_SET + N_STATISTICS);
constexpr size_t STATISTICS_BYTES = STATISTICS_VALUES * sizeof(unsigned long long);

// The stream, summaries and statistics of a lane, once (lane 0: the context's stream; lane 1: one of the scene's own).
int prepare_lane(const PtkScene* s, BatchLane& lane, bool own_stream) {
    if (lane.ready) return PTK_OK;
    // Every resource is created only while its handle is still null, so a call that failed part-way is picked up
    // where it stopped by the next job instead of allocating everything again over the old handles.
    if (own_stream) {
        if (!lane.stream) HIP_TRY(hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking));
        lane.owns_stream = true;
    }
    else lane.stream = s->ctx->stream;
    if (!lane.d_info) HIP_TRY(lane.fixed.alloc(&lane.d_info, 2));
    if (!lane.d_statistics) HIP_TRY(lane.fixed.alloc(&lane.d_statistics, STATISTICS_VALUES));
    if (!lane.h_info) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&lane.h_info), INFO_RING * sizeof(BinInfo), hipHostMallocDefault));
    if (!lane.h_rays) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&lane.h_rays), (2 * N_STATISTICS + 1) * sizeof(unsigned long long), hipHostMallocDefault));
        lane.h_rays[2 * N_STATISTICS] = 0;
    }
    for (int i = 0; i < INFO_RING; i++)
        if (!lane.info_events[i]) HIP_TRY(hipEventCreateWithFlags(&lane.info_events[i], hipEventDisableTiming));
    if (!lane.fence) HIP_TRY(hipEventCreateWithFlags(&lane.fence, hipEventDisableTiming));
    lane.ready = true;
    return PTK_OK;
}

struct EventTimer {
    BatchLane* lane;
    bool on; // profiling as it was set when the batch began
    size_t next = 0;
    hipError_t error = hipSuccess; // the first failed event call; the timings of the batch are then dropped (collect)
    std::vector<std::pair<int, std::pair<size_t, size_t>>> spans; // family, (begin, end)
    void note(hipError_t e) {
        if (e != hipSuccess && error == hipSuccess) error = e;
    }
    hipEvent_t get() {
        if (next == lane->events.size()) {
            hipEvent_t e = nullptr;
            note(hipEventCreate(&e));
            lane->events.push_back(e);
        }
        return lane->events[next++];
    }
    void begin(int family) {
        if (!on) return;
        hipEvent_t e = get();
        if (e) note(hipEventRecord(e, lane->stream));
        spans.push_back({family, {next - 1, 0}});
    }
    void end() {
        if (!on) return;
        hipEvent_t e = get();
        if (e) note(hipEventRecord(e, lane->stream));
        spans.back().second.second = next - 1;
    }
    // Adds the spans' durations to the lane's statistics; returns the first error of any event call of the batch
    // (nothing is added then: a partial sum would be taken for a measurement).
    hipError_t collect() {
        if (!on) return hipSuccess;
        note(hipStreamSynchronize(lane->stream));
        while (!lane->events.empty() && !lane->events.back()) lane->events.pop_back(); // (events that could not be created)
        if (error != hipSuccess) {
            spans.clear();
            next = 0;
            return error;
        }
        std::vector<std::pair<int, float>> measured;
        for (auto& sp : spans) {
            float ms = 0;
            note(hipEventElapsedTime(&ms, lane->events[sp.second.first], lane->events[sp.second.second]));
            measured.push_back({sp.first, ms});
        }
        spans.clear();
        next = 0;
        if (error != hipSuccess) return error;
        for (auto& m : measured) {
            const float ms = m.second;
            const std::pair<int, int> sp{m.first, 0};
            switch (sp.first) {
                case 0: lane->stats.bounce_ms += ms; break;
                case 1: lane->stats.bin_ms += ms; break;
                case 3: lane->stats.traverse_ms += ms; break;
                default: lane->stats.other_ms += ms; break;
            }
        }
        return hipSuccess;
    }
};

// The record count at or below which a batch of n_paths paths is handed to the tail kernel.  Scenes without meshes:
// tail_records.  Scenes with meshes switch earlier, where the hierarchy walks of a bounce no longer fill the device,
// but not before three quarters of the batch's paths have ended: a quarter of the batch, kept between tail_records
// and four times tail_records; batches of at most tail_records paths switch at tail_records like any other scene.
uint64_t tail_threshold(uint32_t tail_records, bool has_meshes, uint32_t n_paths) {
    if (!has_meshes || n_paths <= tail_records) return tail_records;
    return std::min<uint64_t>(4ull * tail_records, std::max<uint64_t>(tail_records, n_paths / 4));
}

// Traces one batch (all samples of a contiguous pixel range) to completion; per-sample results
// are left in the lane's results.
//
// The host queues bounces without waiting for their counts: kernels read record counts from device
// memory, grids are sized by the newest count the host has seen (counts only shrink), and the
// per-bounce summaries come back through a ring of pinned slots that is polled, not waited on.
int trace_batch(PtkScene* s, BatchLane& lane, const BatchParams& b) {
    EventTimer timer{&lane, g_profiling.load()};
    PtkStats& st = lane.stats;
    LaunchContext lc = launch_context(s->ctx);
    lc.stream = lane.stream;
    const hipStream_t g_stream = lane.stream;
    const uint32_t tail_records = g_tail_records.load();
    const int launch_ahead = g_launch_ahead.load();
    HIP_TRY(hipMemsetAsync(lane.d_statistics, 0, STATISTICS_BYTES, g_stream));
    // the class counts start at zero once per batch; from then on every class sort leaves zeros where it read offsets
    HIP_TRY(hipMemsetAsync(lane.block_counts, 0, static_cast<size_t>(lane.count_stride) * N_CLASSES * sizeof(uint32_t), g_stream));
    if (s->dev.has_meshes) HIP_TRY(hipMemsetAsync(lane.queues.counters, 0, 4 * sizeof(uint32_t), g_stream));
    timer.begin(2);
    launch_generate(g_stream, s->dev, b, lane.pools[0], lane.results, &lane.d_info[1], lane.block_counts, lane.count_stride, lane.queues);
    timer.end();
    if (s->dev.has_meshes) { // the camera rays that enter the bounds of a mesh: walk, complete, count the classes
        timer.begin(3);
        launch_traverse(lc, b.n_paths, s->dev, lane.queues, lane.d_statistics + STATISTICS_SET);
        timer.end();
        timer.begin(2);
        launch_primary_resume(g_stream, b.n_paths, s->dev, lane.pools[0], lane.results, lane.queues);
        launch_count_classes(lc, b.n_paths, BIN_ROUNDS, s->dev, lane.pools[0], &lane.d_info[1], lane.block_counts, lane.count_stride);
        timer.end();
        st.traverse_launches++;
    }
    st.paths += b.n_paths;
    st.algorithmic_bytes += GENERATE_BYTES_PER_PATH * b.n_paths;

    uint32_t bound = b.n_paths;     // newest known upper bound of the record count
    uint32_t pool_rounds = BIN_ROUNDS; // chunk size (in rounds) the current pool was written with
    uint64_t records = b.n_paths;   // records the oldest unretired bounce walked
    uint64_t sum_records = 0, sum_vertices = 0;
    uint64_t class_vertices[N_CLASSES] = {};
    int launched = 0, retired = 0;
    bool finished = false;
    auto retire = [&](bool wait) -> int { // consumes completed summaries in order
        while (retired < launched) {
            const int slot = retired % INFO_RING;
            if (wait) HIP_TRY(hipEventSynchronize(lane.info_events[slot]));
            else if (hipEventQuery(lane.info_events[slot]) != hipSuccess) break;
            const uint32_t total = lane.h_info[slot].total;
            sum_records += records;
            sum_vertices += total;
            for (int c = 0; c < N_CLASSES; c++) class_vertices[c] += lane.h_info[slot].count[c];
            records = total;
            bound = total;
            if (total == 0) finished = true;
            retired++;
        }
        return PTK_OK;
    };
    // Bounce `depth` shades the vertices at that depth and finds the vertices at depth + 1.
    for (int depth = 0; depth <= s->settings.max_recursion_depth && !finished; depth++) {
        int rc = retire(false);
        if (rc != PTK_OK) return rc;
        while (launched - retired > launch_ahead) { // far enough ahead: wait for the oldest summary
            HIP_TRY(hipEventSynchronize(lane.info_events[retired % INFO_RING]));
            rc = retire(false);
            if (rc != PTK_OK) return rc;
        }
        if (finished) break;
        const PathPool& current = lane.pools[depth & 1];
        const PathPool& next = lane.pools[(depth + 1) & 1];
        BinInfo* previous = &lane.d_info[(depth + 1) & 1];
        BinInfo* info = &lane.d_info[depth & 1];
        timer.begin(1);
        launch_bin(g_stream, s->dev, current, bound, pool_rounds, previous, lane.block_counts, lane.count_stride, info, lane.list);
        timer.end();
        if (bound <= tail_threshold(tail_records, s->dev.has_meshes != 0, b.n_paths)) { // few paths left: one launch follows every one of them to its end
            timer.begin(0);
            launch_tail(lc, bound, s->dev, b, current, next, lane.results, lane.list, info, lane.d_statistics);
            timer.end();
            const int slot = launched % INFO_RING;
            HIP_TRY(hipMemcpyAsync(&lane.h_info[slot], info, sizeof(BinInfo), hipMemcpyDeviceToHost, g_stream));
            HIP_TRY(hipEventRecord(lane.info_events[slot], g_stream));
            launched++;
            st.bounce_launch_rounds++;
            st.bin_launches += 2;
            st.bounce_launches += 1;
            break;
        }
        timer.begin(0);
        pool_rounds = chunk_rounds(bound);
        if (s->dev.has_meshes) HIP_TRY(hipMemsetAsync(lane.queues.counters, 0, 4 * sizeof(uint32_t), g_stream));
        launch_bounce(lc, bound, pool_rounds, s->dev, b, current, next, lane.results, lane.list, info, lane.block_counts,
                      lane.count_stride, lane.d_statistics, lane.queues);
        timer.end();
        if (s->dev.has_meshes) { // finish what misses every mesh above, walk the hierarchies, finish the rest
            timer.begin(3);
            launch_traverse(lc, bound, s->dev, lane.queues, lane.d_statistics + STATISTICS_SET);
            timer.end();
            timer.begin(2); // (not the k_bounce family: that one's time per launch is k_bounce's own)
            launch_bounce_resume(lc, bound, s->dev, b, next, lane.results, lane.d_statistics, lane.queues);
            launch_count_classes(lc, bound, pool_rounds, s->dev, next, info, lane.block_counts, lane.count_stride);
            timer.end();
            st.traverse_launches++;
        }
        const int slot = launched % INFO_RING;
        HIP_TRY(hipMemcpyAsync(&lane.h_info[slot], info, sizeof(BinInfo), hipMemcpyDeviceToHost, g_stream));
        HIP_TRY(hipEventRecord(lane.info_events[slot], g_stream));
        launched++;
        st.bounce_launch_rounds++;
        st.bin_launches += 2;
        st.bounce_launches += 1;
    }
    unsigned long long* totals = lane.d_statistics + 2 * STATISTICS_SET;
    launch_sum_statistics(g_stream, lane.d_statistics, totals);
    if (s->dev.has_meshes) launch_sum_statistics(g_stream, lane.d_statistics + STATISTICS_SET, totals + N_STATISTICS);
    HIP_TRY(hipMemcpyAsync(lane.h_rays, totals, 2 * N_STATISTICS * sizeof(unsigned long long), hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipMemcpyAsync(lane.h_rays + 2 * N_STATISTICS, s->dev.stack_overflows, sizeof(uint32_t), hipMemcpyDeviceToHost, g_stream));
    int rc = retire(true);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(g_stream));
    const uint64_t segments = lane.h_rays[0], extensions = lane.h_rays[1] + b.n_paths, continued = lane.h_rays[3];
    sum_vertices += lane.h_rays[4]; // vertices k_tail shaded after its first round
    const uint64_t bounce_bytes = BOUNCE_BYTES_PER_PATH * b.n_paths + BOUNCE_BYTES_PER_VERTEX * sum_vertices +
                                  BOUNCE_BYTES_PER_CONTINUED * continued + BOUNCE_BYTES_PER_ENDED * (sum_vertices - continued);
    for (int c = 0; c < N_CLASSES; c++) st.class_vertices[c] += class_vertices[c];
#ifdef PTK_STAMPS
    std::fprintf(stderr, "wave-cycles/16: wait for the record %llu, shade %llu, object scan %llu, finish %llu; vertices %llu, rays %llu\n", lane.h_rays[4],
                 lane.h_rays[6], lane.h_rays[7], lane.h_rays[5], (unsigned long long) sum_vertices, (unsigned long long) (segments + extensions));
#endif
    if (s->dev.has_meshes) { // k_traverse: ray request in, result out, the 112 bytes of a node that hold boxes and children, 72-byte triangle vertices
        const unsigned long long* t = lane.h_rays + N_STATISTICS;
#ifdef PTK_TRAVERSE_PROBE
        std::fprintf(stderr, "traverse probe: rays %llu, nodes %llu, triangles %llu, passes of a wave %llu, lanes with a ray in them %llu, "
                             "rays of more than 128 / 512 / 2048 passes %llu / %llu / %llu\n",
                     t[0], t[1], t[2], t[3], t[7], t[4], t[5], t[6]);
#endif
        const uint64_t tb = TRAVERSE_BYTES_PER_RAY * t[0] + NODE_BYTES * t[1] + sizeof(TriHot) * t[2];
        st.traversed_rays += t[0];
        st.nodes_visited += t[1];
        st.triangles_tested += t[2];
        st.traverse_bytes += tb;
        st.algorithmic_bytes += tb;
    }
    st.stack_overflows = lane.h_rays[2 * N_STATISTICS]; // counted since the scene was created
    st.shadow_rays += segments;
    st.extension_rays += extensions;
    st.vertices += sum_vertices;
    const uint64_t bb = BIN_BYTES_PER_RECORD * sum_records + BIN_BYTES_PER_VERTEX * sum_vertices;
    st.bounce_bytes += bounce_bytes;
    st.bin_bytes += bb;
    st.algorithmic_bytes += bb + bounce_bytes;
    HIP_TRY(timer.collect());
    return PTK_OK;
}

int check_job(const PtkScene* s, const PtkRenderJob* job, const char* who) {
    if (!s || !job) return fail(PTK_ERR_INVALID, std::string(who) + ": null argument");
    const int64_t n = static_cast<int64_t>(s->settings.width) * s->settings.height;
    if (job->start_idx < 0 || job->number_of_pixels < 0 || job->start_idx + job->number_of_pixels > n)
        return fail(PTK_ERR_INVALID, std::string(who) + ": pixel range outside the image");
    if (job->number_of_samples < 0 || job->sample_offset < 0) return fail(PTK_ERR_INVALID, std::string(who) + ": negative sample count");
    if (job->number_of_samples > std::min(s->max_paths, BATCH_MAX_PATHS))
        return fail(PTK_ERR_MEMORY, std::string(who) + ": number_of_samples exceeds the paths-in-flight limit");
    return PTK_OK;
}

// Adds a lane's statistics of the job that has ended to the scene's.
void add_stats(PtkStats& total, const PtkStats& part) {
    total.paths += part.paths;
    total.extension_rays += part.extension_rays;
    total.shadow_rays += part.shadow_rays;
    total.vertices += part.vertices;
    total.bounce_launch_rounds += part.bounce_launch_rounds;
    total.algorithmic_bytes += part.algorithmic_bytes;
    total.bounce_ms += part.bounce_ms;
    total.bin_ms += part.bin_ms;
    total.other_ms += part.other_ms;
    total.traverse_ms += part.traverse_ms;
    total.bounce_launches += part.bounce_launches;
    total.bin_launches += part.bin_launches;
    total.traverse_launches += part.traverse_launches;
    total.bounce_bytes += part.bounce_bytes;
    total.bin_bytes += part.bin_bytes;
    total.traverse_bytes += part.traverse_bytes;
    for (int c = 0; c < 8; c++) total.class_vertices[c] += part.class_vertices[c];
    total.traversed_rays += part.traversed_rays;
    total.nodes_visited += part.nodes_visited;
    total.triangles_tested += part.triangles_tested;
    total.stack_overflows = std::max(total.stack_overflows, part.stack_overflows); // one counter per scene, read by both lanes
}

// Splits a job into batches of whole pixels, traces them and calls `after_batch(lane, batch, first_pixel_of_batch_in_job)`
// for each from the host thread of the batch's lane (the lane's stream and results are the ones to use).
// band_pixels > 0: the job's pixels are runs of that many pixels, band_stride pixels apart (ptk_render_bands).
// Jobs of TWO_LANE_PATHS paths or more go through both lanes, two batches at a time, in scenes with meshes (option
// batch_lanes; measured on one MI355X, 1080p, 8 samples per pixel per job: +4 % on the 1M-triangle mesh and on the room, whose
// hierarchy walks end in a few long rays that leave most of the device idle; nothing on the Cornell scene, -2 % on the
// cloud) and, in every scene, for jobs that take more than one batch.  Every batch covers pixels of its own, so the order in which batches finish does not show in the results.
constexpr int64_t TWO_LANE_PATHS = int64_t(1) << 21;
template<typename F>
int for_each_batch(PtkScene* s, const PtkRenderJob* job, F&& after_batch, uint32_t band_pixels = 0, uint32_t band_stride = 0) {
    if (job->number_of_pixels == 0 || job->number_of_samples == 0) return PTK_OK;
    const int64_t job_paths = job->number_of_pixels * job->number_of_samples;
    const int wanted = g_batch_lanes.load();
    // two lanes: where asked for; in scenes with meshes; and for a job that takes several batches anyway (the lanes then
    // trace batches of the size one lane would, and the short kernels at the end of one batch run beside the large ones
    // of the other)
    const int64_t one_lane_batch = std::min(s->max_paths, BATCH_MAX_PATHS);
    // automatic: two lanes in scenes with meshes and wherever a job takes more than one batch anyway, three in scenes
    // with meshes for jobs of more than two batches (measured, 64 spp per job at 1080p: 1M-triangle mesh 407 / 417 / 404
    // Mpaths/s with 2 / 3 / 4 lanes, room 524 / 536 with 2 / 3, the Cornell scene 1685 with 2 or 3)
    int n_lanes = wanted > 0 ? wanted : 1;
    if (wanted == 0) {
        if (s->dev.has_meshes) n_lanes = job_paths > 2 * one_lane_batch ? 3 : 2;
        else if (job_paths > one_lane_batch) n_lanes = 2;
    }
    if (job_paths < TWO_LANE_PATHS) n_lanes = 1;
    while (n_lanes > 1 && (job->number_of_pixels < n_lanes || n_lanes * job->number_of_samples > s->max_paths)) n_lanes--;
    const int64_t per_batch = std::max<int64_t>(1, std::min(s->max_paths / n_lanes, BATCH_MAX_PATHS) / job->number_of_samples);
    // batches of equal size, as many for one lane as for the other
    int64_t n_batches = (job->number_of_pixels + per_batch - 1) / per_batch;
    n_batches = (n_batches + n_lanes - 1) / n_lanes * n_lanes;
    const int64_t batch_pixels = (job->number_of_pixels + n_batches - 1) / n_batches;
    std::vector<BatchParams> batches;
    std::vector<int64_t> first_pixel;
    for (int64_t p = 0; p < job->number_of_pixels; p += batch_pixels) {
        BatchParams b{};
        b.start_idx = band_pixels ? job->start_idx : job->start_idx + p;
        b.band_pixels = band_pixels;
        b.band_stride = band_stride;
        b.band_first = band_pixels ? static_cast<uint32_t>(p) : 0u;
        b.n_pixels = static_cast<uint32_t>(std::min<int64_t>(batch_pixels, job->number_of_pixels - p));
        b.n_samples = static_cast<uint32_t>(job->number_of_samples);
        b.sample_offset = static_cast<uint32_t>(job->sample_offset);
        b.n_paths = b.n_pixels * b.n_samples;
        b.seed = job->seed;
        batches.push_back(b);
        first_pixel.push_back(p);
    }
    int rc = PTK_OK;
    for (int l = 0; l < n_lanes; l++) {
        if ((rc = prepare_lane(s, s->lanes[l], l > 0)) != PTK_OK) return rc;
        if ((rc = ensure_capacity(s, s->lanes[l], batch_pixels * job->number_of_samples)) != PTK_OK) return rc;
    }
    auto run_lane = [&](int l) -> int {
        BatchLane& lane = s->lanes[l];
        for (size_t k = static_cast<size_t>(l); k < batches.size(); k += static_cast<size_t>(n_lanes)) {
            int r = trace_batch(s, lane, batches[k]);
            if (r == PTK_OK) r = after_batch(lane, batches[k], first_pixel[k]);
            if (r != PTK_OK) return r;
        }
        return PTK_OK;
    };
    if (n_lanes == 1) rc = run_lane(0);
    else {
        // the other lanes' streams begin after what the context's stream holds so far (a film cleared or uploaded), and the
        // context's stream goes on after the other lanes' work
        hipError_t e = hipSuccess;
        for (int l = 1; l < n_lanes && e == hipSuccess; l++) {
            e = hipEventRecord(s->lanes[l].fence, s->ctx->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(s->lanes[l].stream, s->lanes[l].fence, 0);
        }
        if (e != hipSuccess) rc = fail(PTK_ERR_DEVICE, std::string("job in several lanes: ") + hipGetErrorString(e));
        else {
            int other_rc[MAX_LANES] = {};
            std::string other_message[MAX_LANES];
            for (int l = 1; l < n_lanes; l++) {
                if (!s->workers[l - 1]) s->workers[l - 1].reset(new LaneWorker());
                s->workers[l - 1]->start([&, l] {
                    try {
                        other_rc[l] = use(s->ctx);
                        if (other_rc[l] == PTK_OK) other_rc[l] = run_lane(l);
                    }
                    catch (const std::exception& ex) {
                        other_rc[l] = fail(PTK_ERR_INVALID, std::string("unexpected error: ") + ex.what());
                    }
                    catch (...) {
                        other_rc[l] = fail(PTK_ERR_INVALID, "unexpected error");
                    }
                    if (other_rc[l] != PTK_OK) other_message[l] = g_error; // the message is the thread's own
                });
            }
            try {
                rc = run_lane(0);
            }
            catch (...) { // the workers are waited for whatever happens here: they work on this frame's variables
                for (int l = 1; l < n_lanes; l++) s->workers[l - 1]->wait();
                throw;
            }
            for (int l = 1; l < n_lanes; l++) s->workers[l - 1]->wait();
            for (int l = 1; l < n_lanes; l++)
                if (rc == PTK_OK && other_rc[l] != PTK_OK) rc = fail(other_rc[l], other_message[l]);
            for (int l = 1; l < n_lanes && rc == PTK_OK; l++) { // the context's stream goes on after the other lanes' work
                e = hipEventRecord(s->lanes[l].fence, s->lanes[l].stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(s->ctx->stream, s->lanes[l].fence, 0);
                if (e != hipSuccess) rc = fail(PTK_ERR_DEVICE, std::string("job in several lanes: ") + hipGetErrorString(e));
            }
            // nothing of a failed job is left running on the caller's buffers
            if (rc != PTK_OK)
                for (int l = 1; l < n_lanes; l++) (void) hipStreamSynchronize(s->lanes[l].stream);
        }
    }
    // on every path out of the job: the lanes' statistics are added to the scene's and cleared
    for (int l = 0; l < n_lanes; l++) {
        add_stats(s->stats, s->lanes[l].stats);
        s->lanes[l].stats = PtkStats{};
    }
    return rc;
}

} // namespace

// =================================================================================================
extern "C" {

const char* ptk_last_error(void) {
    return g_error.c_str();
}

int ptk_init(int device) {
    PTK_GUARD_BEGIN
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(PTK_ERR_DEVICE, "no HIP device available: this library has no CPU path");
    if (device < 0 || device >= count || device >= MAX_DEVICES) return fail(PTK_ERR_INVALID, "ptk_init: device index out of range");
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_context_mutex);
    DeviceContext& ctx = g_contexts[device];
    if (ctx.device < 0) { // first use of this device: its stream lives until the process ends
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(PTK_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", but the kernels are built for gfx950 only");
        if (prop.multiProcessorCount > 0) ctx.compute_units = static_cast<uint32_t>(prop.multiProcessorCount);
        if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking)); // (kept if a later step of an earlier call failed)
        void* address = nullptr; // of this device's copy of the static scratch array (the device is current)
        HIP_TRY(hipGetSymbolAddress(&address, HIP_SYMBOL(scratch)));
        if (!address) return fail(PTK_ERR_DEVICE, "ptk_init: the static scratch array has no address on this device");
        ctx.static_scratch = static_cast<double*>(address);
        ctx.device = device;
    }
    t_device = device;
    g_default_device = device;
    const char* prof = std::getenv("PTK_PROFILE");
    if (prof && prof[0] == '1') g_profiling = true;
    return PTK_OK;
    PTK_GUARD_END
}

int ptk_set_option(const char* name, int64_t value) {
    if (name && std::strcmp(name, "tail_records") == 0) {
        if (value < 0 || value > static_cast<int64_t>(TAIL_RECORDS_LIMIT))
            return fail(PTK_ERR_INVALID, "ptk_set_option: tail_records must be between 0 and 16777216");
        g_tail_records = static_cast<uint32_t>(value);
        return PTK_OK;
    }
    if (name && std::strcmp(name, "launch_ahead") == 0) {
        if (value < 0 || value >= INFO_RING) return fail(PTK_ERR_INVALID, "ptk_set_option: launch_ahead must be between 0 and 7");
        g_launch_ahead = static_cast<int>(value);
        return PTK_OK;
    }
    if (name && std::strcmp(name, "batch_lanes") == 0) {
        if (value < 0 || value > MAX_LANES) return fail(PTK_ERR_INVALID, "ptk_set_option: batch_lanes must be 0 (automatic) or 1 to 4");
        g_batch_lanes = static_cast<int>(value);
        return PTK_OK;
    }
    if (name && std::strcmp(name, "gpu_bvh_build") == 0) {
        if (value != 0 && value != 1) return fail(PTK_ERR_INVALID, "ptk_set_option: gpu_bvh_build must be 0 or 1");
        g_gpu_bvh_build = value != 0;
        return PTK_OK;
    }
    if (name && std::strcmp(name, "gpu_bvh_memory_limit") == 0) { // bytes; read when a scene is created
        if (value < 0) return fail(PTK_ERR_INVALID, "ptk_set_option: gpu_bvh_memory_limit must not be negative");
        g_gpu_bvh_memory_limit = value;
        return PTK_OK;
    }
    if (name && std::strcmp(name, "scan_runs") == 0) { // read when a scene is created
        if (value != 0 && value != 1) return fail(PTK_ERR_INVALID, "ptk_set_option: scan_runs must be 0 or 1");
        g_scan_runs = value != 0;
        return PTK_OK;
    }
    if (name && std::strcmp(name, "traverse_lanes") == 0) {
        if (value != 1 && value != 4) return fail(PTK_ERR_INVALID, "ptk_set_option: traverse_lanes must be 1 or 4");
        if (value == 4 && BVH_WIDTH != 4) return fail(PTK_ERR_INVALID, "ptk_set_option: four lanes per ray need nodes of four children");
        g_quad_traverse = value == 4;
        return PTK_OK;
    }
    if (name && std::strcmp(name, "specialised_kernels") == 0) {
        if (value != 0 && value != 1) return fail(PTK_ERR_INVALID, "ptk_set_option: specialised_kernels must be 0 or 1");
        g_specialised = value != 0;
        return PTK_OK;
    }
    return fail(PTK_ERR_INVALID, std::string("ptk_set_option: unknown option ") + (name ? name : "(null)"));
}

int ptk_set_profiling(int enabled) {
    g_profiling = enabled != 0;
    return PTK_OK;
}

int ptk_sync(void) {
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

void* ptk_stream(void) {
    DeviceContext* ctx = current_context();
    return ctx ? ctx->stream : nullptr;
}

int ptk_scene_create(const PtkSceneDesc* d, int64_t max_paths_in_flight, PtkScene** out) {
    if (!d || !out) return fail(PTK_ERR_INVALID, "ptk_scene_create: null argument");
    *out = nullptr;
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    PTK_GUARD_BEGIN
    if (d->settings.width < 1 || d->settings.height < 1) return fail(PTK_ERR_INVALID, "ptk_scene_create: empty image");
    if (static_cast<int64_t>(d->settings.width) * d->settings.height > (int64_t(1) << 31))
        return fail(PTK_ERR_INVALID, "ptk_scene_create: image larger than 2^31 pixels");
    if (d->n_objects < 0 || d->n_objects > 65535) return fail(PTK_ERR_INVALID, "ptk_scene_create: at most 65535 top-level objects");
    if (d->n_media < 1 || d->n_media > 32766) return fail(PTK_ERR_INVALID, "ptk_scene_create: between 1 and 32766 media required");
    if (d->settings.max_recursion_depth < 0 || d->settings.max_recursion_depth > 8000)
        return fail(PTK_ERR_INVALID, "ptk_scene_create: max_recursion_depth must be in [0, 8000]");
    if (d->background_medium < 0 || d->background_medium >= d->n_media)
        return fail(PTK_ERR_INVALID, "ptk_scene_create: background_medium out of range");
    if (d->n_triangles < 0 || d->n_triangles > (int64_t(1) << 28)) return fail(PTK_ERR_INVALID, "ptk_scene_create: between 0 and 2^28 triangles");
    if (d->n_materials < 0 || d->n_materials > 65535 || d->n_maps < 0 || d->n_maps > (1 << 24) || d->n_map_values < 0)
        return fail(PTK_ERR_INVALID, "ptk_scene_create: negative or excessive table size");
    if ((d->n_objects > 0 && !d->objects) || (d->n_materials > 0 && !d->materials) || !d->media || (d->n_maps > 0 && !d->maps) ||
        (d->n_map_values > 0 && !d->map_data))
        return fail(PTK_ERR_INVALID, "ptk_scene_create: a table with a positive size has a null pointer");
    if (d->n_triangles > 0 && (!d->tri_vertices || !d->tri_uv || !d->tri_normals || !d->tri_flags))
        return fail(PTK_ERR_INVALID, "ptk_scene_create: the triangle arrays (vertices, uv, normals, flags) must all be present");
    if (d->settings.bvh_leaf_size < 1 || d->settings.bvh_n_axis_splits < 1)
        return fail(PTK_ERR_INVALID, "ptk_scene_create: bvh_leaf_size and bvh_n_axis_splits must be at least 1");

    std::unique_ptr<PtkScene> s(new PtkScene());
    s->ctx = ctx;
    s->settings = d->settings;
    // records are addressed with 32-bit byte offsets (160 bytes each): at most 2^24 in flight; larger jobs are
    // cut into batches anyway
    s->max_paths = max_paths_in_flight > 0 ? std::min(max_paths_in_flight, MAX_PATHS_LIMIT) : DEFAULT_MAX_PATHS;
    DevScene& dev = s->dev;
    DeviceBuffers& buf = s->scene_buffers;

    // maps
    std::vector<DevMap> maps(static_cast<size_t>(d->n_maps));
    for (int i = 0; i < d->n_maps; i++) {
        const PtkMap& m = d->maps[i];
        if (m.width < 1 || m.height < 1 || (m.dim != 1 && m.dim != 3) || m.offset < 0 ||
            m.offset + static_cast<int64_t>(m.width) * m.height * m.dim > d->n_map_values)
            return fail(PTK_ERR_INVALID, "ptk_scene_create: malformed map table");
        maps[static_cast<size_t>(i)] = DevMap{m.width, m.height, m.dim, 0, m.offset};
    }
    std::vector<double> map_data(d->map_data, d->map_data + d->n_map_values);

    // media (Medium::Medium src/medium.cpp:4-11; BeersLawMedium drops scattering :37-39)
    std::vector<DevMedium> media(static_cast<size_t>(d->n_media));
    bool any_scattering = false;
    for (int i = 0; i < d->n_media; i++) {
        const PtkMedium& pm = d->media[i];
        DevMedium m{};
        m.type = pm.type;
        m.ior = pm.refractive_index;
        for (int k = 0; k < 3; k++) {
            m.absorption[k] = pm.absorption[k];
            m.scattering[k] = pm.type == PTK_MEDIUM_BEERS_LAW ? 0.0 : pm.scattering[k];
            m.emission[k] = pm.emission[k];
            m.extinction[k] = m.absorption[k] + m.scattering[k];
        }
        m.is_clear = m.extinction[0] == 0 && m.extinction[1] == 0 && m.extinction[2] == 0;
        any_scattering |= pm.type == PTK_MEDIUM_HOMOGENEOUS_SCATTERING;
        media[static_cast<size_t>(i)] = m;
    }

    // materials (Material::Material src/materials.cpp:3-43; ensure_metallic :322-327)
    auto map_ok = [&](int id, int dim) { return id == -1 || (id >= 0 && id < d->n_maps && d->maps[id].dim == dim); };
    auto map_big = [&](int id) { return id >= 0 && (d->maps[id].width > 1 || d->maps[id].height > 1); };
    std::vector<DevMaterial> materials(static_cast<size_t>(d->n_materials));
    for (int i = 0; i < d->n_materials; i++) {
        const PtkMaterial& pm = d->materials[i];
        if (pm.type < 0 || pm.type >= PTK_MAT_COUNT) return fail(PTK_ERR_INVALID, "ptk_scene_create: unknown material type");
        if (!map_ok(pm.albedo_map, 3) || !map_ok(pm.emission_color_map, 3) || !map_ok(pm.light_intensity_map, 1) ||
            !map_ok(pm.roughness_map, 1) || !map_ok(pm.transparency_map, 1))
            return fail(PTK_ERR_INVALID, "ptk_scene_create: material refers to a missing map or one of the wrong channel count");
        if (pm.internal_medium < -1 || pm.internal_medium >= d->n_media || pm.external_medium < -1 || pm.external_medium >= d->n_media)
            return fail(PTK_ERR_INVALID, "ptk_scene_create: material refers to a missing medium");
        DevMaterial m{};
        m.type = pm.type;
        m.is_light = pm.is_light_source != 0;
        m.is_dielectric = pm.is_dielectric != 0;
        if (pm.type == PTK_MAT_METALLIC_MICROFACET || pm.type == PTK_MAT_REFLECTIVE_MICROFACET) m.is_dielectric = 0;
        m.n = pm.surface_refractive_index;
        m.k = m.is_dielectric ? 0.0 : pm.extinction_coefficient;
        m.albedo = pm.albedo_map;
        m.emission = pm.emission_color_map;
        m.intensity = pm.light_intensity_map;
        m.roughness = pm.roughness_map;
        m.transparency = pm.transparency_map;
        m.internal_medium = pm.internal_medium;
        m.external_medium = pm.external_medium;
        m.needs_uv = map_big(m.albedo) || map_big(m.emission) || map_big(m.intensity) || map_big(m.roughness) || map_big(m.transparency);
        // first texels (ValueMap::get with no texture coordinates reads element 0; a missing map reads as 0)
        auto texel = [&](int id, int channel) { return id < 0 ? 0.0 : d->map_data[d->maps[id].offset + channel]; };
        const double intensity = texel(m.intensity, 0);
        for (int c = 0; c < 3; c++) {
            m.c_albedo[c] = texel(m.albedo, c);
            m.c_emittance[c] = texel(m.emission, c) * intensity; // Material::get_light_emittance src/materials.cpp:72-74
        }
        m.c_roughness = texel(m.roughness, 0);
        m.c_transparency = texel(m.transparency, 0);
        materials[static_cast<size_t>(i)] = m;
    }

    // objects and triangles
    std::vector<DevObject> objects(static_cast<size_t>(d->n_objects));
    std::vector<TriHot> tri_hot;
    std::vector<TriCold> tri_cold;
    std::vector<BvhNode4> nodes;
    std::vector<double> light_area;
    std::vector<int32_t> light_tri;
    std::vector<int32_t> lights;
    uint32_t class_mask = any_scattering ? (1u << CLASS_SCATTER) : 0u;
    const int build_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    for (int i = 0; i < d->n_objects; i++) {
        const PtkObject& po = d->objects[i];
        if (po.material < 0 || po.material >= d->n_materials) return fail(PTK_ERR_INVALID, "ptk_scene_create: object refers to a missing material");
        DevObject o{};
        o.type = po.type;
        o.material = po.material;
        o.is_light = materials[static_cast<size_t>(po.material)].is_light;
        o.bvh_root = -1;
        o.index = i;
        o.mat = materials[static_cast<size_t>(po.material)];
        class_mask |= 1u << materials[static_cast<size_t>(po.material)].type;
        if (o.is_light) lights.push_back(i);
        switch (po.type) {
            case PTK_OBJ_SPHERE: // Sphere::Sphere src/objects.cpp:90-94
                o.px = po.position[0];
                o.py = po.position[1];
                o.pz = po.position[2];
                o.radius = po.radius;
                o.area = 4 * M_PI * o.radius * o.radius;
                break;
            case PTK_OBJ_PLANE:
            case PTK_OBJ_RECTANGLE: { // Plane::Plane :170-176, Rectangle::Rectangle :223-229
                o.px = po.position[0];
                o.py = po.position[1];
                o.pz = po.position[2];
                H3 v1 = hnormalize(h3(po.v1)), v2 = hnormalize(h3(po.v2));
                H3 n = hnormalize(hcross(v1, v2));
                o.v1x = v1.x;
                o.v1y = v1.y;
                o.v1z = v1.z;
                o.v2x = v2.x;
                o.v2y = v2.y;
                o.v2z = v2.z;
                o.nx = n.x;
                o.ny = n.y;
                o.nz = n.z;
                if (po.type == PTK_OBJ_RECTANGLE) {
                    o.L1 = po.L1;
                    o.L2 = po.L2;
                    o.area = o.L1 * o.L2;
                }
                break;
            }
            case PTK_OBJ_TRIANGLE: {
                double p[9];
                std::memcpy(p, po.position, 24);
                std::memcpy(p + 3, po.v1, 24);
                std::memcpy(p + 6, po.v2, 24);
                TriHot hot{};
                TriCold cold{};
                make_triangle(p, nullptr, nullptr, 0, 0, hot, cold);
                o.first_tri = static_cast<int64_t>(tri_hot.size());
                o.n_tri = 1;
                o.area = cold.area;
                tri_hot.push_back(hot);
                tri_cold.push_back(cold);
                break;
            }
            case PTK_OBJ_MESH: {
                if (po.first_triangle < 0 || po.n_triangles < 0 || po.first_triangle + po.n_triangles > d->n_triangles)
                    return fail(PTK_ERR_INVALID, "ptk_scene_create: mesh triangle range out of bounds");
                const int64_t n = po.n_triangles;
                o.first_tri = static_cast<int64_t>(tri_hot.size());
                o.n_tri = n;
                if (n == 0) break;
                const double* verts = d->tri_vertices + 9 * po.first_triangle;
                // bvh_leaf_size tunes the reference's tree for its cost balance; here an exact fp64 triangle test costs
                // about as much as a 4-wide node, so leaves hold at most two triangles (measured on the 1M-triangle
                // workload, k_traverse per frame: 24.7 ms with one triangle per leaf, 23.3 with two, 23.1 with three, 23.9 with four)
                const int stack_limit = std::min(LDS_STACK_DEPTH + SPILL_STACK_DEPTH, QUAD_STACK_DEPTH - 1);
                const int max_leaf = std::min(d->settings.bvh_leaf_size, 2);
                BuiltBvh bvh;
                double build_ms = 0;
                hipError_t build_error = hipSuccess;
                const auto wall0 = std::chrono::steady_clock::now();
                if (g_gpu_bvh_build.load() && build_bvh4_gpu(verts, n, max_leaf, stack_limit, ctx->stream, bvh, &build_ms, &build_error,
                                                             static_cast<size_t>(g_gpu_bvh_memory_limit.load())))
                    s->bvh_meshes_on_gpu++;
                else {
                    // A build the device has no memory for (some 370 bytes per triangle) is no reason to refuse a scene
                    // that loads with the option off: the hierarchy is built on the host.  (The error is not sticky;
                    // reading it clears it.)  Any other HIP error fails the call.
                    if (build_error == hipErrorOutOfMemory) {
                        (void) hipGetLastError();
                        build_error = hipSuccess;
                    }
                    HIP_TRY(build_error);
                    const auto t0 = std::chrono::steady_clock::now();
                    bvh = build_bvh4(verts, n, max_leaf, d->settings.bvh_n_axis_splits, build_threads, stack_limit);
                    build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    s->bvh_meshes_on_host++;
                }
                s->bvh_build_ms += build_ms;
                s->bvh_build_wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
                s->bvh_nodes += static_cast<int64_t>(bvh.nodes.size());
                // backstop: the builder bounds the stack its tree needs (build_bvh4), so this cannot fail
                if (bvh.max_stack > std::min(LDS_STACK_DEPTH + SPILL_STACK_DEPTH, QUAD_STACK_DEPTH - 1))
                    return fail(PTK_ERR_MEMORY, "ptk_scene_create: mesh hierarchy needs a deeper traversal stack than the kernels have");
                std::vector<int32_t> leaf_position(static_cast<size_t>(n)); // face -> position in leaf order
                for (int64_t k = 0; k < n; k++) leaf_position[bvh.order[static_cast<size_t>(k)]] = static_cast<int32_t>(k);
                const size_t base = tri_hot.size();
                tri_hot.resize(base + static_cast<size_t>(n));
                tri_cold.resize(base + static_cast<size_t>(n));
                double area = 0; // ObjectUnion::ObjectUnion sums in face order (src/objectunion.cpp:9-14)
                std::vector<double> face_area(static_cast<size_t>(n));
                for (int64_t f = 0; f < n; f++) {
                    const int64_t g = po.first_triangle + f;
                    const size_t slot = base + static_cast<size_t>(leaf_position[static_cast<size_t>(f)]);
                    make_triangle(d->tri_vertices + 9 * g, d->tri_uv + 6 * g, d->tri_normals + 9 * g, d->tri_flags[g],
                                  static_cast<int32_t>(f), tri_hot[slot], tri_cold[slot]);
                    face_area[static_cast<size_t>(f)] = tri_cold[slot].area;
                    area += tri_cold[slot].area;
                }
                o.area = area;
                o.bvh_root = static_cast<int32_t>(nodes.size());
                const int32_t node_base = static_cast<int32_t>(nodes.size());
                for (BvhNode4 node : bvh.nodes) {
                    for (int c = 0; c < BVH_WIDTH; c++)
                        if (node.child[c] >= 0) node.child[c] += node_base;
                    nodes.push_back(node);
                }
                if (o.is_light) {
                    // The reference's area table is filled in face order before the BVH root sorts the
                    // triangle array in place, and is then used to index the sorted array
                    // (src/objectunion.cpp:19-43, :100-142).  Reproduce both orders.
                    o.n_light_tris = static_cast<int32_t>(n);
                    o.light_table = static_cast<int64_t>(light_area.size());
                    double c = 0;
                    for (int64_t f = 0; f < n; f++) {
                        c = f == 0 ? face_area[0] : face_area[static_cast<size_t>(f)] + c;
                        light_area.push_back(c);
                    }
                    o.mesh_light_area = c;
                    std::vector<int> sorted(static_cast<size_t>(n));
                    for (int64_t f = 0; f < n; f++) sorted[static_cast<size_t>(f)] = static_cast<int>(f);
                    if (n > d->settings.bvh_leaf_size) { // root Node::Node: get_split_axis + sort_by_axis (src/bvh.cpp:129-177)
                        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                        for (int64_t f = 0; f < n; f++)
                            for (int v = 0; v < 3; v++)
                                for (int a = 0; a < 3; a++) {
                                    lo[a] = std::min(lo[a], verts[9 * f + 3 * v + a]);
                                    hi[a] = std::max(hi[a], verts[9 * f + 3 * v + a]);
                                }
                        int axis = 0;
                        double max_length = 0;
                        for (int a = 0; a < 3; a++) {
                            double length = (hi[a] + EPSILON) - (lo[a] - EPSILON);
                            if (length >= max_length) {
                                axis = a;
                                max_length = length;
                            }
                        }
                        std::sort(sorted.begin(), sorted.end(), [verts, axis](int a, int b) {
                            double ca = (verts[9 * a + axis] + verts[9 * a + 3 + axis] + verts[9 * a + 6 + axis]) / 3.0;
                            double cb = (verts[9 * b + axis] + verts[9 * b + 3 + axis] + verts[9 * b + 6 + axis]) / 3.0;
                            return ca < cb;
                        });
                    }
                    for (int64_t k = 0; k < n; k++)
                        light_tri.push_back(static_cast<int32_t>(base) + leaf_position[static_cast<size_t>(sorted[static_cast<size_t>(k)])]);
                }
                break;
            }
            default: return fail(PTK_ERR_INVALID, "ptk_scene_create: unknown object type");
        }
        objects[static_cast<size_t>(i)] = o;
    }

    // camera (Camera::Camera src/camera.cpp:4-22)
    const PtkCamera& c = d->camera;
    H3 view = rotate_yzx(H3{0, -1, 0}, c.orientation_y, c.orientation_z, c.orientation_x);
    H3 screen_y = rotate_yzx(H3{0, 0, -1}, c.orientation_y, c.orientation_z, c.orientation_x);
    H3 screen_x = hcross(view, screen_y);
    std::memcpy(dev.cam.pos, c.position, 24);
    put(dev.cam.view, view);
    put(dev.cam.screen_x, screen_x);
    put(dev.cam.screen_y, screen_y);
    dev.cam.width = c.camera_width;
    dev.cam.screen_height = c.camera_width * static_cast<double>(d->settings.height) / static_cast<double>(d->settings.width);
    dev.cam.focal = c.focal_length;
    dev.cam.aperture = c.aperture_size;
    dev.cam.focus = c.focus_distance;
    dev.cam.dof_mode = c.dof_mode;

    dev.n_objects = d->n_objects;
    dev.n_materials = d->n_materials;
    dev.n_media = d->n_media;
    dev.n_maps = d->n_maps;
    dev.n_lights = static_cast<int32_t>(lights.size());
    dev.has_triangles = !tri_hot.empty();
    // Material::allow_direct_light (src/materials.cpp:55-60, :129-136) can only say yes through a transparency above
    // zero somewhere in the material's transparency map, or for a Transparent material between media of equal
    // refractive index.
    auto lets_light_through = [&](const DevMaterial& m) {
        double most = 0;
        if (m.transparency >= 0) {
            const PtkMap& map = d->maps[m.transparency];
            const int64_t texels = static_cast<int64_t>(map.width) * map.height;
            for (int64_t k = 0; k < texels; k++) most = std::max(most, d->map_data[map.offset + k]);
        }
        auto ior = [&](int medium) { return medium >= 0 ? media[static_cast<size_t>(medium)].ior : 1.0; };
        return most > 0 || (m.type == PTK_MAT_TRANSPARENT && ior(m.internal_medium) == ior(m.external_medium));
    };
    // A connection to a point on an emissive mesh has no other object's distance to bound it (an analytic light
    // is found by the object scan, which then limits the walk), so a triangle of another mesh beyond the light
    // would pass for a blocker if the walk stopped at the first one it finds: such scenes walk every mesh exactly.
    bool has_mesh_light = false;
    for (const DevObject& o : objects)
        if (o.type == PTK_OBJ_MESH && o.mat.is_light) has_mesh_light = true;
    std::vector<DevShape> shapes(objects.size());
    for (size_t i = 0; i < objects.size(); i++) {
        const DevObject& o = objects[i];
        DevShape& sh = shapes[i];
        std::memset(&sh, 0, sizeof(sh));
        sh.type = o.type;
        sh.index = static_cast<int32_t>(i);
        sh.first_tri = static_cast<int32_t>(o.first_tri);
        sh.bvh_root = o.bvh_root;
        if (o.type == PTK_OBJ_SPHERE) {
            sh.f[0] = o.px;
            sh.f[1] = o.py;
            sh.f[2] = o.pz;
            sh.f[3] = o.radius;
        }
        else if (o.type == PTK_OBJ_MESH && o.bvh_root >= 0) { // bounds of the whole mesh = union of the root's children
            const BvhNode4& root = nodes[static_cast<size_t>(o.bvh_root)];
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int c = 0; c < BVH_WIDTH; c++) {
                if (root.child[c] == BVH_EMPTY) continue;
                lo[0] = std::min(lo[0], static_cast<double>(root.lo_x[c]));
                lo[1] = std::min(lo[1], static_cast<double>(root.lo_y[c]));
                lo[2] = std::min(lo[2], static_cast<double>(root.lo_z[c]));
                hi[0] = std::max(hi[0], static_cast<double>(root.hi_x[c]));
                hi[1] = std::max(hi[1], static_cast<double>(root.hi_y[c]));
                hi[2] = std::max(hi[2], static_cast<double>(root.hi_z[c]));
            }
            for (int a = 0; a < 3; a++) {
                sh.f[a] = lo[a];
                sh.f[3 + a] = hi[a];
            }
            // Material::allow_direct_light (src/materials.cpp:55-60, :129-136) can only say yes through a
            // transparency above zero or a Transparent material between media of equal refractive index;
            // a light has to be found exactly (its distance is compared with the sampled point's).
            const DevMaterial& m = o.mat;
            sh.f[6] = (!lets_light_through(m) && !m.is_light && !has_mesh_light) ? 1.0 : 0.0;
        }
        else if (o.type == PTK_OBJ_PLANE || o.type == PTK_OBJ_RECTANGLE) {
            // (the half sides with the test's slack: the same two IEEE operations the test would perform)
            const double v[14] = {o.px, o.py, o.pz, o.nx, o.ny, o.nz, o.v1x, o.v1y, o.v1z, o.v2x, o.v2y, o.v2z, o.L1 / 2.0 + EPSILON,
                                  o.L2 / 2.0 + EPSILON};
            std::memcpy(sh.f, v, sizeof(v));
        }
    }
    // The tables of the closest-hit loop (ScanHeader in scene_device.h): runs of one type each, so that the loop
    // over a run has no dispatch on the type and reads 64 bytes per object.  The first object of the table wins an
    // exact tie in the reference; the order is kept within a run and the earlier run wins between runs, so a
    // rectangle along the axes stays out of its run (and is tested in table order with the rest) when a plane or
    // rectangle of another kind lies in its plane: those are the objects that can tie with it bit for bit.
    ScanHeader scan_header{};
    std::vector<ShapeUnit> units[4]; // axis x, y, z; spheres
    std::vector<DevShape> others;
    {
        struct AxisPlane {
            int axis;
            double at;
        };
        auto axis_rectangle = [&](const DevObject& o, int& kn, int& k1) {
            kn = unit_axis(o.nx, o.ny, o.nz);
            k1 = unit_axis(o.v1x, o.v1y, o.v1z);
            const int k2 = unit_axis(o.v2x, o.v2y, o.v2z);
            return o.type == PTK_OBJ_RECTANGLE && kn >= 0 && k1 >= 0 && k2 >= 0 && kn != k1 && kn != k2 && k1 != k2;
        };
        const bool runs = g_scan_runs.load();
        std::vector<AxisPlane> taken; // planes along the axes that hold a plane or a rectangle of another kind
        for (const DevObject& o : objects) {
            int kn, k1;
            if ((o.type == PTK_OBJ_PLANE || o.type == PTK_OBJ_RECTANGLE) && !axis_rectangle(o, kn, k1) && kn >= 0) {
                const double p[3] = {o.px, o.py, o.pz};
                taken.push_back(AxisPlane{kn, p[kn]});
            }
        }
        std::vector<DevShape> meshes;
        for (size_t i = 0; i < objects.size(); i++) {
            const DevObject& o = objects[i];
            int kn = -1, k1 = -1;
            ShapeUnit unit{};
            unit.index = static_cast<int32_t>(i);
            if (runs && o.type == PTK_OBJ_SPHERE) {
                std::memcpy(unit.f, shapes[i].f, 4 * sizeof(double));
                units[3].push_back(unit);
            }
            else if (runs && axis_rectangle(o, kn, k1)) {
                const double p[3] = {o.px, o.py, o.pz};
                bool shared = false;
                for (const AxisPlane& plane : taken) shared = shared || (plane.axis == kn && plane.at == p[kn]);
                if (shared) {
                    others.push_back(shapes[i]);
                    continue;
                }
                const int j = (kn + 1) % 3, m = (kn + 2) % 3;
                unit.f[0] = p[kn];
                unit.f[1] = p[j];
                unit.f[2] = p[m];
                unit.f[3] = (k1 == j ? o.L1 : o.L2) / 2.0 + EPSILON;
                unit.f[4] = (k1 == j ? o.L2 : o.L1) / 2.0 + EPSILON;
                units[kn].push_back(unit);
                dev.has_axis_rectangles = 1;
            }
            else if (runs && o.type == PTK_OBJ_MESH) meshes.push_back(shapes[i]); // (walked with the tightest limit)
            else others.push_back(shapes[i]);
        }
        others.insert(others.end(), meshes.begin(), meshes.end());
        for (int k = 0; k < 3; k++) scan_header.n_axis[k] = static_cast<int32_t>(units[k].size());
        scan_header.n_spheres = static_cast<int32_t>(units[3].size());
        scan_header.n_others = static_cast<int32_t>(others.size());
    }
    dev.mesh_bound_abs = 0;
    for (const DevShape& sh : shapes)
        if (sh.type == PTK_OBJ_MESH && sh.bvh_root >= 0)
            for (int k = 0; k < 6; k++) dev.mesh_bound_abs = std::max(dev.mesh_bound_abs, static_cast<float>(std::fabs(sh.f[k])) * (1.0f + 0x1p-20f));
    dev.has_meshes = 0;
    dev.needs_uv = 0;
    dev.has_pass_through = 0;
    dev.tables_fit_lds = (objects.size() + lights.size()) * sizeof(DevObject) + media.size() * sizeof(DevMedium) <=
                         static_cast<size_t>(LDS_SCENE_DOUBLES) * sizeof(double);
    for (const DevMaterial& m : materials)
        if (lets_light_through(m)) dev.has_pass_through = 1;
    dev.has_transparency = 0;
    for (const DevMaterial& m : materials) {
        if (m.transparency < 0) continue;
        const PtkMap& map = d->maps[m.transparency];
        const int64_t texels = static_cast<int64_t>(map.width) * map.height;
        for (int64_t k = 0; k < texels; k++)
            if (!(d->map_data[map.offset + k] <= 0)) dev.has_transparency = 1; // (a NaN texel keeps the draws too)
    }
    for (const DevMaterial& m : materials)
        if (m.needs_uv) dev.needs_uv = 1;
    for (const DevObject& o : objects)
        if (o.type == PTK_OBJ_MESH) dev.has_meshes = 1;
    dev.width = d->settings.width;
    dev.height = d->settings.height;
    dev.max_depth = d->settings.max_recursion_depth;
    dev.min_depth = d->settings.min_recursion_steps;
    dev.nee = d->settings.enable_next_event_estimation != 0;
    dev.aa = d->settings.enable_anti_aliasing != 0;
    dev.background_medium = d->background_medium;
    std::memcpy(dev.background, d->background_color, 24);
    dev.background_is_black = d->background_color[0] == 0 && d->background_color[1] == 0 && d->background_color[2] == 0;
    dev.class_mask = class_mask;
    for (int k = 0; k < N_CLASSES; k++)
        if ((class_mask >> k) & 1u) s->classes.push_back(k);

    HIP_TRY(buf.upload(&dev.objects, objects));
    {
        std::vector<unsigned char> scan(sizeof(ScanHeader) + (units[0].size() + units[1].size() + units[2].size() + units[3].size()) * sizeof(ShapeUnit) +
                                        others.size() * sizeof(DevShape));
        unsigned char* at = scan.data();
        auto append = [&at](const void* src, size_t bytes) {
            if (bytes) std::memcpy(at, src, bytes);
            at += bytes;
        };
        append(&scan_header, sizeof(scan_header));
        for (int k = 0; k < 4; k++) append(units[k].data(), units[k].size() * sizeof(ShapeUnit));
        append(others.data(), others.size() * sizeof(DevShape));
        const unsigned char* d_scan = nullptr;
        HIP_TRY(buf.upload(&d_scan, scan));
        dev.scan = reinterpret_cast<const ScanHeader*>(d_scan);
    }
    HIP_TRY(buf.upload(&dev.materials, materials));
    HIP_TRY(buf.upload(&dev.media, media));
    HIP_TRY(buf.upload(&dev.maps, maps));
    HIP_TRY(buf.upload(&dev.map_data, map_data));
    HIP_TRY(buf.upload(&dev.lights, lights));
    std::vector<DevObject> light_objects;
    for (int32_t i : lights) light_objects.push_back(objects[static_cast<size_t>(i)]);
    HIP_TRY(buf.upload(&dev.light_objects, light_objects));
    if (!tri_hot.empty()) tri_hot.push_back(TriHot{}); // k_traverse fetches 80 bytes where a triangle begins: eight past the last one
    HIP_TRY(buf.upload(&dev.tri_hot, tri_hot));
    HIP_TRY(buf.upload(&dev.tri_cold, tri_cold));
    HIP_TRY(buf.upload(&dev.nodes, nodes));
    {
        std::vector<BvhNodeQuad> quads(nodes.size()); // the same nodes in the layout of k_traverse_quad
        for (size_t i = 0; i < nodes.size() && BVH_WIDTH == 4; i++) // (the walk with four lanes per ray knows nodes of four only)
            for (int c = 0; c < 4; c++) {
                BvhChildBox& box = quads[i].c[c];
                box.lo[0] = nodes[i].lo_x[c];
                box.lo[1] = nodes[i].lo_y[c];
                box.lo[2] = nodes[i].lo_z[c];
                box.hi[0] = nodes[i].hi_x[c];
                box.hi[1] = nodes[i].hi_y[c];
                box.hi[2] = nodes[i].hi_z[c];
                box.child = nodes[i].child[c];
                box._pad = 0;
            }
        HIP_TRY(buf.upload(&dev.nodes_quad, quads));
    }
    std::vector<DevMeshRef> mesh_refs; // what k_traverse needs to start the walk of each mesh
    for (size_t i = 0; i < shapes.size(); i++)
        if (shapes[i].type == PTK_OBJ_MESH && shapes[i].bvh_root >= 0)
            mesh_refs.push_back(DevMeshRef{static_cast<int32_t>(i), shapes[i].bvh_root, shapes[i].first_tri, shapes[i].f[6] != 0 ? 1 : 0});
    dev.n_meshes = static_cast<int32_t>(mesh_refs.size());
    HIP_TRY(buf.upload(&dev.meshes, mesh_refs));
    HIP_TRY(buf.upload(&dev.light_cumulative_area, light_area));
    HIP_TRY(buf.upload(&dev.light_tri_index, light_tri));
    HIP_TRY(buf.alloc(&dev.stack_overflows, 1));
    HIP_TRY(hipMemset(dev.stack_overflows, 0, sizeof(uint32_t)));
    *out = s.release();
    return PTK_OK;
    PTK_GUARD_END
}

int ptk_scene_clone(PtkScene* source, PtkScene** out) {
    if (!source || !out) return fail(PTK_ERR_INVALID, "ptk_scene_clone: null argument");
    *out = nullptr;
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    PTK_GUARD_BEGIN
    std::unique_ptr<PtkScene> s(new PtkScene());
    s->ctx = ctx;
    s->dev = source->dev;
    s->settings = source->settings;
    s->max_paths = source->max_paths;
    s->classes = source->classes;
    s->bvh_build_ms = source->bvh_build_ms;
    s->bvh_build_wall_ms = source->bvh_build_wall_ms;
    s->bvh_meshes_on_gpu = source->bvh_meshes_on_gpu;
    s->bvh_meshes_on_host = source->bvh_meshes_on_host;
    s->bvh_nodes = source->bvh_nodes;
    // every table of the source, copied device to device; the description's pointers are moved to the copies
    const DeviceBuffers& from = source->scene_buffers;
    DeviceBuffers& to = s->scene_buffers;
    if (use(source->ctx) != PTK_OK) return PTK_ERR_DEVICE;
    HIP_TRY(hipStreamSynchronize(source->ctx->stream));
    if (use(ctx) != PTK_OK) return PTK_ERR_DEVICE;
    for (size_t i = 0; i < from.allocations.size(); i++) {
        unsigned char* copy = nullptr;
        HIP_TRY(to.alloc(&copy, from.bytes[i]));
        if (source->ctx->device == ctx->device) HIP_TRY(hipMemcpyAsync(copy, from.allocations[i], from.bytes[i], hipMemcpyDeviceToDevice, ctx->stream));
        else HIP_TRY(hipMemcpyPeerAsync(copy, ctx->device, from.allocations[i], source->ctx->device, from.bytes[i], ctx->stream));
    }
    bool all_found = true;
    auto moved = [&](auto*& pointer) {
        if (!pointer) return;
        const unsigned char* p = reinterpret_cast<const unsigned char*>(pointer);
        for (size_t i = 0; i < from.allocations.size(); i++) {
            const unsigned char* base = static_cast<const unsigned char*>(from.allocations[i]);
            if (p >= base && p < base + from.bytes[i]) {
                pointer = reinterpret_cast<std::remove_reference_t<decltype(pointer)>>(static_cast<unsigned char*>(to.allocations[i]) + (p - base));
                return;
            }
        }
        all_found = false;
    };
    DevScene& dev = s->dev;
    moved(dev.objects);
    moved(dev.scan);
    moved(dev.materials);
    moved(dev.media);
    moved(dev.maps);
    moved(dev.map_data);
    moved(dev.lights);
    moved(dev.light_objects);
    moved(dev.tri_hot);
    moved(dev.tri_cold);
    moved(dev.nodes);
    moved(dev.nodes_quad);
    moved(dev.light_cumulative_area);
    moved(dev.light_tri_index);
    moved(dev.meshes);
    moved(dev.stack_overflows);
    if (!all_found) return fail(PTK_ERR_INVALID, "ptk_scene_clone: a table of the source scene lies outside its buffers");
    HIP_TRY(hipMemsetAsync(dev.stack_overflows, 0, sizeof(uint32_t), ctx->stream)); // the copy counts its own
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = s.release();
    return PTK_OK;
    PTK_GUARD_END
}

void ptk_scene_destroy(PtkScene* scene) {
    if (!scene) return;
    if (scene->ctx && hipSetDevice(scene->ctx->device) == hipSuccess) (void) hipStreamSynchronize(scene->ctx->stream);
    delete scene;
}

int ptk_trace_primary(PtkScene* s, const PtkRenderJob* job, int32_t* object_index, int32_t* primitive_index, double* distance,
                      double* position, double* normal) {
    PTK_GUARD_BEGIN
    int rc = check_job(s, job, "ptk_trace_primary");
    if (rc != PTK_OK) return rc;
    if ((rc = use(s->ctx)) != PTK_OK) return rc;
    const hipStream_t g_stream = s->ctx->stream;
    const size_t n = static_cast<size_t>(job->number_of_pixels);
    if (n == 0) return PTK_OK;
    DeviceBuffers tmp;
    int32_t *d_obj, *d_prim;
    double *d_dist, *d_pos, *d_nrm;
    HIP_TRY(tmp.alloc(&d_obj, n));
    HIP_TRY(tmp.alloc(&d_prim, n));
    HIP_TRY(tmp.alloc(&d_dist, n));
    HIP_TRY(tmp.alloc(&d_pos, 3 * n));
    HIP_TRY(tmp.alloc(&d_nrm, 3 * n));
    BatchParams b{};
    b.start_idx = job->start_idx;
    b.n_pixels = static_cast<uint32_t>(n);
    b.n_samples = 1;
    b.sample_offset = static_cast<uint32_t>(job->sample_offset);
    b.n_paths = b.n_pixels;
    b.seed = job->seed;
    launch_primary(g_stream, s->dev, b, d_obj, d_prim, d_dist, d_pos, d_nrm);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (object_index) HIP_TRY(hipMemcpy(object_index, d_obj, n * 4, hipMemcpyDeviceToHost));
    if (primitive_index) HIP_TRY(hipMemcpy(primitive_index, d_prim, n * 4, hipMemcpyDeviceToHost));
    if (distance) HIP_TRY(hipMemcpy(distance, d_dist, n * 8, hipMemcpyDeviceToHost));
    if (position) HIP_TRY(hipMemcpy(position, d_pos, n * 24, hipMemcpyDeviceToHost));
    if (normal) HIP_TRY(hipMemcpy(normal, d_nrm, n * 24, hipMemcpyDeviceToHost));
    return PTK_OK;
    PTK_GUARD_END
}

int ptk_trace_samples(PtkScene* s, const PtkRenderJob* job, double* color, double* position, double* normal) {
    PTK_GUARD_BEGIN
    int rc = check_job(s, job, "ptk_trace_samples");
    if (rc != PTK_OK) return rc;
    if ((rc = use(s->ctx)) != PTK_OK) return rc;
    const hipStream_t g_stream = s->ctx->stream;
    const size_t total = static_cast<size_t>(job->number_of_pixels) * static_cast<size_t>(job->number_of_samples) * 3;
    if (total == 0) return PTK_OK;
    DeviceBuffers tmp;
    double *d_color = nullptr, *d_pos = nullptr, *d_nrm = nullptr;
    if (color) HIP_TRY(tmp.alloc(&d_color, total));
    if (position) HIP_TRY(tmp.alloc(&d_pos, total));
    if (normal) HIP_TRY(tmp.alloc(&d_nrm, total));
    rc = for_each_batch(s, job, [&](BatchLane& lane, const BatchParams& b, int64_t pixel_base) {
        launch_gather_samples(lane.stream, b, lane.results, static_cast<size_t>(pixel_base), d_color, d_pos, d_nrm);
        HIP_TRY(hipGetLastError());
        return static_cast<int>(PTK_OK);
    });
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (color) HIP_TRY(hipMemcpy(color, d_color, total * 8, hipMemcpyDeviceToHost));
    if (position) HIP_TRY(hipMemcpy(position, d_pos, total * 8, hipMemcpyDeviceToHost));
    if (normal) HIP_TRY(hipMemcpy(normal, d_nrm, total * 8, hipMemcpyDeviceToHost));
    return PTK_OK;
    PTK_GUARD_END
}

static int film_create(DeviceContext* ctx, int32_t width, int32_t height, PtkFilm** out) {
    PTK_GUARD_BEGIN
    std::unique_ptr<PtkFilm> f(new PtkFilm());
    f->ctx = ctx;
    f->width = width;
    f->height = height;
    const size_t n = static_cast<size_t>(width) * static_cast<size_t>(height);
    PtkFilm* film = f.release(); // from here on ptk_film_destroy frees whatever has been allocated
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&film->image), n * 24);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&film->position), n * 24);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&film->normal), n * 24);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&film->counters), n * 8);
    int rc = e == hipSuccess ? ptk_film_clear(film) : fail(PTK_ERR_MEMORY, std::string("ptk_film_create: ") + hipGetErrorString(e));
    if (rc != PTK_OK) {
        const std::string message = g_error;
        ptk_film_destroy(film);
        g_error = message;
        return rc;
    }
    *out = film;
    return PTK_OK;
    PTK_GUARD_END
}

int ptk_film_create(int32_t width, int32_t height, PtkFilm** out) {
    if (!out || width < 1 || height < 1) return fail(PTK_ERR_INVALID, "ptk_film_create: bad argument");
    *out = nullptr;
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    return film_create(ctx, width, height, out);
}

int ptk_film_wrap(int32_t width, int32_t height, double* d_image, double* d_position, double* d_normal, uint64_t* d_counters,
                  PtkFilm** out) {
    if (!out || width < 1 || height < 1 || !d_image || !d_position || !d_normal || !d_counters)
        return fail(PTK_ERR_INVALID, "ptk_film_wrap: bad argument");
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    PtkFilm* f = new (std::nothrow) PtkFilm();
    if (!f) return fail(PTK_ERR_MEMORY, "out of host memory");
    f->ctx = ctx;
    f->width = width;
    f->height = height;
    f->owns_buffers = false;
    f->image = d_image;
    f->position = d_position;
    f->normal = d_normal;
    f->counters = reinterpret_cast<unsigned long long*>(d_counters);
    *out = f;
    return PTK_OK;
}

void ptk_film_destroy(PtkFilm* f) {
    if (!f) return;
    if (f->ctx && hipSetDevice(f->ctx->device) == hipSuccess) (void) hipStreamSynchronize(f->ctx->stream);
    if (f->owns_buffers) {
        (void) hipFree(f->image);
        (void) hipFree(f->position);
        (void) hipFree(f->normal);
        (void) hipFree(f->counters);
    }
    (void) hipFree(f->stage_image); // (null unless the film took part in a merge across devices)
    (void) hipFree(f->stage_position);
    (void) hipFree(f->stage_normal);
    (void) hipFree(f->stage_counters);
    delete f;
}

int ptk_film_clear(PtkFilm* f) {
    if (!f) return fail(PTK_ERR_INVALID, "ptk_film_clear: null film");
    if (use(f->ctx) != PTK_OK) return PTK_ERR_DEVICE;
    const hipStream_t g_stream = f->ctx->stream;
    const size_t n = static_cast<size_t>(f->width) * static_cast<size_t>(f->height);
    HIP_TRY(hipMemsetAsync(f->image, 0, n * 24, g_stream));
    HIP_TRY(hipMemsetAsync(f->position, 0, n * 24, g_stream));
    HIP_TRY(hipMemsetAsync(f->normal, 0, n * 24, g_stream));
    HIP_TRY(hipMemsetAsync(f->counters, 0, n * 8, g_stream));
    return PTK_OK;
}

int ptk_film_upload(PtkFilm* f, const double* image, const double* position, const double* normal, const uint64_t* counters) {
    if (!f) return fail(PTK_ERR_INVALID, "ptk_film_upload: null film");
    if (use(f->ctx) != PTK_OK) return PTK_ERR_DEVICE;
    const hipStream_t g_stream = f->ctx->stream;
    const size_t n = static_cast<size_t>(f->width) * static_cast<size_t>(f->height);
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (image) HIP_TRY(hipMemcpy(f->image, image, n * 24, hipMemcpyHostToDevice));
    if (position) HIP_TRY(hipMemcpy(f->position, position, n * 24, hipMemcpyHostToDevice));
    if (normal) HIP_TRY(hipMemcpy(f->normal, normal, n * 24, hipMemcpyHostToDevice));
    if (counters) HIP_TRY(hipMemcpy(f->counters, counters, n * 8, hipMemcpyHostToDevice));
    return PTK_OK;
}

int ptk_film_download(PtkFilm* f, double* image, double* position, double* normal, uint64_t* counters) {
    if (!f) return fail(PTK_ERR_INVALID, "ptk_film_download: null film");
    if (use(f->ctx) != PTK_OK) return PTK_ERR_DEVICE;
    const hipStream_t g_stream = f->ctx->stream;
    const size_t n = static_cast<size_t>(f->width) * static_cast<size_t>(f->height);
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (image) HIP_TRY(hipMemcpy(image, f->image, n * 24, hipMemcpyDeviceToHost));
    if (position) HIP_TRY(hipMemcpy(position, f->position, n * 24, hipMemcpyDeviceToHost));
    if (normal) HIP_TRY(hipMemcpy(normal, f->normal, n * 24, hipMemcpyDeviceToHost));
    if (counters) HIP_TRY(hipMemcpy(counters, f->counters, n * 8, hipMemcpyDeviceToHost));
    return PTK_OK;
}

int ptk_film_merge(PtkFilm* into, PtkFilm* from) {
    PTK_GUARD_BEGIN
    if (!into || !from || into == from) return fail(PTK_ERR_INVALID, "ptk_film_merge: two different films are required");
    if (into->width != from->width || into->height != from->height) return fail(PTK_ERR_INVALID, "ptk_film_merge: the films differ in size");
    const size_t n = static_cast<size_t>(into->width) * static_cast<size_t>(into->height);
    // what has been queued for `from` is finished first: the copy or the kernel below runs on the other film's stream
    if (use(from->ctx) != PTK_OK) return PTK_ERR_DEVICE;
    HIP_TRY(hipStreamSynchronize(from->ctx->stream));
    if (use(into->ctx) != PTK_OK) return PTK_ERR_DEVICE;
    const hipStream_t stream = into->ctx->stream;
    const double *src_image = from->image, *src_position = from->position, *src_normal = from->normal;
    const unsigned long long* src_counters = from->counters;
    if (from->ctx->device != into->ctx->device) { // device to device (xGMI between the GPUs of a node), no host copy
        if (!into->stage_image) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&into->stage_image), n * 24));
        if (!into->stage_position) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&into->stage_position), n * 24));
        if (!into->stage_normal) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&into->stage_normal), n * 24));
        if (!into->stage_counters) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&into->stage_counters), n * 8));
        const int to = into->ctx->device, fr = from->ctx->device;
        HIP_TRY(hipMemcpyPeerAsync(into->stage_image, to, from->image, fr, n * 24, stream));
        HIP_TRY(hipMemcpyPeerAsync(into->stage_position, to, from->position, fr, n * 24, stream));
        HIP_TRY(hipMemcpyPeerAsync(into->stage_normal, to, from->normal, fr, n * 24, stream));
        HIP_TRY(hipMemcpyPeerAsync(into->stage_counters, to, from->counters, fr, n * 8, stream));
        src_image = into->stage_image;
        src_position = into->stage_position;
        src_normal = into->stage_normal;
        src_counters = into->stage_counters;
    }
    launch_merge_films(stream, n, into->image, into->position, into->normal, into->counters, src_image, src_position, src_normal, src_counters);
    HIP_TRY(hipGetLastError());
    // `from` may be rendered into again as soon as the call returns
    HIP_TRY(hipStreamSynchronize(stream));
    return PTK_OK;
    PTK_GUARD_END
}

int ptk_film_device_pointers(PtkFilm* f, double** image, double** position, double** normal, uint64_t** counters) {
    if (!f) return fail(PTK_ERR_INVALID, "ptk_film_device_pointers: null film");
    if (image) *image = f->image;
    if (position) *position = f->position;
    if (normal) *normal = f->normal;
    if (counters) *counters = reinterpret_cast<uint64_t*>(f->counters);
    return PTK_OK;
}

int ptk_render(PtkScene* s, PtkFilm* film, const PtkRenderJob* job) {
    PTK_GUARD_BEGIN
    int rc = check_job(s, job, "ptk_render");
    if (rc != PTK_OK) return rc;
    if (!film || film->width != s->settings.width || film->height != s->settings.height)
        return fail(PTK_ERR_INVALID, "ptk_render: film size does not match the scene");
    if (film->ctx != s->ctx) return fail(PTK_ERR_INVALID, "ptk_render: the film and the scene live on different devices");
    if ((rc = use(s->ctx)) != PTK_OK) return rc;
    const hipStream_t g_stream = s->ctx->stream;
    return for_each_batch(s, job, [&](BatchLane& lane, const BatchParams& b, int64_t) {
        launch_accumulate(lane.stream, b, lane.results, film->image, film->position, film->normal, film->counters);
        HIP_TRY(hipGetLastError());
        lane.stats.algorithmic_bytes += ACCUMULATE_BYTES_PER_PATH * b.n_paths;
        return static_cast<int>(PTK_OK);
    });
    PTK_GUARD_END
}

int ptk_render_bands(PtkScene* s, PtkFilm* film, const PtkRenderJob* job, int64_t band_pixels, int64_t band_stride) {
    PTK_GUARD_BEGIN
    if (!s || !job) return fail(PTK_ERR_INVALID, "ptk_render_bands: null argument");
    if (band_pixels < 1 || band_stride < band_pixels || band_stride > (int64_t(1) << 31))
        return fail(PTK_ERR_INVALID, "ptk_render_bands: need 1 <= band_pixels <= band_stride");
    if (job->number_of_pixels < 0 || job->start_idx < 0) return fail(PTK_ERR_INVALID, "ptk_render_bands: negative pixel range");
    if (job->number_of_pixels > 0) { // the last pixel of the last (possibly partial) band must lie inside the image
        const int64_t last = job->number_of_pixels - 1;
        const int64_t last_pixel = job->start_idx + (last / band_pixels) * band_stride + last % band_pixels;
        if (last_pixel >= static_cast<int64_t>(s->settings.width) * s->settings.height)
            return fail(PTK_ERR_INVALID, "ptk_render_bands: pixel range outside the image");
    }
    PtkRenderJob contiguous = *job; // the remaining checks are those of a contiguous job of as many pixels
    contiguous.start_idx = 0;
    int rc = check_job(s, &contiguous, "ptk_render_bands");
    if (rc != PTK_OK) return rc;
    if (!film || film->width != s->settings.width || film->height != s->settings.height)
        return fail(PTK_ERR_INVALID, "ptk_render_bands: film size does not match the scene");
    if (film->ctx != s->ctx) return fail(PTK_ERR_INVALID, "ptk_render_bands: the film and the scene live on different devices");
    if ((rc = use(s->ctx)) != PTK_OK) return rc;
    const hipStream_t g_stream = s->ctx->stream;
    return for_each_batch(
        s, job,
        [&](BatchLane& lane, const BatchParams& b, int64_t) {
            launch_accumulate(lane.stream, b, lane.results, film->image, film->position, film->normal, film->counters);
            HIP_TRY(hipGetLastError());
            lane.stats.algorithmic_bytes += ACCUMULATE_BYTES_PER_PATH * b.n_paths;
            return static_cast<int>(PTK_OK);
        },
        static_cast<uint32_t>(band_pixels), static_cast<uint32_t>(band_stride));
    PTK_GUARD_END
}

int ptk_render_host(PtkScene* s, const PtkRenderJob* job, double* image, double* position, double* normal, uint64_t* counters) {
    PTK_GUARD_BEGIN
    int rc = check_job(s, job, "ptk_render_host");
    if (rc != PTK_OK) return rc;
    if (!image || !position || !normal || !counters) return fail(PTK_ERR_INVALID, "ptk_render_host: null buffer");
    if ((rc = use(s->ctx)) != PTK_OK) return rc;
    const hipStream_t g_stream = s->ctx->stream;
    if (!s->host_film) {
        rc = film_create(s->ctx, s->settings.width, s->settings.height, &s->host_film);
        if (rc != PTK_OK) return rc;
    }
    PtkFilm* f = s->host_film;
    const size_t first = static_cast<size_t>(job->start_idx), n = static_cast<size_t>(job->number_of_pixels);
    if (n == 0) return PTK_OK;
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(f->image + 3 * first, image + 3 * first, n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(f->position + 3 * first, position + 3 * first, n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(f->normal + 3 * first, normal + 3 * first, n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(f->counters + first, counters + first, n * 8, hipMemcpyHostToDevice));
    rc = ptk_render(s, f, job);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(image + 3 * first, f->image + 3 * first, n * 24, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(position + 3 * first, f->position + 3 * first, n * 24, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(normal + 3 * first, f->normal + 3 * first, n * 24, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(counters + first, f->counters + first, n * 8, hipMemcpyDeviceToHost));
    return PTK_OK;
    PTK_GUARD_END
}

const char* ptk_scene_bounce_kernel(PtkScene* s) {
    static thread_local std::string name;
    if (!s) return "";
    name = bounce_kernel_name(s->dev, g_specialised.load());
    return name.c_str();
}

int ptk_scene_build_wall_ms(PtkScene* s, double* wall_ms) {
    if (!s || !wall_ms) return fail(PTK_ERR_INVALID, "ptk_scene_build_wall_ms: null argument");
    *wall_ms = s->bvh_build_wall_ms;
    return PTK_OK;
}

int ptk_scene_build_info(PtkScene* s, double* bvh_build_ms, int32_t* meshes_built_on_gpu, int32_t* meshes_built_on_host, int64_t* bvh_nodes) {
    if (!s) return fail(PTK_ERR_INVALID, "ptk_scene_build_info: null scene");
    if (bvh_build_ms) *bvh_build_ms = s->bvh_build_ms;
    if (meshes_built_on_gpu) *meshes_built_on_gpu = s->bvh_meshes_on_gpu;
    if (meshes_built_on_host) *meshes_built_on_host = s->bvh_meshes_on_host;
    if (bvh_nodes) *bvh_nodes = s->bvh_nodes;
    return PTK_OK;
}

int ptk_get_stats(PtkScene* s, PtkStats* stats) {
    if (!s || !stats) return fail(PTK_ERR_INVALID, "ptk_get_stats: null argument");
    *stats = s->stats;
    return PTK_OK;
}

int ptk_reset_stats(PtkScene* s) {
    if (!s) return fail(PTK_ERR_INVALID, "ptk_reset_stats: null argument");
    s->stats = PtkStats{};
    return PTK_OK;
}

int ptk_selftest_record_staging(int64_t n_records, int64_t* mismatches) {
    PTK_GUARD_BEGIN
    if (!mismatches || n_records < 1 || n_records > (int64_t(1) << 22)) return fail(PTK_ERR_INVALID, "ptk_selftest_record_staging: 1 to 2^22 records");
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    const uint32_t n = static_cast<uint32_t>(n_records);
    const size_t padded = (static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255);
    const size_t pool_bytes = padded * RECORD_BYTES;
    auto offset_of = [](uint32_t i, int plane) { // the layout of PathPool (scene_device.h)
        return static_cast<size_t>(i / RECORD_TILE) * RECORD_TILE_BYTES + (i % RECORD_TILE) * 16u + static_cast<size_t>(plane) * RECORD_PLANE_STRIDE;
    };
    std::vector<uint32_t> words(pool_bytes / 4);
    for (uint32_t i = 0; i < n; i++)
        for (int plane = 0; plane < N_PLANES; plane++)
            for (uint32_t w = 0; w < 4; w++) words[offset_of(i, plane) / 4 + w] = (i * 2654435761u) ^ (static_cast<uint32_t>(plane) * 0x9E3779B9u + w * 0x85EBCA6Bu);
    std::vector<uint32_t> list(n);
    uint32_t state = 12345u;
    for (uint32_t i = 0; i < n; i++) { // arbitrary sources, as a class-sorted list has them
        state = state * 1664525u + 1013904223u;
        list[i] = (state >> 8) % n;
    }
    const uint32_t noise_mask = (1u << 16) - 1;
    std::vector<double> noise(noise_mask + 1, 1.0);
    DeviceBuffers tmp;
    PathPool in{}, out{};
    uint32_t* d_list = nullptr;
    double *d_noise = nullptr, *d_sink = nullptr;
    const uint32_t blocks = std::min<uint32_t>((n + 255) / 256, 4 * ctx->compute_units);
    HIP_TRY(tmp.alloc(&in.base, pool_bytes));
    HIP_TRY(tmp.alloc(&out.base, pool_bytes));
    HIP_TRY(tmp.alloc(&d_list, n));
    HIP_TRY(tmp.alloc(&d_noise, noise.size()));
    HIP_TRY(tmp.alloc(&d_sink, static_cast<size_t>(blocks) * 256));
    HIP_TRY(hipMemcpy(in.base, words.data(), pool_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(out.base, 0, pool_bytes));
    HIP_TRY(hipMemcpy(d_list, list.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_noise, noise.data(), noise.size() * sizeof(double), hipMemcpyHostToDevice));
    launch_selftest_staging(ctx->stream, blocks, in, out, d_list, n, d_noise, noise_mask, d_sink);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> got(pool_bytes / 4);
    HIP_TRY(hipMemcpy(got.data(), out.base, pool_bytes, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (uint32_t i = 0; i < n; i++) {
        bool same = true;
        for (int plane = 0; plane < N_PLANES; plane++)
            for (uint32_t w = 0; w < 4; w++) {
                const uint32_t want = i % 5u == 3u ? 0u : words[offset_of(list[i], plane) / 4 + w]; // skipped lanes write nothing
                same = same && got[offset_of(i, plane) / 4 + w] == want;
            }
        if (!same) bad++;
    }
    *mismatches = bad;
    return PTK_OK;
    PTK_GUARD_END
}

// ---- denoise --------------------------------------------------------------------------------------
} // extern "C"

namespace {
// The pipeline on device buffers (both device entry points): every kernel reads one image buffer and writes the other,
// every pixel of it, so the scratch buffer needs no particular contents on entry.  Allocates nothing.
int denoise_on_device(DeviceContext* ctx, const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t width, int32_t height, double* d_image,
                      const double* d_position, const double* d_normal, double* d_scratch) {
    const hipStream_t g_stream = ctx->stream;
    if ((!tasks && n_tasks > 0) || n_tasks < 0 || width < 1 || height < 1 || !d_image || !d_position || !d_normal || !d_scratch)
        return fail(PTK_ERR_INVALID, "ptk_denoise_device: bad argument");
    const size_t bytes = static_cast<size_t>(width) * static_cast<size_t>(height) * 24;
    // The whole pipeline is checked before its first kernel runs.  An a-trous tap lies up to 2^iterations pixels from
    // its centre in the last iteration, and the border rule reflects a coordinate once (src/denoise.cpp:20-38): a tap
    // further out than the image is wide or high lands outside the buffers (the reference reads there all the same).
    for (int32_t t = 0; t < n_tasks; t++) {
        const PtkDenoiseTask& task = tasks[t];
        if (task.mode == PTK_DENOISE_MEDIAN) {
            if (task.kernel_size < 1 || task.kernel_size % 2 == 0 || task.kernel_size > 15)
                return fail(PTK_ERR_INVALID, "ptk_denoise: median kernel_size must be odd and at most 15");
            if (task.kernel_size > std::min(width, height))
                return fail(PTK_ERR_INVALID, "Median filter kernel size is too large for this image size.");
            continue;
        }
        if (task.mode != PTK_DENOISE_ATROUS) return fail(PTK_ERR_INVALID, "ptk_denoise: unknown mode");
        if (task.iterations < 0 || task.iterations > 30) return fail(PTK_ERR_INVALID, "ptk_denoise: a-trous iterations must be between 0 and 30");
        if (task.iterations > 0 && (1ll << task.iterations) > std::min(width, height) - 1)
            return fail(PTK_ERR_INVALID, "ptk_denoise: the a-trous taps of the last iteration (2^iterations pixels out) reach beyond one reflection at the "
                                         "border of an image this small");
    }
    double* src = d_image;
    double* dst = d_scratch;
    for (int32_t t = 0; t < n_tasks; t++) { // denoise() src/denoise.cpp:203-212
        const PtkDenoiseTask& task = tasks[t];
        if (task.mode == PTK_DENOISE_ATROUS) { // atrous_filter :135-147
            double sigma_rt = task.sigma_rt, sigma_x = task.sigma_x, sigma_n = task.sigma_n;
            int hole_width = 0;
            for (int it = 0; it < task.iterations; it++) {
                launch_atrous(g_stream, width, height, src, dst, d_position, d_normal, sigma_rt, sigma_x, sigma_n, hole_width);
                HIP_TRY(hipGetLastError());
                std::swap(src, dst);
                sigma_rt /= 2.0;
                sigma_x /= 2.0;
                sigma_n /= 2.0;
                hole_width += static_cast<int>(std::pow(2, it));
            }
        }
        else if (task.mode == PTK_DENOISE_MEDIAN) {
            if (task.kernel_size < 1 || task.kernel_size % 2 == 0 || task.kernel_size > 15)
                return fail(PTK_ERR_INVALID, "ptk_denoise: median kernel_size must be odd and at most 15");
            if (task.kernel_size > std::min(width, height))
                return fail(PTK_ERR_INVALID, "Median filter kernel size is too large for this image size.");
            launch_median(g_stream, width, height, src, dst, task.kernel_size, task.threshold);
            HIP_TRY(hipGetLastError());
            std::swap(src, dst);
        }
        else {
            return fail(PTK_ERR_INVALID, "ptk_denoise: unknown mode");
        }
    }
    if (src != d_image) HIP_TRY(hipMemcpyAsync(d_image, src, bytes, hipMemcpyDeviceToDevice, g_stream));
    return PTK_OK;
}
} // namespace

extern "C" {

int ptk_denoise_device(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t width, int32_t height, double* d_image,
                       const double* d_position, const double* d_normal, double* d_scratch) {
    PTK_GUARD_BEGIN
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    return denoise_on_device(ctx, tasks, n_tasks, width, height, d_image, d_position, d_normal, d_scratch);
    PTK_GUARD_END
}

int ptk_denoise_device_static(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t width, int32_t height, double* d_image,
                              const double* d_position, const double* d_normal) {
    PTK_GUARD_BEGIN
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    if (width < 1 || height < 1) return fail(PTK_ERR_INVALID, "ptk_denoise_device_static: empty image");
    if (static_cast<int64_t>(width) * height > STATIC_SCRATCH_PIXELS)
        return fail(PTK_ERR_MEMORY, "ptk_denoise_device_static: the image has more than 2097152 pixels, the size of the library's static "
                                    "scratch buffer; use ptk_denoise_device with a scratch buffer of your own");
    return denoise_on_device(ctx, tasks, n_tasks, width, height, d_image, d_position, d_normal, ctx->static_scratch);
    PTK_GUARD_END
}

int ptk_denoise(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t width, int32_t height, double* image,
                const double* position, const double* normal) {
    PTK_GUARD_BEGIN
    DeviceContext* ctx = current_context();
    if (!ctx) return PTK_ERR_DEVICE;
    const hipStream_t g_stream = ctx->stream;
    if ((!tasks && n_tasks > 0) || n_tasks < 0 || width < 1 || height < 1 || !image || !position || !normal)
        return fail(PTK_ERR_INVALID, "ptk_denoise: bad argument");
    const size_t n = static_cast<size_t>(width) * static_cast<size_t>(height) * 3;
    DeviceBuffers tmp;
    double *d_image, *d_pos, *d_nrm, *d_scratch;
    HIP_TRY(tmp.alloc(&d_image, n));
    HIP_TRY(tmp.alloc(&d_pos, n));
    HIP_TRY(tmp.alloc(&d_nrm, n));
    HIP_TRY(tmp.alloc(&d_scratch, n));
    HIP_TRY(hipMemcpy(d_image, image, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_pos, position, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_nrm, normal, n * 8, hipMemcpyHostToDevice));
    int rc = ptk_denoise_device(tasks, n_tasks, width, height, d_image, d_pos, d_nrm, d_scratch);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(image, d_image, n * 8, hipMemcpyDeviceToHost));
    return PTK_OK;
    PTK_GUARD_END
}

} // extern "C"
