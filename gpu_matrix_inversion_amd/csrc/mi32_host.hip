// mi32_host.hip -- host runtime of libmat_inv_32.so: context, workspace cache,
// the C ABI of include/mat_inv_32_c.h and the C++ drop-in of include/mat_inv_32.h.
//
// Replaces the host half of /root/reference/Matlab/mat_inv_32/mat_inv_32/
// mat_inv_32.cpp:206-395.  Where the reference re-creates platform, context,
// queue, six JIT-built programs and four buffers on every call (:238-290, 1.44 s
// of its 4.37 s at N=4096) and tears them down again (:388), this keeps one
// AOT-compiled code object, one stream and one grow-only workspace per context.
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "mat_inv_32.h"
#include "mat_inv_64.h"
#include "mat_inv_bench.h"
#include "mi32_internal.h"

using namespace mi32;

// Event-pair profiler: one (start, stop) pair per launch, summed per kernel class on demand.
struct EventProfiler : public Profiler {
    struct Rec { int k; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t cur = nullptr;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    // Records accumulate until mi32_get_profile() collects them: a caller that leaves profiling on and never
    // collects stops recording after kMaxRecs launches instead of growing without bound.
    static constexpr size_t kMaxRecs = 1u << 20;
    void begin(int, hipStream_t s) override
    {
        if (recs.size() >= kMaxRecs) { cur = nullptr; return; }
        cur = get();
        (void)hipEventRecord(cur, s);
    }
    void end(int k, hipStream_t s) override
    {
        if (!cur) return;
        hipEvent_t b = get();
        (void)hipEventRecord(b, s);
        recs.push_back({k, cur, b});
        cur = nullptr;
    }
    void collect(double *ms, long long *count)
    {
        for (auto &r : recs) {
            float t = 0.f;
            if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
                ms[r.k] += t;
                count[r.k] += 1;
            }
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        recs.clear();
    }
    ~EventProfiler() override
    {
        for (auto &r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

// Host <-> device copies of the host-pointer entry points.  The caller's vectors are pageable memory: a plain
// hipMemcpy moves 64 MiB (N = 4096) in 4.6 ms each way on this platform, while the DMA engine needs 1.2 ms from
// pinned memory and pinning the caller's pages (hipHostRegister) costs 4 ms by itself (tools/h2d_probe.hip).
// So: kLanes host threads, each with two pinned 2 MiB buffers and a stream of its own, memcpy chunk i + 1 into one
// buffer while the DMA engine drains chunk i from the other: N = 4096 end to end 16.8-21.6 -> 12.4 ms (8.8 ms of it
// compute), N = 8192 65 -> 52 ms.
struct HostCopier {
    static constexpr int kLanes = 6;
    static constexpr size_t kChunk = 2u << 20;
    static constexpr size_t kMinBytes = 32u << 20;  // below this a plain hipMemcpyAsync wins (measured cross-over)
    char *pin[kLanes][2] = {};
    hipStream_t stream[kLanes] = {};
    hipEvent_t ev[kLanes][2] = {};
    // the lanes are persistent threads (a fresh thread's first HIP call costs more than the copy it would do)
    std::thread th[kLanes];
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    unsigned long long job_id = 0;  // incremented per job; a lane runs job j when job_id == j > its last one
    int pending = 0;
    bool quit = false;
    int device = 0;
    void *j_dev = nullptr, *j_host = nullptr;
    size_t j_bytes = 0;
    bool j_to_device = true;
    hipError_t j_err[kLanes];
    bool ready = false;

    hipError_t init(int dev)
    {
        if (ready) return hipSuccess;
        device = dev;
        for (int t = 0; t < kLanes; ++t) {
            hipError_t e = hipStreamCreateWithFlags(&stream[t], hipStreamNonBlocking);
            if (e != hipSuccess) return e;
            for (int q = 0; q < 2; ++q) {
                if ((e = hipHostMalloc((void **)&pin[t][q], kChunk, hipHostMallocDefault)) != hipSuccess) return e;
                if ((e = hipEventCreateWithFlags(&ev[t][q], hipEventDisableTiming)) != hipSuccess) return e;
            }
        }
        for (int t = 0; t < kLanes; ++t) th[t] = std::thread([this, t]() { lane_main(t); });
        ready = true;
        return hipSuccess;
    }
    void destroy()
    {
        if (!ready) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_job.notify_all();
        for (int t = 0; t < kLanes; ++t)
            if (th[t].joinable()) th[t].join();
        for (int t = 0; t < kLanes; ++t) {
            for (int q = 0; q < 2; ++q) {
                if (pin[t][q]) (void)hipHostFree(pin[t][q]);
                if (ev[t][q]) (void)hipEventDestroy(ev[t][q]);
            }
            if (stream[t]) (void)hipStreamDestroy(stream[t]);
        }
        ready = false;
    }
    void lane_main(int t)
    {
        (void)hipSetDevice(device);
        unsigned long long done_id = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&]() { return quit || job_id != done_id; });
                if (quit) return;
                done_id = job_id;
            }
            j_err[t] = lane_copy(t);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    hipError_t lane_copy(int t)
    {
        const size_t bytes = j_bytes;
        const size_t nchunks = (bytes + kChunk - 1) / kChunk;
        char *dev = (char *)j_dev, *host = (char *)j_host;
        auto span = [&](size_t c, size_t &off, size_t &len) { off = c * kChunk; len = bytes - off < kChunk ? bytes - off : kChunk; };
        hipError_t e = hipSuccess;
        size_t k = 0;  // this lane's chunk counter
        if (j_to_device) {
            for (size_t c = t; c < nchunks && e == hipSuccess; c += kLanes, ++k) {
                size_t off, len;
                span(c, off, len);
                const int q = (int)(k & 1);
                if (k >= 2) e = hipEventSynchronize(ev[t][q]);  // the DMA that last read this buffer
                if (e != hipSuccess) break;
                std::memcpy(pin[t][q], host + off, len);
                e = hipMemcpyAsync(dev + off, pin[t][q], len, hipMemcpyHostToDevice, stream[t]);
                if (e == hipSuccess) e = hipEventRecord(ev[t][q], stream[t]);
            }
        } else {
            size_t off, len;
            if ((size_t)t < nchunks) {
                span(t, off, len);
                e = hipMemcpyAsync(pin[t][0], dev + off, len, hipMemcpyDeviceToHost, stream[t]);
                if (e == hipSuccess) e = hipEventRecord(ev[t][0], stream[t]);
            }
            for (size_t c = t; c < nchunks && e == hipSuccess; c += kLanes, ++k) {
                const int q = (int)(k & 1);
                if (c + kLanes < nchunks) {  // the next chunk of this lane into the other buffer
                    size_t noff, nlen;
                    span(c + kLanes, noff, nlen);
                    e = hipMemcpyAsync(pin[t][q ^ 1], dev + noff, nlen, hipMemcpyDeviceToHost, stream[t]);
                    if (e == hipSuccess) e = hipEventRecord(ev[t][q ^ 1], stream[t]);
                    if (e != hipSuccess) break;
                }
                span(c, off, len);
                e = hipEventSynchronize(ev[t][q]);
                if (e != hipSuccess) break;
                std::memcpy(host + off, pin[t][q], len);
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream[t]);
        return e;
    }
    // to_device: dev <- host, else host <- dev.  Synchronous: returns when every byte has arrived.
    hipError_t run(void *dev, void *host, size_t bytes, bool to_device)
    {
        std::unique_lock<std::mutex> lk(mu);
        j_dev = dev; j_host = host; j_bytes = bytes; j_to_device = to_device;
        pending = kLanes;
        ++job_id;
        cv_job.notify_all();
        cv_done.wait(lk, [&]() { return pending == 0; });
        for (int t = 0; t < kLanes; ++t)
            if (j_err[t] != hipSuccess) return j_err[t];
        return hipSuccess;
    }
};

struct mi32_context {
    int device = 0;
    HostCopier copier;
    EventProfiler *prof = nullptr;
    hipEvent_t switch_event = nullptr;
    hipStream_t aux_stream = nullptr;   // look-ahead half of the rank-bw updates (lowest priority)
    hipStream_t split_stream = nullptr; // second half of a split batch (same priority as the main stream)
    hipEvent_t la_events[8] = {};
    int cu_count = 0;
    bool lookahead = true;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    int algo = MI32_ALGO_AUTO;
    bool pivoting = true;  // false: the reference's no-pivot variant (sweep kernels, the diagonal entry is the pivot)
    int panel_w = 0;
    int block_w = 0;
    // staging for the host-pointer entry points
    float *d_in = nullptr, *d_out = nullptr;
    int *d_status = nullptr;
    size_t io_floats = 0, status_ints = 0;  // io_floats: capacity of d_in / d_out in 4-byte units
    // status words of device-resident calls that pass no status buffer: the kernels always have one to flag
    // a bad pivot or a lost panel partner in (a given-up matrix is then skipped and comes out as NaN)
    int *d_istatus = nullptr;
    size_t istatus_ints = 0;
    std::mutex mu;
};

static thread_local std::string g_last_error;
// Host-pointer entry points (fp32 and fp64 alike) share the default context's staging buffers and the two
// timing words below: ONE mutex serialises them all.
static std::mutex g_host_call_mu;
static double g_last_total = 0.0, g_last_compute = 0.0;  // guarded by g_host_call_mu

static int fail(hipError_t e, const char *what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return MI32_RUNTIME_ERROR;
}
#define MI32_HIP(call)                                  \
    do {                                                \
        hipError_t e__ = (call);                        \
        if (e__ != hipSuccess) return fail(e__, #call); \
    } while (0)

static int env_int(const char *name, int dflt)
{
    const char *s = std::getenv(name);
    return (s && *s) ? std::atoi(s) : dflt;
}

// The look-ahead half of a single large matrix (blocked_invert): how many CUs it runs on and whether it shares them.
// Measured on MI355X (256 CUs), ms per inversion, "free CUs / LDS KB per look-ahead workgroup":
//   N  4096:  32/84 8.81   32/156 8.66   64/156 8.53   96/156 8.51   128/156 8.49
//   N  6144:  16/84 18.96  32/84 17.88   32/156 17.78  64/156 17.01  128/156 17.33
//   N  8192:  16/84 30.6   32/156 30.3   64/156 29.5
//   N 10240:  16/84 41.7   32/84 40.6    32/156 42.1   64/156 40.9   128/156 50.2
//   N 12288:  16/84 58.5   32/84 57.9    32/156 59.0   64/156 60.8   128/156 84.7
//   N 16384:  16/84 107.5  32/84 111.4   32/156 125.4  64/156 124.3  (8/84 107.0, 5/84 137.7: the shared panels starve)
// Round 3 (reference-order strips, bw = 256 everywhere), same notation:
//   N  8192:  8/84 34.8   16/84 34.9   32/84 33.2   64/84 33.9   32/156 35.6   64/156 33.9
//   N 12288:  8/84 70.5   16/84 70.4   32/84 66.9   64/84 67.9   32/156 72.8   64/156 68.4
//   N 16384:  8/84 119.1  16/84 121.8  32/84 125.4  64/84 141.9  32/156 134.1
//   (two or three look-ahead workgroups per CU, 76 / 50 KB each: 8192 37.7, 12288 77.8, 16384 143-145: the shared panels starve)
// Up to ~7168 rows the half is short against the panel phase: it gets few CUs, all to itself, and the panel chain
// keeps the rest undisturbed; above, every block waits for the half: it gets all but 32 / 16 CUs and shares them.
// MI32_RESERVED_CUS / MI32_LA_EXCLUSIVE override.
static void lookahead_geometry(int cus, int n, int *workgroups, bool *exclusive)
{
    int reserve;
    bool excl;
    if (n < 5120) { reserve = cus / 2; excl = true; }
    else if (n <= 7168) { reserve = cus / 4; excl = true; }
    else if (n <= 14336) { reserve = cus / 8; excl = false; }
    else { reserve = cus / 32; excl = false; }
    const int r_env = env_int("MI32_RESERVED_CUS", 0);
    if (r_env > 0) reserve = r_env;
    if (reserve < 1) reserve = 1;
    if (reserve > cus - 1) reserve = cus - 1;
    const int x_env = env_int("MI32_LA_EXCLUSIVE", -1);
    if (x_env >= 0) excl = x_env != 0;
    *workgroups = cus - reserve;
    *exclusive = excl;
}

static int resolve_algo(const mi32_context *h, int n)
{
    int algo = h ? h->algo : MI32_ALGO_AUTO;
    if (algo == MI32_ALGO_AUTO) algo = env_int("MI32_ALGO", MI32_ALGO_AUTO);
    // the no-pivot variant (fp32): blocked from 512 rows on (the W x W diagonal block is its whole "panel")
    const int cross = (h && !h->pivoting) ? 512 : 32;
    if (algo != MI32_ALGO_SWEEP && algo != MI32_ALGO_BLOCKED) algo = (n >= cross) ? MI32_ALGO_BLOCKED : MI32_ALGO_SWEEP;  // measured cross-over on MI355X
    if (algo == MI32_ALGO_BLOCKED && !blocked_supported(n)) algo = MI32_ALGO_SWEEP;  // panel would not fit in registers
    return algo;
}
static BlockedPlan plan_blocked(const mi32_context *h, int n, int batch)
{
    int w = h && h->panel_w ? h->panel_w : env_int("MI32_PANEL_W", 0);
    int bw = h && h->block_w ? h->block_w : env_int("MI32_BLOCK_W", 0);
    if (bw == 0) {
        // A single matrix is bound by the pivot chain and bw = 256 gives the rank-bw update its best
        // arithmetic intensity.  A batch that fills the GPU is bound by the HBM traffic of the in-block
        // updates (np x bw re-written per sub-panel): bw = 128 halves it (measured 64 x 2048^2:
        // 23.0 ms vs 24.7 ms; single 4096^2: 11.4 ms vs 11.2 ms).
        const double elems = (double)batch * (double)n * (double)n;
        bw = (batch >= 8 && elems >= 64.0 * 1024.0 * 1024.0) ? 128 : 256;
        // (Rounds 1-2 ran N > 14336 with bw = 512 for the rank-bw update's sake: 120 instead of 114 TFLOP/s.  With the
        // pivot-row strips of round 3 an in-block update tile costs more and there are twice as many per sub-panel at
        // 512: 16384^2 122 ms at 256, 126 at 384, 140 at 512; 12288^2 65.5 vs 75.9; 8192^2 33.9 vs 38.0.)
    }
    return make_blocked_plan(n, w, bw, batch);
}
static size_t ws_bytes_for(const mi32_context *h, int n, int batch, int algo)
{
    size_t a;
    if (algo == MI32_ALGO_SWEEP) a = sweep_workspace_bytes(make_sweep_plan(n), batch, sizeof(float));
    else {
        const BlockedPlan p = plan_blocked(h, n, batch);
        // a batch that may be split in two halves (mi32_inv_device) carves one workspace per half
        a = blocked_workspace_bytes(p, (batch + 1) / 2) + blocked_workspace_bytes(p, batch - (batch + 1) / 2);
        const size_t whole = blocked_workspace_bytes(p, batch);
        if (whole > a) a = whole;
    }
    size_t r = residual_workspace_bytes(n, batch);
    return a > r ? a : r;
}

static int sync_all_streams(mi32_context *h)
{
    MI32_HIP(hipStreamSynchronize(h->stream));
    if (h->aux_stream) MI32_HIP(hipStreamSynchronize(h->aux_stream));
    if (h->split_stream) MI32_HIP(hipStreamSynchronize(h->split_stream));
    return MI32_OK;
}

// the status buffer a device-resident call runs with: the caller's, or the context's own
static int status_buffer(mi32_context *h, int *d_status, int batch, int **out)
{
    *out = d_status;
    if (d_status) return MI32_OK;
    if ((size_t)batch > h->istatus_ints) {
        if (h->d_istatus) {
            int rc = sync_all_streams(h);
            if (rc != MI32_OK) return rc;
            MI32_HIP(hipFree(h->d_istatus));
            h->d_istatus = nullptr;
            h->istatus_ints = 0;
        }
        MI32_HIP(hipMalloc((void **)&h->d_istatus, (size_t)batch * sizeof(int)));
        h->istatus_ints = (size_t)batch;
    }
    *out = h->d_istatus;
    return MI32_OK;
}

static int ensure_ws(mi32_context *h, size_t bytes)
{
    if (bytes <= h->ws_bytes) return MI32_OK;
    if (h->ws) {
        int rc = sync_all_streams(h);
        if (rc != MI32_OK) return rc;
        MI32_HIP(hipFree(h->ws));
        h->ws = nullptr;
        h->ws_bytes = 0;
    }
    MI32_HIP(hipMalloc(&h->ws, bytes));
    h->ws_bytes = bytes;
    return MI32_OK;
}

extern "C" {

int mi32_version(void) { return 120; }
const char *mi32_last_error(void) { return g_last_error.c_str(); }

int mi32_create(mi32_handle_t *out, int device)
{
    if (!out) return MI32_BAD_SHAPE;
    *out = nullptr;
    int count = 0;
    MI32_HIP(hipGetDeviceCount(&count));
    if (count <= 0) {
        g_last_error = "no HIP device visible";
        return MI32_RUNTIME_ERROR;
    }
    if (device < 0) MI32_HIP(hipGetDevice(&device));
    if (device >= count) {
        g_last_error = "device ordinal out of range";
        return MI32_RUNTIME_ERROR;
    }
    mi32_context *h = new (std::nothrow) mi32_context();
    if (!h) return MI32_RUNTIME_ERROR;
    h->device = device;
    MI32_HIP(hipSetDevice(device));
    MI32_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    if (env_int("MI32_LOOKAHEAD", 1)) {
        // Look-ahead (see blocked_invert): the half of each rank-bw update that is not on the critical
        // path runs on a second stream as a persistent kernel with one workgroup per CU on all but
        // MI32_RESERVED_CUS compute units, which stay free for the panel / in-block kernels.
        // (Tried and rejected on MI355X: a plain second stream -- its workgroups fill every CU's register
        // file and the critical-path kernels queue behind them; hipExtStreamCreateWithCUMask -- it
        // serialises the two queues, 17 ms instead of 11.5.)
        hipDeviceProp_t prop;
        MI32_HIP(hipGetDeviceProperties(&prop, device));
        h->cu_count = prop.multiProcessorCount;
        int prio_low = 0, prio_high = 0;
        MI32_HIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        MI32_HIP(hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, prio_low));
        for (auto &ev : h->la_events) MI32_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        MI32_HIP(hipStreamCreateWithFlags(&h->split_stream, hipStreamNonBlocking));
    }
    *out = h;
    return MI32_OK;
}

int mi32_destroy(mi32_handle_t h)
{
    if (!h) return MI32_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->ws) (void)hipFree(h->ws);
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->d_out) (void)hipFree(h->d_out);
    if (h->d_status) (void)hipFree(h->d_status);
    if (h->d_istatus) (void)hipFree(h->d_istatus);
    h->copier.destroy();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->switch_event) (void)hipEventDestroy(h->switch_event);
    if (h->aux_stream) {
        (void)hipStreamSynchronize(h->aux_stream);
        (void)hipStreamDestroy(h->aux_stream);
    }
    if (h->split_stream) {
        (void)hipStreamSynchronize(h->split_stream);
        (void)hipStreamDestroy(h->split_stream);
    }
    for (auto ev : h->la_events)
        if (ev) (void)hipEventDestroy(ev);
    delete h->prof;
    delete h;
    return MI32_OK;
}

int mi32_set_stream(mi32_handle_t h, void *hip_stream)
{
    if (!h) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t ns = (hipStream_t)hip_stream;  // NULL is a stream too: HIP's default stream
    if (ns != h->stream) {
        // the workspace is shared by every call on this context: work enqueued on the new stream
        // must wait for what is still running on the old one
        MI32_HIP(hipSetDevice(h->device));
        if (!h->switch_event) MI32_HIP(hipEventCreateWithFlags(&h->switch_event, hipEventDisableTiming));
        MI32_HIP(hipEventRecord(h->switch_event, h->stream));
        MI32_HIP(hipStreamWaitEvent(ns, h->switch_event, 0));
        h->stream = ns;
    }
    return MI32_OK;
}

int mi32_set_algo(mi32_handle_t h, int algo)
{
    if (!h || algo < MI32_ALGO_AUTO || algo > MI32_ALGO_BLOCKED) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    h->algo = algo;
    return MI32_OK;
}

int mi32_set_pivoting(mi32_handle_t h, int enable)
{
    if (!h) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    h->pivoting = enable != 0;
    return MI32_OK;
}

int mi32_set_lookahead(mi32_handle_t h, int enable)
{
    if (!h) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    h->lookahead = enable != 0;
    return MI32_OK;
}

int mi32_set_blocking(mi32_handle_t h, int panel_width, int block_width)
{
    if (!h || panel_width < 0 || block_width < 0) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    h->panel_w = panel_width;
    h->block_w = block_width;
    return MI32_OK;
}

size_t mi32_workspace_bytes(int n, int batch, int algo)
{
    if (n <= 0 || batch <= 0) return 0;
    mi32_context tmp;
    tmp.algo = algo;
    return ws_bytes_for(&tmp, n, batch, resolve_algo(&tmp, n));
}

int mi32_resolve_algo(mi32_handle_t h, int n, int /*batch*/) { return resolve_algo(h, n); }

int mi32_resolve_blocking(mi32_handle_t h, int n, int batch, int *panel_width, int *block_width)
{
    if (n <= 0 || batch <= 0) return MI32_BAD_SHAPE;
    const BlockedPlan p = plan_blocked(h, n, batch);
    if (panel_width) *panel_width = p.w;
    if (block_width) *block_width = p.bw;
    return MI32_OK;
}

int mi32_resolve_panel_widths(mi32_handle_t h, int n, int batch, int *widths, int capacity, int *nblocks)
{
    if (n <= 0 || batch <= 0 || capacity < 0 || (capacity > 0 && !widths)) return MI32_BAD_SHAPE;
    const BlockedPlan p = plan_blocked(h, n, batch);
    if (nblocks) *nblocks = p.nblk;
    for (int b = 0; b < p.nblk && b < capacity; ++b) widths[b] = p.wblk[b];
    return MI32_OK;
}

const char *mi32_dominant_kernel(int algo)
{
    return algo == MI32_ALGO_SWEEP ? "gj_sweep_step_kernel" : "gj_rank_bw2_kernel";
}

int mi32_reserve(mi32_handle_t h, int n, int batch)
{
    if (!h || n <= 0 || batch <= 0) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    MI32_HIP(hipSetDevice(h->device));
    return ensure_ws(h, ws_bytes_for(h, n, batch, resolve_algo(h, n)));
}

// A GPU-filling batch of the blocked path is run as two halves on the context's two streams: the MFMA-bound
// rank-bw launches of one half overlap the latency / HBM-bound sub-panel launches of the other (64 x 2048^2:
// 18.5 -> 17.4 ms; three or four parts lose; round 3: 8 x 4096^2 23.6 -> 22.1 ms, 4 x 4096^2 15.7 -> 15.0: from four
// matrices on).  Both halves use the blocking of the whole batch, so a matrix's
// result does not depend on the split; mi32_set_lookahead(h, 0) turns the second stream off altogether.
static bool split_batch(const mi32_context *h, int algo, int n, int batch)
{
    return algo == MI32_ALGO_BLOCKED && h->lookahead && h->split_stream != nullptr && batch >= env_int("MI32_BATCH_SPLIT_MIN", 4) &&
           (double)batch * n * n >= 64.0 * 1024.0 * 1024.0 && env_int("MI32_BATCH_SPLIT", 1) != 0;
}

int mi32_inv_device(mi32_handle_t h, const float *d_a, int n, int batch, float *d_inv, int *d_status)
{
    if (!h || !d_a || !d_inv || n <= 0 || batch <= 0 || d_a == d_inv) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    MI32_HIP(hipSetDevice(h->device));
    const int algo = resolve_algo(h, n);
    int rc = ensure_ws(h, ws_bytes_for(h, n, batch, algo));
    if (rc != MI32_OK) return rc;
    rc = status_buffer(h, d_status, batch, &d_status);
    if (rc != MI32_OK) return rc;
    hipError_t e;
    if (algo == MI32_ALGO_SWEEP)
        e = sweep_invert(make_sweep_plan(n), d_a, d_inv, batch, d_status, h->ws, h->stream, h->prof, h->pivoting);
    else {
        BlockedExec ex;
        ex.stream = h->stream;
        ex.aux = h->lookahead ? h->aux_stream : nullptr;
        ex.events = h->la_events;
        ex.n_events = h->aux_stream ? 8 : 0;
        lookahead_geometry(h->cu_count, n, &ex.aux_workgroups, &ex.aux_exclusive);
        ex.prof = h->prof;
        ex.pivoting = h->pivoting;
        const BlockedPlan p = plan_blocked(h, n, batch);
        // (plans with panels shared by several workgroups are not split: those workgroups need whole CUs at the same
        // time, which the other half's rank-bw grid would keep from them for the length of its launch)
        if (!split_batch(h, algo, n, batch) || p.multi_panel) {
            e = blocked_invert(p, d_a, d_inv, batch, d_status, h->ws, ex);
        } else {
            const int b0 = (batch + 1) / 2, b1 = batch - b0;
            const size_t mat = (size_t)n * n;
            char *ws1 = (char *)h->ws + blocked_workspace_bytes(p, b0);  // ws_bytes_for reserved both parts
            // the second stream joins here and is joined again at the end (events 0 and 1 are free: the
            // look-ahead, their other user, only runs for single matrices)
            MI32_HIP(hipEventRecord(h->la_events[0], h->stream));
            MI32_HIP(hipStreamWaitEvent(h->split_stream, h->la_events[0], 0));
            BlockedExec ex0 = ex, ex1 = ex;
            ex0.aux = ex1.aux = nullptr;
            ex1.stream = h->split_stream;
            e = blocked_invert(p, d_a, d_inv, b0, d_status, h->ws, ex0);
            if (e == hipSuccess)
                e = blocked_invert(p, d_a + (size_t)b0 * mat, d_inv + (size_t)b0 * mat, b1,
                                   d_status + b0, ws1, ex1);
            MI32_HIP(hipEventRecord(h->la_events[1], h->split_stream));
            MI32_HIP(hipStreamWaitEvent(h->stream, h->la_events[1], 0));
        }
    }
    if (e != hipSuccess) return fail(e, "kernel launch");
    return MI32_OK;
}

// fp64: blocked (windowed steps + rank-bw updates on the fp64 matrix cores) from N = 256 on, measured cross-over;
// the no-pivot variant and MI32_ALGO_SWEEP keep the unblocked sweep
static int resolve_algo_f64(const mi32_context *h, int n)
{
    if (h && !h->pivoting) return MI32_ALGO_SWEEP;
    int algo = h ? h->algo : MI32_ALGO_AUTO;
    if (algo == MI32_ALGO_AUTO) algo = env_int("MI32_ALGO", MI32_ALGO_AUTO);
    if (algo != MI32_ALGO_SWEEP && algo != MI32_ALGO_BLOCKED) algo = (n >= 256) ? MI32_ALGO_BLOCKED : MI32_ALGO_SWEEP;
    return algo;
}
static Blocked64Plan plan_blocked64(const mi32_context *h, int n)
{
    return make_blocked64_plan(n, h && h->block_w ? h->block_w : env_int("MI32_BLOCK_W64", 0));
}

int mi32_resolve_blocking_f64(mi32_handle_t h, int n, int *block_width)
{
    if (n <= 0 || !block_width) return MI32_BAD_SHAPE;
    *block_width = resolve_algo_f64(h, n) == MI32_ALGO_BLOCKED ? plan_blocked64(h, n).bw : 0;
    return MI32_OK;
}

int mi32_inv_device_f64(mi32_handle_t h, const double *d_a, int n, int batch, double *d_inv, int *d_status)
{
    if (!h || !d_a || !d_inv || n <= 0 || batch <= 0 || d_a == d_inv) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    MI32_HIP(hipSetDevice(h->device));
    if (resolve_algo_f64(h, n) == MI32_ALGO_BLOCKED) {
        const Blocked64Plan bp = plan_blocked64(h, n);
        int rc = ensure_ws(h, blocked64_workspace_bytes(bp, batch));
        if (rc != MI32_OK) return rc;
        rc = status_buffer(h, d_status, batch, &d_status);
        if (rc != MI32_OK) return rc;
        hipError_t eb = blocked64_invert(bp, d_a, d_inv, batch, d_status, h->ws, h->stream, h->prof);
        if (eb != hipSuccess) return fail(eb, "kernel launch");
        return MI32_OK;
    }
    const SweepPlan p = make_sweep_plan(n);
    int rc = ensure_ws(h, sweep_workspace_bytes(p, batch, sizeof(double)));
    if (rc != MI32_OK) return rc;
    rc = status_buffer(h, d_status, batch, &d_status);
    if (rc != MI32_OK) return rc;
    hipError_t e = sweep_invert_f64(p, d_a, d_inv, batch, d_status, h->ws, h->stream, h->prof, h->pivoting);
    if (e != hipSuccess) return fail(e, "kernel launch");
    return MI32_OK;
}

int mi32_set_profiling(mi32_handle_t h, int enable)
{
    if (!h) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    MI32_HIP(hipSetDevice(h->device));
    if (enable && !h->prof) h->prof = new (std::nothrow) EventProfiler();
    if (!enable && h->prof) {
        MI32_HIP(hipStreamSynchronize(h->stream));
        delete h->prof;
        h->prof = nullptr;
    }
    return MI32_OK;
}

int mi32_get_profile(mi32_handle_t h, double *ms_per_class, long long *launches_per_class, int nclasses)
{
    if (!h || !ms_per_class || !launches_per_class || nclasses < KC_COUNT) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    for (int i = 0; i < nclasses; ++i) { ms_per_class[i] = 0.0; launches_per_class[i] = 0; }
    if (!h->prof) return MI32_OK;
    MI32_HIP(hipSetDevice(h->device));
    h->prof->collect(ms_per_class, launches_per_class);
    return MI32_OK;
}

int mi32_residual_device(mi32_handle_t h, const float *d_a, const float *d_x, int n, int batch, double *d_out)
{
    if (!h || !d_a || !d_x || !d_out || n <= 0 || batch <= 0) return MI32_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(h->mu);
    MI32_HIP(hipSetDevice(h->device));
    int rc = ensure_ws(h, ws_bytes_for(h, n, batch, resolve_algo(h, n)));
    if (rc != MI32_OK) return rc;
    hipError_t e = residual_launch(d_a, d_x, n, batch, d_out, h->ws, h->stream);
    if (e != hipSuccess) return fail(e, "residual launch");
    return MI32_OK;
}

// ---- host-pointer entry points on the default context ---------------------------
static mi32_context *g_default = nullptr;
static std::mutex g_default_mu;

static int default_context(mi32_context **out)
{
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        mi32_handle_t h = nullptr;
        int rc = mi32_create(&h, env_int("MI32_DEVICE", 0));
        if (rc != MI32_OK) return rc;
        g_default = h;
    }
    *out = g_default;
    return MI32_OK;
}

static int ensure_io(mi32_context *h, size_t floats, size_t ints)
{
    if (floats > h->io_floats) {
        if (h->d_in) MI32_HIP(hipFree(h->d_in));
        if (h->d_out) MI32_HIP(hipFree(h->d_out));
        h->d_in = h->d_out = nullptr;
        h->io_floats = 0;
        MI32_HIP(hipMalloc((void **)&h->d_in, floats * sizeof(float)));
        MI32_HIP(hipMalloc((void **)&h->d_out, floats * sizeof(float)));
        h->io_floats = floats;
    }
    if (ints > h->status_ints) {
        if (h->d_status) MI32_HIP(hipFree(h->d_status));
        h->d_status = nullptr;
        h->status_ints = 0;
        MI32_HIP(hipMalloc((void **)&h->d_status, ints * sizeof(int)));
        h->status_ints = ints;
    }
    return MI32_OK;
}

// dev <- host (to_device) or host <- dev; synchronous.  Large transfers go through the pinned ring of HostCopier
// (MI32_HOST_COPY=0 keeps the runtime's pageable path), small ones through one hipMemcpyAsync.
static int host_copy(mi32_context *h, void *dev, void *host, size_t bytes, bool to_device)
{
    if (bytes >= HostCopier::kMinBytes && env_int("MI32_HOST_COPY", 1) != 0) {
        MI32_HIP(h->copier.init(h->device));
        MI32_HIP(hipStreamSynchronize(h->stream));  // the lanes' streams are not ordered with the context's stream
        MI32_HIP(h->copier.run(dev, host, bytes, to_device));
        return MI32_OK;
    }
    if (to_device) MI32_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, h->stream));
    else MI32_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream));
    MI32_HIP(hipStreamSynchronize(h->stream));
    return MI32_OK;
}

// The reference's ten timing slots (FP32_bench.cpp:256-443) from the profiler's per-class milliseconds and the host
// clock stamps of a host-pointer call (tq0: before the context, t0: after it, t1: H2D done, t2: compute done, t3: D2H done).
static void fill_times10(double *times10, const double *ms, std::chrono::steady_clock::time_point tq0,
                         std::chrono::steady_clock::time_point t0, std::chrono::steady_clock::time_point t1,
                         std::chrono::steady_clock::time_point t2, std::chrono::steady_clock::time_point t3)
{
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };
    times10[0] = sec(tq0, t0);
    times10[1] = sec(t0, t1);
    times10[2] = 0.0;  // one ahead-of-time compiled code object: nothing is built at run time
    times10[3] = ms[KC_INIT] * 1e-3;
    // the panel kernel IS maxPivot + finalMaxPivot + pivotElements + fixRow (+ fixColumn on the panel's own columns); the
    // fused step launches of the sweep path are accounted to the column slot, where the reference spends its time
    times10[4] = ms[KC_PANEL] * 1e-3;
    times10[5] = 0.0;  // fixRow has no launch of its own
    times10[6] = (ms[KC_SWEEP_STEP] + ms[KC_UPDATE_IN] + ms[KC_UPDATE_OUT] + ms[KC_TRANSPOSE]) * 1e-3;
    times10[7] = sec(t1, t2);
    times10[8] = ms[KC_FINISH] * 1e-3 + sec(t2, t3);
    times10[9] = sec(tq0, t3);
}

// The host-pointer path.  times10 (may be NULL): the reference's timing vector, FP32_bench.cpp:256-443 /
// res_struct.h:4-6 -- [0] queue/context, [1] buffers (+ the H2D copy the reference's CL_MEM_COPY_HOST_PTR does),
// [2] program build, [3] makeAugmented, [4] pivot, [5] row, [6] column, [7] compute, [8] getInverted (+ D2H),
// [9] total; seconds.  The per-phase slots come from HIP events on the launch stream (mi32_set_profiling).
// Pre-faulting of a large host buffer on several threads, WITHOUT writing to it: madvise(MADV_POPULATE_WRITE) makes
// the kernel install writable pages (zero pages for fresh memory, the present contents otherwise) -- the buffer's
// bytes, and any C++ object that lives or will live there, are never touched by us.  The kernel hands out pages one
// fault at a time: 64 MiB cost ~12 ms on one thread of the MI355X host, ~2 ms on eight.  Used (a) on the result
// vector's reserved storage before it is value-initialised and (b) on the caller's output buffer while the device
// works -- which therefore keeps its contents until the copy back (mat_inv_32_c.h: "written only on MI32_OK /
// MI32_SINGULAR").  Where the kernel does not know the advice the pages are faulted by the copy itself, as before.
static void parallel_populate(void *p, size_t bytes, bool may_rewrite = false)
{
    const size_t kPage = 4096, kMin = (size_t)8 << 20;
    if (bytes < kMin) return;
    const uintptr_t lo = ((uintptr_t)p + kPage - 1) & ~(uintptr_t)(kPage - 1);
    const uintptr_t hi = ((uintptr_t)p + bytes) & ~(uintptr_t)(kPage - 1);
    if (hi <= lo) return;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt > 8 ? 8 : (nt < 1 ? 1 : nt);
    const size_t span = hi - lo;
    const size_t chunk = ((span / nt) + kPage - 1) & ~(kPage - 1);
    // may_rewrite (the caller's output buffer, ours to write for the duration of the call): where the kernel does not
    // know the advice (Linux < 5.14) every page's first byte is read and written back unchanged instead
    auto populate = [may_rewrite](uintptr_t a, size_t len) {
        int rc = -1;
#ifdef MADV_POPULATE_WRITE
        rc = madvise(reinterpret_cast<void *>(a), len, MADV_POPULATE_WRITE);
#endif
        if (rc != 0 && may_rewrite) {
            for (size_t off = 0; off < len; off += 4096) {
                volatile char *q = reinterpret_cast<volatile char *>(a + off);
                const char v = *q;
                *q = v;
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; ++i) {
        const size_t off = (size_t)i * chunk;
        if (off >= span) break;
        const size_t len = (off + chunk <= span) ? chunk : span - off;
        try { th.emplace_back(populate, lo + off, len); } catch (...) { populate(lo + off, len); }
    }
    populate(lo, chunk < span ? chunk : span);
    for (auto &t : th) t.join();
}

// Error returns of the host-pointer paths leave the context as they found it.
struct ProfilingGuard {
    mi32_context *h = nullptr;
    ~ProfilingGuard() { if (h) (void)mi32_set_profiling(h, 0); }
};

// `late_out`: where the result goes is only asked for once the kernels are queued -- the std::vector entry points
// allocate and first-touch their 4 N^2 result bytes (64 MiB of page faults at N = 4096, ~8 ms) while the device works
typedef void *(*LateOut)(void *ctx);
// One host-pointer inversion on context h; the caller holds the lock that guards h's staging buffers.
// total_s / compute_s: the reference's two numbers ("Tempo Totale Impiegato" / "Tempo Computazione", mat_inv_32.cpp:385-386).
static int host_invert_32_on(mi32_context *h, std::chrono::steady_clock::time_point tq0, const float *a, int n, int batch,
                             float *inv, int *status, double *times10, LateOut late_out, void *late_ctx, double *total_s,
                             double *compute_s)
{
    const auto t0 = std::chrono::steady_clock::now();
    MI32_HIP(hipSetDevice(h->device));
    const size_t floats = (size_t)batch * n * n;
    int rc = ensure_io(h, floats, (size_t)batch);
    if (rc != MI32_OK) return rc;
    ProfilingGuard prof_guard;  // profiling is switched off again on every way out
    if (times10) {
        rc = mi32_reserve(h, n, batch);  // workspace allocation belongs to the "buffers" slot
        if (rc != MI32_OK) return rc;
        rc = mi32_set_profiling(h, 1);
        if (rc != MI32_OK) return rc;
        prof_guard.h = h;
        double ms[KC_COUNT]; long long cnt[KC_COUNT];
        (void)mi32_get_profile(h, ms, cnt, KC_COUNT);  // drop what an earlier call left
    }
    rc = host_copy(h, h->d_in, const_cast<float *>(a), floats * sizeof(float), true);
    if (rc != MI32_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    rc = mi32_inv_device(h, h->d_in, n, batch, h->d_out, h->d_status);
    if (rc != MI32_OK) return rc;
    if (late_out) {
        inv = static_cast<float *>(late_out(late_ctx));
        if (!inv) { (void)hipStreamSynchronize(h->stream); return MI32_RUNTIME_ERROR; }
    } else {
        parallel_populate(inv, floats * sizeof(float), true);  // the device is busy for the next milliseconds
    }
    MI32_HIP(hipStreamSynchronize(h->stream));
    const auto t2 = std::chrono::steady_clock::now();
    std::vector<int> st((size_t)batch);
    MI32_HIP(hipMemcpyAsync(st.data(), h->d_status, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    rc = host_copy(h, h->d_out, inv, floats * sizeof(float), false);
    if (rc != MI32_OK) return rc;
    MI32_HIP(hipStreamSynchronize(h->stream));
    const auto t3 = std::chrono::steady_clock::now();
    if (total_s) *total_s = std::chrono::duration<double>(t3 - t0).count();
    if (compute_s) *compute_s = std::chrono::duration<double>(t2 - t1).count();
    if (times10) {
        double ms[KC_COUNT]; long long cnt[KC_COUNT];
        rc = mi32_get_profile(h, ms, cnt, KC_COUNT);
        if (rc != MI32_OK) return rc;
        fill_times10(times10, ms, tq0, t0, t1, t2, t3);
    }
    int worst = MI32_OK;
    for (int b = 0; b < batch; ++b) {
        if (status) status[b] = st[(size_t)b];
        if (st[(size_t)b] > worst) worst = st[(size_t)b];
    }
    if (worst == MI32_RUNTIME_ERROR)  // the only status-borne runtime error (mi32_blocked.hip, shared panels)
        g_last_error = "a workgroup of a shared panel timed out waiting for its partners (the device was not ours "
                       "alone); the affected inverse is NaN-filled -- retry, or set MI32_MULTI_PANEL=0";
    return worst;
}

static void print_reference_timing_lines()
{
    if (env_int("MI32_VERBOSE", 0)) {
        // the reference's two stdout lines (mat_inv_32.cpp:385-386)
        std::printf("Tempo Totale Impiegato: %g seconds\nTempo Computazione: %g seconds\n", g_last_total,
                    g_last_compute);
        std::fflush(stdout);
    }
}

static int host_invert_32(const float *a, int n, int batch, float *inv, int *status, double *times10,
                          LateOut late_out = nullptr, void *late_ctx = nullptr)
{
    if (!a || (!inv && !late_out) || n <= 0 || batch <= 0) return MI32_BAD_SHAPE;
    const auto tq0 = std::chrono::steady_clock::now();
    mi32_context *h = nullptr;
    int rc = default_context(&h);  // the reference's platform / device / context / queue bring-up (cached here)
    if (rc != MI32_OK) return rc;
    std::lock_guard<std::mutex> lk(g_host_call_mu);  // one host-pointer call at a time: the staging buffers are shared
    rc = host_invert_32_on(h, tq0, a, n, batch, inv, status, times10, late_out, late_ctx, &g_last_total, &g_last_compute);
    if (rc == MI32_OK || rc == MI32_SINGULAR || rc == MI32_RUNTIME_ERROR) print_reference_timing_lines();
    return rc;
}

// ---- the batch over several GPUs (what replaces the reference's platforms[0] / devices[0],
//      mat_inv_32.cpp:239-244) -------------------------------------------------------------------------
// One context and one host thread per GPU; GPU g owns the matrices [g * ceil(B / G), min(B, (g + 1) * ceil(B / G)))
// and copies ITS OWN shard host -> device, inverts it and copies it back: no data-path exchange between the GPUs,
// the worst status word is the return value.  The contexts are created on first use and kept.
// MI32_MULTI_OVERSUBSCRIBE=1 (tests, single-GPU hosts): logical GPU g runs on device g % (visible devices), each with
// a context of its own -- the threading, the ragged shards and the status reduction run exactly as on a real node.
namespace {
struct MultiSlot {
    mi32_context *h = nullptr;
    std::mutex mu;  // guards h's staging buffers, like g_host_call_mu guards the default context's
};
std::mutex g_multi_mu;                 // guards the table
std::vector<MultiSlot *> g_multi;      // logical GPU -> slot (never shrinks; slots are never freed)
}  // namespace

// GPU g of G owns the matrices [g * ceil(B / G), min(B, (g + 1) * ceil(B / G))) (possibly none)
extern "C" int mi32_shard_range(int batch, int ngpus, int g, int *lo, int *hi)
{
    if (batch < 0 || ngpus <= 0 || g < 0 || g >= ngpus || !lo || !hi) return MI32_BAD_SHAPE;
    const int per = (batch + ngpus - 1) / ngpus;
    const long long l = (long long)g * per;
    *lo = l < batch ? (int)l : batch;
    *hi = (l + per < batch) ? (int)(l + per) : batch;
    return MI32_OK;
}

extern "C" int mi32_matrix_inv_32_batched_multi(const float *a, int n, int batch, float *inv, int *status, int ngpus)
{
    if (!a || !inv || n <= 0 || batch <= 0) return MI32_BAD_SHAPE;
    const auto tq0 = std::chrono::steady_clock::now();
    int visible = 0;
    MI32_HIP(hipGetDeviceCount(&visible));
    if (visible <= 0) {
        g_last_error = "no HIP device visible";
        return MI32_RUNTIME_ERROR;
    }
    const bool oversub = env_int("MI32_MULTI_OVERSUBSCRIBE", 0) != 0;
    if (ngpus <= 0) ngpus = visible;
    if (ngpus > visible && !oversub) {
        g_last_error = "mi32_matrix_inv_32_batched_multi: more GPUs asked for than are visible";
        return MI32_BAD_SHAPE;
    }
    if (ngpus > batch) ngpus = batch;  // at least one matrix per GPU
    std::vector<MultiSlot *> slots((size_t)ngpus, nullptr);
    {
        std::lock_guard<std::mutex> lk(g_multi_mu);
        while ((int)g_multi.size() < ngpus) g_multi.push_back(new (std::nothrow) MultiSlot());
        for (int g = 0; g < ngpus; ++g) {
            if (!g_multi[(size_t)g]) return MI32_RUNTIME_ERROR;
            slots[(size_t)g] = g_multi[(size_t)g];
        }
    }
    std::vector<int> rcs((size_t)ngpus, MI32_OK);
    std::vector<std::string> errs((size_t)ngpus);
    std::vector<double> tot((size_t)ngpus, 0.0), cmp((size_t)ngpus, 0.0);
    const size_t mat = (size_t)n * n;
    auto work = [&](int g) {
        int lo = 0, hi = 0;
        (void)mi32_shard_range(batch, ngpus, g, &lo, &hi);
        if (lo >= hi) return;  // ragged tail: this GPU has nothing
        MultiSlot *sl = slots[(size_t)g];
        std::lock_guard<std::mutex> lk(sl->mu);
        int rc = MI32_OK;
        if (!sl->h) {
            mi32_handle_t nh = nullptr;
            rc = mi32_create(&nh, g % visible);
            if (rc == MI32_OK) sl->h = nh;
        }
        if (rc == MI32_OK)
            rc = host_invert_32_on(sl->h, tq0, a + (size_t)lo * mat, n, hi - lo, inv + (size_t)lo * mat,
                                   status ? status + lo : nullptr, nullptr, nullptr, nullptr, &tot[(size_t)g], &cmp[(size_t)g]);
        rcs[(size_t)g] = rc;
        if (rc != MI32_OK) errs[(size_t)g] = g_last_error;  // thread-local: carried to the caller's thread below
    };
    std::vector<std::thread> th;
    for (int g = 1; g < ngpus; ++g) {
        try { th.emplace_back(work, g); } catch (...) { work(g); }
    }
    work(0);
    for (auto &t : th) t.join();
    int worst = MI32_OK;
    double t_tot = 0.0, t_cmp = 0.0;
    for (int g = 0; g < ngpus; ++g) {
        const int rc = rcs[(size_t)g];
        if (rc == MI32_BAD_SHAPE) return MI32_BAD_SHAPE;
        if (rc > worst) { worst = rc; if (!errs[(size_t)g].empty()) g_last_error = errs[(size_t)g]; }
        if (tot[(size_t)g] > t_tot) t_tot = tot[(size_t)g];
        if (cmp[(size_t)g] > t_cmp) t_cmp = cmp[(size_t)g];
    }
    {
        std::lock_guard<std::mutex> lk(g_host_call_mu);
        g_last_total = t_tot;
        g_last_compute = t_cmp;
        print_reference_timing_lines();
    }
    return worst;
}

int mi32_matrix_inv_32_batched(const float *a, int n, int batch, float *inv, int *status)
{
    return host_invert_32(a, n, batch, inv, status, nullptr);
}

int mi32_bench_32(const float *a_rowmajor, size_t a_len, int n, float *inv_rowmajor, double *times10)
{
    if (n <= 0 || !times10) return MI32_BAD_SHAPE;
    if ((int)(a_len / (size_t)n) != n) return MI32_BAD_SHAPE;
    return host_invert_32(a_rowmajor, n, 1, inv_rowmajor, nullptr, times10);
}

int mi32_matrix_inv_32(const float *a_rowmajor, size_t a_len, int n, float *inv_rowmajor)
{
    // the reference's guards, mat_inv_32.cpp:206-215 (integer division included)
    if (n <= 0) return MI32_BAD_SHAPE;
    if ((int)(a_len / (size_t)n) != n) return MI32_BAD_SHAPE;
    return mi32_matrix_inv_32_batched(a_rowmajor, n, 1, inv_rowmajor, nullptr);
}

static int host_invert_64(const double *a_rowmajor, size_t a_len, int n, double *inv_rowmajor, bool pivoting,
                          double *times10, LateOut late_out = nullptr, void *late_ctx = nullptr);

int mi32_matrix_inv_64(const double *a_rowmajor, size_t a_len, int n, double *inv_rowmajor)
{
    return host_invert_64(a_rowmajor, a_len, n, inv_rowmajor, true, nullptr);
}

int mi32_matrix_inversion_no_pivots(const double *a_rowmajor, size_t a_len, int n, double *inv_rowmajor)
{
    return host_invert_64(a_rowmajor, a_len, n, inv_rowmajor, false, nullptr);
}

int mi32_bench_64(const double *a_rowmajor, size_t a_len, int n, double *inv_rowmajor, double *times10, int pivoting)
{
    if (!times10) return MI32_BAD_SHAPE;
    return host_invert_64(a_rowmajor, a_len, n, inv_rowmajor, pivoting != 0, times10);
}

static int host_invert_64(const double *a_rowmajor, size_t a_len, int n, double *inv_rowmajor, bool pivoting,
                          double *times10, LateOut late_out, void *late_ctx)
{
    // the guards of the fp32 library (mat_inv_32.cpp:206-215); matrix_inversion_FP64.cpp has the same two
    if (n <= 0) return MI32_BAD_SHAPE;
    if ((int)(a_len / (size_t)n) != n) return MI32_BAD_SHAPE;
    if (!a_rowmajor || (!inv_rowmajor && !late_out)) return MI32_BAD_SHAPE;
    const auto tq0 = std::chrono::steady_clock::now();
    mi32_context *h = nullptr;
    int rc = default_context(&h);
    if (rc != MI32_OK) return rc;
    std::lock_guard<std::mutex> lk(g_host_call_mu);  // the staging buffers are shared with the fp32 host-pointer calls
    const auto t0 = std::chrono::steady_clock::now();
    MI32_HIP(hipSetDevice(h->device));
    const size_t elems = (size_t)n * n;
    rc = ensure_io(h, 2 * elems, 1);  // doubles: two 4-byte units each
    if (rc != MI32_OK) return rc;
    ProfilingGuard prof_guard;  // profiling is switched off again on every way out
    if (times10) {
        rc = mi32_set_profiling(h, 1);
        if (rc != MI32_OK) return rc;
        prof_guard.h = h;
        double ms0[KC_COUNT]; long long cnt0[KC_COUNT];
        (void)mi32_get_profile(h, ms0, cnt0, KC_COUNT);  // drop what an earlier call left
    }
    double *din = reinterpret_cast<double *>(h->d_in), *dout = reinterpret_cast<double *>(h->d_out);
    rc = host_copy(h, din, const_cast<double *>(a_rowmajor), elems * sizeof(double), true);
    if (rc != MI32_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    {
        const bool saved = h->pivoting;  // the default context is only ever used under g_host_call_mu
        h->pivoting = pivoting;
        rc = mi32_inv_device_f64(h, din, n, 1, dout, h->d_status);
        h->pivoting = saved;
    }
    if (rc != MI32_OK) return rc;
    if (late_out) {
        inv_rowmajor = static_cast<double *>(late_out(late_ctx));
        if (!inv_rowmajor) { (void)hipStreamSynchronize(h->stream); return MI32_RUNTIME_ERROR; }
    } else {
        parallel_populate(inv_rowmajor, elems * sizeof(double), true);
    }
    MI32_HIP(hipStreamSynchronize(h->stream));
    const auto t2 = std::chrono::steady_clock::now();
    int st = MI32_OK;
    MI32_HIP(hipMemcpyAsync(&st, h->d_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    rc = host_copy(h, dout, inv_rowmajor, elems * sizeof(double), false);
    if (rc != MI32_OK) return rc;
    MI32_HIP(hipStreamSynchronize(h->stream));
    const auto t3 = std::chrono::steady_clock::now();
    g_last_total = std::chrono::duration<double>(t3 - t0).count();
    g_last_compute = std::chrono::duration<double>(t2 - t1).count();
    if (times10) {
        double ms[KC_COUNT]; long long cnt[KC_COUNT];
        rc = mi32_get_profile(h, ms, cnt, KC_COUNT);
        if (rc != MI32_OK) return rc;
        fill_times10(times10, ms, tq0, t0, t1, t2, t3);
    }
    return st;
}

// matrix_multiply of the reference (matrix_multiply.cpp:15-212): C = A * B in double on the device, returns
// sqrt(N) - ||C||_F -- the scalar the experiment driver writes per size (main_file.cpp:80-81).
int mi32_matrix_multiply_64(const double *a, const double *b, size_t len, double *errore)
{
    if (!a || !b || !errore || len == 0) return MI32_BAD_SHAPE;
    const int n = (int)std::llround(std::sqrt((double)len));  // the reference takes the order as sqrt(size), :44
    if (n <= 0 || (size_t)n * n != len) return MI32_BAD_SHAPE;
    mi32_context *h = nullptr;
    int rc = default_context(&h);
    if (rc != MI32_OK) return rc;
    std::lock_guard<std::mutex> lk(g_host_call_mu);
    MI32_HIP(hipSetDevice(h->device));
    rc = ensure_io(h, 2 * len, 1);
    if (rc != MI32_OK) return rc;
    {
        std::lock_guard<std::mutex> lk2(h->mu);
        rc = ensure_ws(h, residual_workspace_bytes(n, 1) + 64);
    }
    if (rc != MI32_OK) return rc;
    double *da = reinterpret_cast<double *>(h->d_in), *db = reinterpret_cast<double *>(h->d_out);
    rc = host_copy(h, da, const_cast<double *>(a), len * sizeof(double), true);
    if (rc != MI32_OK) return rc;
    rc = host_copy(h, db, const_cast<double *>(b), len * sizeof(double), true);
    if (rc != MI32_OK) return rc;
    double *d_out = reinterpret_cast<double *>((char *)h->ws + residual_workspace_bytes(n, 1));
    hipError_t e = frobenius_launch_f64(da, db, n, d_out, h->ws, h->stream);
    if (e != hipSuccess) return fail(e, "matrix_multiply launch");
    MI32_HIP(hipMemcpyAsync(errore, d_out, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MI32_HIP(hipStreamSynchronize(h->stream));
    return MI32_OK;
}

int mi32_last_timing(double *total_seconds, double *compute_seconds)
{
    std::lock_guard<std::mutex> lk(g_host_call_mu);
    if (total_seconds) *total_seconds = g_last_total;
    if (compute_seconds) *compute_seconds = g_last_compute;
    return MI32_OK;
}

}  // extern "C"

// ---- the reference's entry point, unchanged signature (Matlab/mat_inv_32.h:4) ----
std::vector<float> matrix_inv_32(std::vector<float> matrix_vector, int matrix_order)
{
    if (matrix_order <= 0) return {};                                        // mat_inv_32.cpp:206-208
    if ((int)(matrix_vector.size() / (size_t)matrix_order) != matrix_order) return {};  // :211-214
    // the result vector comes into being (value-initialised, every page touched) while the device works
    std::vector<float> result;
    struct Ctx { std::vector<float> *v; size_t n; } ctx = {&result, (size_t)matrix_order * matrix_order};
    const int rc = host_invert_32(matrix_vector.data(), matrix_order, 1, nullptr, nullptr, nullptr,
                                  [](void *c) -> void * {
                                      Ctx *x = static_cast<Ctx *>(c);
                                      try {
                                          x->v->reserve(x->n);  // pages first (several threads), then the value-initialisation
                                          parallel_populate(x->v->data(), x->n * sizeof(float));
                                          x->v->resize(x->n);
                                      } catch (...) { return nullptr; }
                                      return x->v->data();
                                  }, &ctx);
    if (rc == MI32_OK) return result;
    // README.md:54 "In case of invalid matrix an empty vector is returned"; the experiment twin
    // does so for a singular input (matrix_inversion_FP32.cpp:814-835).  MI32_SINGULAR_KEEP=1
    // returns the inf/NaN result instead, as the shipped library does.
    if (rc == MI32_SINGULAR && env_int("MI32_SINGULAR_KEEP", 0)) return result;
    if (rc == MI32_RUNTIME_ERROR) std::fprintf(stderr, "matrix_inv_32: %s\n", mi32_last_error());
    return {};
}

// ---- the reference's no-pivot variant, unchanged signature (matrix_inversion/headers.h:11) ----
std::vector<double> matrix_inversion_no_pivots(std::vector<double> matrix_vector, int matrix_order)
{
    if (matrix_order <= 0) return {};
    if ((int)(matrix_vector.size() / (size_t)matrix_order) != matrix_order) return {};
    std::vector<double> result((size_t)matrix_order * matrix_order, 0.0);
    const int rc = mi32_matrix_inversion_no_pivots(matrix_vector.data(), matrix_vector.size(), matrix_order, result.data());
    if (rc == MI32_OK) return result;
    // a zero diagonal entry on the way: {} like the reference (exact-identity check, matrix_inversion_no_pivots.cpp:670)
    if (rc == MI32_SINGULAR && env_int("MI32_SINGULAR_KEEP", 0)) return result;
    if (rc == MI32_RUNTIME_ERROR) std::fprintf(stderr, "matrix_inversion_no_pivots: %s\n", mi32_last_error());
    return {};
}

// ---- the reference's benchmark twin, unchanged signature (matrix_inversion/headers.h:15, FP32_bench.cpp:11) ----
Res FP32_bench(std::vector<float> matrix_vector, int matrix_order)
{
    Res res;
    if (matrix_order <= 0) return res;                                                       // FP32_bench.cpp:212
    if ((int)(matrix_vector.size() / (size_t)matrix_order) != matrix_order) return res;      // :217
    std::vector<float> inv((size_t)matrix_order * matrix_order, 0.0f);
    std::vector<double> times(10, 0.0);
    const int rc = mi32_bench_32(matrix_vector.data(), matrix_vector.size(), matrix_order, inv.data(), times.data());
    if (rc != MI32_OK) return res;   // {} like the reference's error paths (:456)
    res.inversa32 = std::move(inv);
    res.times = std::move(times);
    return res;
}

Res FP64_bench(std::vector<double> matrix_vector, int matrix_order)
{
    Res res;
    if (matrix_order <= 0) return res;
    if ((int)(matrix_vector.size() / (size_t)matrix_order) != matrix_order) return res;
    std::vector<double> inv((size_t)matrix_order * matrix_order, 0.0), times(10, 0.0);
    if (mi32_bench_64(matrix_vector.data(), matrix_vector.size(), matrix_order, inv.data(), times.data(), 1) != MI32_OK) return res;
    res.inversa64 = std::move(inv);
    res.times = std::move(times);
    return res;
}

Res no_pivots_bench(std::vector<double> matrix_vector, int matrix_order)
{
    Res res;
    if (matrix_order <= 0) return res;
    if ((int)(matrix_vector.size() / (size_t)matrix_order) != matrix_order) return res;
    std::vector<double> inv((size_t)matrix_order * matrix_order, 0.0), times(10, 0.0);
    if (mi32_bench_64(matrix_vector.data(), matrix_vector.size(), matrix_order, inv.data(), times.data(), 0) != MI32_OK) return res;
    res.inversa64 = std::move(inv);
    res.times = std::move(times);
    return res;
}

// the experiment twin of matrix_inv_32 (headers.h:7, matrix_inversion_FP32.cpp:11): same call shape; {} for an
// invalid matrix (its exact-identity check of the reduced left half, :814-835)
std::vector<float> matrix_inversion_FP32(std::vector<float> matrix_vector, int matrix_order)
{
    return matrix_inv_32(static_cast<std::vector<float> &&>(matrix_vector), matrix_order);
}

// headers.h:5, matrix_multiply.cpp:15: sqrt(N) - ||A * B||_F, N = sqrt(size)
double matrix_multiply(std::vector<double> matriceA, std::vector<double> matriceB)
{
    double errore = std::nan("");
    if (matriceA.size() != matriceB.size()) return errore;
    const int rc = mi32_matrix_multiply_64(matriceA.data(), matriceB.data(), matriceA.size(), &errore);
    if (rc == MI32_RUNTIME_ERROR) std::fprintf(stderr, "matrix_multiply: %s\n", mi32_last_error());
    return errore;
}

// ---- the reference's fp64 entry point, unchanged signature (matrix_inversion/headers.h:9) ----
std::vector<double> matrix_inversion_FP64(std::vector<double> matrix_vector, int matrix_order)
{
    if (matrix_order <= 0) return {};
    if ((int)(matrix_vector.size() / (size_t)matrix_order) != matrix_order) return {};
    std::vector<double> result;  // allocated and first touched while the device works (see matrix_inv_32)
    struct Ctx { std::vector<double> *v; size_t n; } ctx = {&result, (size_t)matrix_order * matrix_order};
    const int rc = host_invert_64(matrix_vector.data(), matrix_vector.size(), matrix_order, nullptr, true, nullptr,
                                  [](void *c) -> void * {
                                      Ctx *x = static_cast<Ctx *>(c);
                                      try {
                                          x->v->reserve(x->n);
                                          parallel_populate(x->v->data(), x->n * sizeof(double));
                                          x->v->resize(x->n);
                                      } catch (...) { return nullptr; }
                                      return x->v->data();
                                  }, &ctx);
    if (rc == MI32_OK) return result;
    // a singular input: {} like the reference (its exact-identity check, matrix_inversion_FP64.cpp:846-867)
    if (rc == MI32_SINGULAR && env_int("MI32_SINGULAR_KEEP", 0)) return result;
    if (rc == MI32_RUNTIME_ERROR) std::fprintf(stderr, "matrix_inversion_FP64: %s\n", mi32_last_error());
    return {};
}
