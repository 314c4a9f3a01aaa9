// hx_runtime.hip -- device selection, streams, timers, error state of libhxsht.so.
#include <sys/mman.h>

#include "hx_common.h"

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>

namespace hx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

Runtime &rt()
{
    static Runtime r;
    return r;
}

int ensure_ready()
{
    if (rt().ready) return HX_OK;
    int dev = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(HX_ERR_NO_DEVICE, "no HIP device available (libhxsht has no CPU fallback)");
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return hx_init(dev);
}

ProfScope::ProfScope(const char *n) : name(n)
{
    Runtime &r = rt();
    if (!r.profiling) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        a = b = nullptr;
        return;
    }
    (void)hipEventRecord(a, r.stream);
}

ProfScope::~ProfScope()
{
    Runtime &r = rt();
    if (!a || !b) return;
    (void)hipEventRecord(b, r.stream);
    r.prof[name].pending.emplace_back(a, b);
}

bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear: plain host memory is reported as an error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int DevBuf::alloc(size_t n)
{
    if (n <= bytes && p) return HX_OK;
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
        p = nullptr;
        return fail(HX_ERR_MEM, "hipMalloc of %zu bytes failed: %s", n, hipGetErrorString(e));
    }
    bytes = n;
    return HX_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

// ---- host <-> device staging ---------------------------------------------------------------
// hipMemcpy to or from PAGEABLE host memory is staged by the runtime on one thread (measured
// 14.5 GB/s for the 1.6 GB maps of the Mapper boundary, seven times the transform's compute
// time).  Large pageable transfers therefore go through two pinned buffers of the library: chunks
// are copied by several host threads (which also spreads the first-touch page faults of a fresh
// numpy output array) while the previous chunk is on the PCIe link.
namespace {

constexpr size_t STAGE_CHUNK = (size_t)64 << 20;
constexpr size_t STAGE_MIN = (size_t)16 << 20;  // below this the plain copy is as good

// CPUs of the NUMA node the GPU hangs on (sysfs: numa_node of its PCI device, cpulist of that node); empty if unknown.
// The staging threads and the pinned buffers are kept there: a copy that crosses the socket interconnect twice
// (pageable -> pinned on the far socket -> PCIe root on the near one) ran at 30 GB/s instead of 54.
std::vector<int> gpu_node_cpus()
{
    std::vector<int> cpus;
    char bus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) {
        (void)hipGetLastError();
        return cpus;
    }
    for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    int node = -1;
    if (FILE *f = fopen(path, "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return cpus;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return cpus;
    int a = 0, b = 0;
    while (fscanf(f, "%d", &a) == 1) {
        b = a;
        int ch = fgetc(f);
        if (ch == '-') {
            if (fscanf(f, "%d", &b) != 1) b = a;
            ch = fgetc(f);
        }
        for (int c = a; c <= b; ++c) cpus.push_back(c);
        if (ch != ',') break;
    }
    fclose(f);
    return cpus;
}

void pin_thread_to(const std::vector<int> &cpus)
{
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
    (void)sched_setaffinity(0, sizeof(set), &set);  // best effort: a container may forbid it
}

// Persistent copy workers (a fresh std::thread per 64 MB chunk and worker cost about as much as the chunk's copy).
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    char *dst = nullptr;
    const char *src = nullptr;
    size_t n = 0, per = 0;
    unsigned long long gen = 0;
    int pending = 0;
    void start(int nthreads, const std::vector<int> &cpus)
    {
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([this, t, cpus] {
                pin_thread_to(cpus);
                unsigned long long seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_go.wait(lk, [&] { return gen != seen; });
                    seen = gen;
                    char *d = dst;
                    const char *s = src;
                    const size_t total = n, chunk = per;
                    lk.unlock();
                    const size_t off = (size_t)t * chunk;
                    if (off < total) memcpy(d + off, s + off, std::min(chunk, total - off));
                    lk.lock();
                    if (--pending == 0) cv_done.notify_one();
                }
            });
        for (auto &x : th) x.detach();  // the pool lives as long as the process
    }
    void copy(void *d, const void *s, size_t bytes)
    {
        std::unique_lock<std::mutex> lk(mu);
        dst = (char *)d; src = (const char *)s; n = bytes;
        per = ((bytes / th.size()) + 4095) & ~(size_t)4095;
        pending = (int)th.size();
        ++gen;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

struct Stager {
    static constexpr int NPIN = 4;  // uploads rotate through all four (three DMAs queued while the host fills the fourth); downloads use the first two
    void *pin[NPIN] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[NPIN] = {nullptr, nullptr, nullptr, nullptr};
    int nthreads = 1;
    bool ok = false, tried = false;
    CopyPool *pool = nullptr;  // leaked on purpose: its detached workers may outlive static destruction
    bool init()
    {
        if (tried) return ok;
        tried = true;
        const std::vector<int> cpus = gpu_node_cpus();
        // pinned buffers allocated and first touched from the GPU's node (the calling thread goes back afterwards)
        cpu_set_t old;
        const bool have_old = sched_getaffinity(0, sizeof(old), &old) == 0;
        pin_thread_to(cpus);
        for (int i = 0; i < NPIN; ++i) {
            if (hipHostMalloc(&pin[i], STAGE_CHUNK, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                if (have_old) (void)sched_setaffinity(0, sizeof(old), &old);
                return false;
            }
            memset(pin[i], 0, STAGE_CHUNK);
        }
        if (have_old) (void)sched_setaffinity(0, sizeof(old), &old);
        const unsigned hw = cpus.empty() ? std::thread::hardware_concurrency() : (unsigned)cpus.size();
        nthreads = (int)std::min(16u, std::max(1u, hw / 2));
        if (const char *e = getenv("HX_COPY_THREADS")) nthreads = std::max(1, atoi(e));
        if (nthreads > 1) {
            pool = new CopyPool;
            pool->start(nthreads, cpus);
            static bool fork_handler = false;
            if (!fork_handler) {
                fork_handler = true;
                (void)pthread_atfork(nullptr, nullptr, forget_pool_in_child);
            }
        }
        ok = true;
        return true;
    }
    static void forget_pool_in_child();
    // hx_init on another device: the events belong to the old device's context
    void reset_events()
    {
        for (int i = 0; i < NPIN; ++i)
            if (ev[i]) {
                (void)hipEventDestroy(ev[i]);
                ev[i] = nullptr;
                if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    ev[i] = nullptr;
                    ok = false;
                }
            }
    }
};
Stager &stager()
{
    static Stager s;
    return s;
}
// A forked child has none of the detached copy workers: pool->copy would wait for them forever.  The child falls back to a
// plain memcpy into the pinned buffers (HIP itself does not survive fork() either; this only keeps a child that never touches
// the GPU, or that re-initialises it, from hanging inside the stager).
void Stager::forget_pool_in_child() { stager().pool = nullptr; }

void parallel_memcpy(void *dst, const void *src, size_t n, int nthreads)
{
    CopyPool *pool = stager().pool;
    if (nthreads <= 1 || !pool || n < ((size_t)4 << 20)) {
        memcpy(dst, src, n);
        return;
    }
    pool->copy(dst, src, n);
}

}  // namespace

bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

void stager_reset_events() { stager().reset_events(); }

hipStream_t copy_stream()
{
    Runtime &r = rt();
    if (!r.copy && hipStreamCreateWithFlags(&r.copy, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        r.copy = nullptr;
    }
    return r.copy;
}

// The two pinned buffers, their events and the copy pool's job fields are one shared set: staging is serialised process-wide
// (ctypes releases the GIL, so two Python threads can be inside the library at once; everything else in the library is
// single-threaded by contract, include/hxsht.h).
static std::mutex g_stage_mu;

int copy_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t on)
{
    hipStream_t st = on ? on : rt().stream;
    std::lock_guard<std::mutex> stage_lock(g_stage_mu);
    Stager &s = stager();
    if (bytes < STAGE_MIN || is_pinned_host(src_host) || !s.init()) {
        HX_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        return HX_OK;
    }
    // equal chunks (a 67 MB copy as 64 + 3 MB has nothing to overlap its large DMA with), and the buffers alternate ACROSS calls: a
    // caller that uploads many pieces in a row (the ring slabs of hx_map2alm_multi: 40 pieces of 67 MB each) keeps the host copy of one
    // piece under the DMA of the piece before it -- starting every call on buffer 0 halved its rate (27 instead of 56 GB/s; two buffers in turn: 51)
    static int next_buf = 0;
    const size_t nchunk = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    const size_t chunk = ((bytes + nchunk - 1) / nchunk + 4095) & ~(size_t)4095;
    for (size_t off = 0; off < bytes; off += chunk) {
        const int i = next_buf;
        next_buf = (next_buf + 1) % Stager::NPIN;
        const size_t len = std::min(chunk, bytes - off);
        HX_HIP(hipEventSynchronize(s.ev[i]));  // the DMA that last read this buffer has finished
        parallel_memcpy(s.pin[i], (const char *)src_host + off, len, s.nthreads);
        HX_HIP(hipMemcpyAsync((char *)dst_dev + off, s.pin[i], len, hipMemcpyHostToDevice, st));
        HX_HIP(hipEventRecord(s.ev[i], st));
    }
    return HX_OK;
}

int copy_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t on)
{
    hipStream_t st = on ? on : rt().stream;
    std::lock_guard<std::mutex> stage_lock(g_stage_mu);
    Stager &s = stager();
    if (bytes < STAGE_MIN || is_pinned_host(dst_host) || !s.init()) {
        HX_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        HX_HIP(hipStreamSynchronize(st));
        return HX_OK;
    }
    if (bytes >= (size_t)64 << 20) {
        // a large destination is usually fresh memory (np.empty): ask for transparent huge pages before the first touch -- 512 times
        // fewer page faults under the copy threads (906 MB of mixing matrices: 89 -> 35 ms of first touch on the GPU box, where THP is
        // in "madvise" mode).  A hint: errors are ignored.
        const uintptr_t a0 = ((uintptr_t)dst_host + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
        const uintptr_t a1 = ((uintptr_t)dst_host + bytes) & ~(((uintptr_t)2 << 20) - 1);
        if (a1 > a0) (void)madvise((void *)a0, (size_t)(a1 - a0), MADV_HUGEPAGE);
    }
    // (an upload on another stream may still be reading the two buffers used here)
    HX_HIP(hipEventSynchronize(s.ev[0]));
    HX_HIP(hipEventSynchronize(s.ev[1]));
    const size_t nchunk = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    auto issue = [&](size_t c) -> int {
        const size_t off = c * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
        HX_HIP(hipMemcpyAsync(s.pin[c & 1], (const char *)src_dev + off, len, hipMemcpyDeviceToHost, st));
        HX_HIP(hipEventRecord(s.ev[c & 1], st));
        return HX_OK;
    };
    HX_TRY(issue(0));
    for (size_t c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) HX_TRY(issue(c + 1));  // its buffer was drained in the previous iteration
        HX_HIP(hipEventSynchronize(s.ev[c & 1]));
        const size_t off = c * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
        parallel_memcpy((char *)dst_host + off, s.pin[c & 1], len, s.nthreads);
    }
    return HX_OK;
}

int InView::bind(const void *src, size_t bytes)
{
    if (!src) {
        dev = nullptr;
        return HX_OK;
    }
    if (is_device_ptr(src)) {
        dev = src;
        return HX_OK;
    }
    HX_TRY(tmp.alloc(bytes));
    HX_TRY(copy_h2d(tmp.p, src, bytes));
    dev = tmp.p;
    return HX_OK;
}

int OutView::bind(void *dst, size_t n)
{
    bytes = n;
    if (is_device_ptr(dst)) {
        dev = dst;
        host = nullptr;
        return HX_OK;
    }
    HX_TRY(tmp.alloc(n));
    dev = tmp.p;
    host = dst;
    return HX_OK;
}

int OutView::finish()
{
    if (!host) return HX_OK;
    return copy_d2h(host, dev, bytes);  // synchronous on return
}

static double g_scratch_budget = -1.0;  // < 0: not yet read from the environment

double scratch_budget_bytes()
{
    if (g_scratch_budget < 0.0) {
        const char *e = getenv("HX_SCRATCH_GB");
        g_scratch_budget = e ? std::max(0.0, atof(e)) * 1e9 : 0.0;
    }
    return g_scratch_budget;
}

int finish_call()
{
    HX_HIP(hipGetLastError());
    if (!rt().async) HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

}  // namespace hx

using namespace hx;

extern "C" {

const char *hx_version(void) { return "hxsht 0.1 (gfx950)"; }

const char *hx_last_error(void) { return hx::g_err; }

int hx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int hx_init(int device)
{
    Runtime &r = rt();
    int n = hx_device_count();
    if (n <= 0) return fail(HX_ERR_NO_DEVICE, "no HIP device available (libhxsht has no CPU fallback)");
    if (device < 0 || device >= n) return fail(HX_ERR_ARG, "device %d out of range [0,%d)", device, n);
    HX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    HX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HX_ERR_NO_DEVICE, "device %d is %s; libhxsht is built for gfx950 only", device,
                    prop.gcnArchName);
    if (r.ready && r.device == device) return HX_OK;
    if (r.own_stream && r.stream) (void)hipStreamDestroy(r.stream);
    if (r.ready) {
        // re-initialisation on another device: the second (copy) stream and the stager's events were created on the old one
        if (r.copy) (void)hipStreamDestroy(r.copy);
        r.copy = nullptr;
        if (r.order_ev) (void)hipEventDestroy(r.order_ev);
        r.order_ev = nullptr;
        mixmat_drop_cache();
        alm2cl_drop_cache();
        stager_reset_events();
    }
    HX_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    r.own_stream = true;
    if (!r.t0) HX_HIP(hipEventCreate(&r.t0));
    if (!r.t1) HX_HIP(hipEventCreate(&r.t1));
    r.device = device;
    r.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    r.ready = true;
    return HX_OK;
}

int hx_set_stream(void *stream)
{
    HX_TRY(ensure_ready());
    Runtime &r = rt();
    if (stream == nullptr) {
        if (!r.own_stream) {
            HX_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
            r.own_stream = true;
        }
        return HX_OK;
    }
    if (r.own_stream && r.stream) (void)hipStreamDestroy(r.stream);
    r.stream = static_cast<hipStream_t>(stream);
    r.own_stream = false;
    return HX_OK;
}

// dst <- src: either side host or device memory.  Host <-> device goes through the pinned staging pipeline (~55 GB/s from / to
// pageable memory, several host threads), which the runtime's own hipMemcpy of pageable memory does not (14 GB/s).  Complete on return.
int hx_copy(void *dst, const void *src, int64_t bytes)
{
    HX_TRY(ensure_ready());
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(HX_ERR_ARG, "hx_copy: bad arguments");
    if (bytes == 0) return HX_OK;
    const bool dd = is_device_ptr(dst), sd = is_device_ptr(src);
    if (dd && sd) {
        HX_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, rt().stream));
    } else if (dd) {
        HX_TRY(copy_h2d(dst, src, (size_t)bytes));
    } else if (sd) {
        return copy_d2h(dst, src, (size_t)bytes);
    } else {
        memcpy(dst, src, (size_t)bytes);
        return HX_OK;
    }
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

int hx_host_alloc(int64_t bytes, void **out)
{
    HX_TRY(ensure_ready());
    if (bytes <= 0 || !out) return fail(HX_ERR_ARG, "hx_host_alloc: bad arguments");
    *out = nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(HX_ERR_MEM, "hx_host_alloc: %lld bytes of page-locked host memory not available", (long long)bytes);
    }
    *out = p;
    return HX_OK;
}

int hx_host_free(void *p)
{
    if (!p) return HX_OK;
    if (hipHostFree(p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(HX_ERR_ARG, "hx_host_free: not a pointer of hx_host_alloc");
    }
    return HX_OK;
}

void *hx_get_stream(void)
{
    if (ensure_ready() != HX_OK) return nullptr;
    return rt().stream;
}

int hx_set_async(int on)
{
    rt().async = on != 0;
    return HX_OK;
}

int hx_set_scratch_budget(double bytes)
{
    if (!(bytes >= 0.0)) return fail(HX_ERR_ARG, "hx_set_scratch_budget: negative or NaN budget");
    hx::g_scratch_budget = bytes;
    return HX_OK;
}

int hx_synchronize(void)
{
    HX_TRY(ensure_ready());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

int hx_timer_start(void)
{
    HX_TRY(ensure_ready());
    HX_HIP(hipEventRecord(rt().t0, rt().stream));
    return HX_OK;
}

int hx_timer_stop(float *ms)
{
    HX_TRY(ensure_ready());
    HX_HIP(hipEventRecord(rt().t1, rt().stream));
    HX_HIP(hipEventSynchronize(rt().t1));
    float t = 0.f;
    HX_HIP(hipEventElapsedTime(&t, rt().t0, rt().t1));
    if (ms) *ms = t;
    return HX_OK;
}

int hx_profile_enable(int on)
{
    rt().profiling = on != 0;
    return HX_OK;
}

static void drain(Runtime::Prof &p)
{
    for (auto &ev : p.pending) {
        float t = 0.f;
        if (hipEventSynchronize(ev.second) == hipSuccess &&
            hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) {
            p.ms += t;
            p.launches += 1;
        }
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    p.pending.clear();
}

int hx_profile_reset(void)
{
    for (auto &kv : rt().prof) {
        drain(kv.second);
        kv.second.ms = 0.0;
        kv.second.launches = 0;
    }
    return HX_OK;
}

int hx_profile_get(const char *name, int *launches, double *total_ms)
{
    auto it = rt().prof.find(name ? name : "");
    if (it == rt().prof.end()) {
        if (launches) *launches = 0;
        if (total_ms) *total_ms = 0.0;
        return HX_OK;
    }
    drain(it->second);
    if (launches) *launches = it->second.launches;
    if (total_ms) *total_ms = it->second.ms;
    return HX_OK;
}

}  // extern "C"
