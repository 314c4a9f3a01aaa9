// hx_common.h -- runtime state, error handling and device-buffer helpers shared by the
// translation units of libhxsht.so.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/hxsht.h"

namespace hx {

// ---- error state -------------------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define HX_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t _e = (call);                                                          \
        if (_e != hipSuccess)                                                            \
            return ::hx::fail(HX_ERR_HIP, "%s failed: %s (%s:%d)", #call,                \
                              hipGetErrorString(_e), __FILE__, __LINE__);                \
    } while (0)

#define HX_TRY(expr)                                                                     \
    do {                                                                                 \
        int _rc = (expr);                                                                \
        if (_rc != HX_OK) return _rc;                                                    \
    } while (0)

// ---- runtime -----------------------------------------------------------------------
struct Runtime {
    bool ready = false;
    int device = -1;
    int cus = 256;  // compute units of the device (grid size of persistent kernels)
    hipStream_t stream = nullptr;
    hipStream_t copy = nullptr;  // created on first use (copy_stream())
    bool own_stream = false;
    bool async = false;
    bool profiling = false;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipEvent_t order_ev = nullptr;  // library stream -> copy stream dependency (hx_mixmat.hip); recreated by hx_init on another device
    struct Prof {
        int launches = 0;
        double ms = 0.0;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    };
    std::map<std::string, Prof> prof;
    std::vector<hipEvent_t> event_pool;
};
Runtime &rt();
int ensure_ready();  // lazily hx_init(current device); HX_ERR_NO_DEVICE if none

// Scoped kernel-family timer: records HIP events on the library stream when profiling.
struct ProfScope {
    const char *name;
    hipEvent_t a = nullptr, b = nullptr;
    explicit ProfScope(const char *n);
    ~ProfScope();
};

// ---- device buffers ----------------------------------------------------------------
bool is_device_ptr(const void *p);
bool is_pinned_host(const void *p);  // page-locked host memory the runtime knows (hipHostMalloc / hipHostRegister)

// Owning device allocation.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n);    // (re)allocate exactly n bytes if current capacity < n
    void release();
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// Input view: device pointer as-is, host pointer copied to a temporary device buffer.
struct InView {
    const void *dev = nullptr;
    DevBuf tmp;
    int bind(const void *src, size_t bytes);
    template <class T> const T *as() const { return static_cast<const T *>(dev); }
};

// Output view: device pointer as-is; host pointer -> temporary device buffer, copied back
// by finish().
struct OutView {
    void *dev = nullptr;
    void *host = nullptr;
    size_t bytes = 0;
    DevBuf tmp;
    int bind(void *dst, size_t bytes);
    int finish();
    template <class T> T *as() const { return static_cast<T *>(dev); }
};

// Host <-> device copies on the library stream; large pageable host buffers are staged through
// pinned memory by several host threads.  copy_d2h is complete on return, copy_h2d is stream-ordered.
int copy_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t on = nullptr);  // on: another stream than the library's
void stager_reset_events();  // hx_init on another device: recreate the staging events there
hipStream_t copy_stream();  // second stream of the library (uploads that overlap its kernels); nullptr if it cannot be created
int copy_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t on = nullptr);  // on: another stream than the library's (the caller orders it behind the producer)

int finish_call();  // synchronise unless async

// Scratch budget of one analysis m-chunk in bytes (hx_set_scratch_budget, else HX_SCRATCH_GB); 0 = automatic
double scratch_budget_bytes();

// G = T diag(s) T2^T on the FP64 matrix unit (hx_mixmat.hip; used by hx_svd.hip)
int launch_gemm_tst(const double *T, int rows1_pad, const double *T2, int rows2_pad, int kpad, const double *s, int n1, int n2, double *G, long long ldg);

// hx_init on another device: drop the context hx_mixmat / hx_mixmat_eb keep between calls (hx_mixmat.hip)
void mixmat_drop_cache();
void alm2cl_drop_cache();  // hx_twopoint.hip: the buffers hx_alm2cl_pairs keeps between calls

// Gauss-Legendre nodes/weights into device arrays (hx_mixmat.hip)
int launch_gauss_legendre(int n, double *d_x, double *d_w, double *d_xlo = nullptr);  // d_xlo: node k = x[k] + xlo[k] (see k_gauss_legendre)

}  // namespace hx
