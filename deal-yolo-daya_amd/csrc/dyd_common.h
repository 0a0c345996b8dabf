// dyd_common.h — internal helpers shared by the translation units of libdyd_gfx950.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>

#include "../../include/dyd.h"

namespace dyd {

constexpr int kWave = 64;

// Process-wide context, created lazily by dyd_init (reference processor.py has no
// equivalent: it never leaves CPython).
struct Context {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;  // the library's own stream (used when the caller passes NULL)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cu = 0;
    char name[256] = {0};
    // reusable device scratch (hash tables, staging), grown on demand
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    hipEvent_t scratch_ev = nullptr;  // recorded after the last kernel that used the scratch
    bool scratch_used = false;
    // one word on the device that kernels OR their failure bits into (K4: 1 = hash table full, 2 = an inserted key was not
    // found again); host-pointer entry points read and clear it before they return, `_dev` callers ask dyd_device_status
    int *dev_status = nullptr;
    // queue for image rows of thousands of boxes (k2_wave.h), shared by the K2 / fused entry points
    void *bigq = nullptr;
    hipEvent_t bigq_ev = nullptr;
    hipStream_t bigq_stream = nullptr;
    bool bigq_busy = false, bigq_dirty = false;
    int bigq_turn = 0;
};

// A staging slot: everything one host-pointer pass needs to reach the device and come back without creating or destroying
// anything — its own stream and two timing events, a device arena and a pinned host arena, all kept by the context between
// calls (round 2's host entries made a stream, two events and six hipMalloc'ed buffers per call and hipFree'd them on return:
// hipFree waits for the whole device, so thirty-two threads staging "side by side" took turns).  Slots are handed out under
// their own mutex, never under api_mutex.
struct StageSlot {
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    void *dev = nullptr;
    size_t dev_cap = 0;
    void *pin = nullptr;
    size_t pin_cap = 0;
    bool busy = false;
};
// a free slot (a new one while fewer than 64 exist, else the call waits), its pinned arena grown to `pin_bytes` if the
// context's pinned budget allows (pin_cap tells); the calling thread is bound to the context's device
int stage_acquire(size_t pin_bytes, StageSlot **out);
// the slot's device arena grown to at least `bytes` (contents are not kept)
int stage_device(StageSlot *slot, size_t bytes);
void stage_release(StageSlot *slot);
void stage_free_all();          // dyd_shutdown / rebinding; the caller holds api_mutex

Context &ctx();
std::recursive_mutex &api_mutex();
void set_error(const char *fmt, ...);
void set_last_kernel_ms(double ms);
int ensure_init();
// returns a device buffer of at least `bytes` owned by the context.  `st` is made to wait for the
// previous user of the scratch; call release_scratch(st) after the last launch that touches it.
int get_scratch(size_t bytes, void **out, hipStream_t st);
void release_scratch(hipStream_t st);
// reads and clears the device status word after synchronising `st`; returns DYD_OK or DYD_ERR_HIP with the message set
int take_device_status(hipStream_t st, const char *what);
// `_dev` entry points launch on exactly the hipStream_t they are given (NULL = HIP's null stream,
// which is also torch's default stream); the library's own stream serves the host-pointer calls.
inline hipStream_t pick_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define DYD_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            ::dyd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                             __LINE__);                                                    \
            return e__ == hipErrorOutOfMemory ? DYD_ERR_OOM : DYD_ERR_HIP;                 \
        }                                                                                  \
    } while (0)

#define DYD_REQUIRE(cond, msg)                      \
    do {                                            \
        if (!(cond)) {                              \
            ::dyd::set_error("invalid argument: %s", msg); \
            return DYD_ERR_INVALID;                 \
        }                                           \
    } while (0)

#define DYD_API_ENTER()                                              \
    std::lock_guard<std::recursive_mutex> lock__(::dyd::api_mutex()); \
    do {                                                             \
        int rc__ = ::dyd::ensure_init();                             \
        if (rc__ != DYD_OK) return rc__;                             \
    } while (0)

// RAII device buffer for the host-pointer entry points: stream-ordered allocation from the device's default memory pool, whose
// release threshold the context raises at start (dyd_context.hip), so a buffer freed on return is handed to the next call instead
// of going back to the driver — hipFree waits for the whole device, hipFreeAsync only takes its place in the stream.
struct DevBuf {
    void *p = nullptr;
    hipStream_t st = nullptr;
    ~DevBuf() {
        if (p) (void)hipFreeAsync(p, st);
    }
    int alloc(size_t bytes, hipStream_t stream = nullptr) {
        if (bytes == 0) bytes = 16;
        st = stream ? stream : ctx().stream;
        hipError_t e = hipMallocAsync(&p, bytes, st);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            set_error("hipMallocAsync(%zu) failed: %s", bytes, hipGetErrorString(e));
            return DYD_ERR_OOM;
        }
        return DYD_OK;
    }
    template <class T>
    T *as() { return static_cast<T *>(p); }
};

// times the kernels launched between begin() and end() on one stream
struct KernelTimer {
    hipStream_t s;
    explicit KernelTimer(hipStream_t st) : s(st) { (void)hipEventRecord(ctx().ev0, s); }
    void finish() {
        (void)hipEventRecord(ctx().ev1, s);
        (void)hipEventSynchronize(ctx().ev1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ctx().ev0, ctx().ev1);
        set_last_kernel_ms(ms);
    }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#if defined(__HIPCC__)
// Streaming (read-once) 16-byte load: `global_load_dwordx4 ... nt`.  Measured on MI355X with
// k0_membench: 6.6-6.9 TB/s for nt reads against 5.4-5.8 TB/s for plain ones (1.4 GB array), but a
// copy (reads + writes) gains only 3-7 %, K1 nothing and the fused wave kernel loses ~6 %, so the
// kernels use plain loads; kept for experiments.
typedef double dyd_f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_stream(const double2 *p) {
    const dyd_f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const dyd_f64x2 *>(p));
    return make_double2(v.x, v.y);
}
#endif

}  // namespace dyd
