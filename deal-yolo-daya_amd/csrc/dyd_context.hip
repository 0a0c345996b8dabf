// dyd_context.hip — context, error state, device memory and the host-side MT19937
// permutation of libdyd_gfx950.so (C ABI: include/dyd.h).
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "dyd_common.h"

namespace dyd {

static thread_local char g_err[1024] = "";
static thread_local double g_last_ms = 0.0;

Context &ctx() {
    static Context c;
    return c;
}
std::recursive_mutex &api_mutex() {
    static std::recursive_mutex m;
    return m;
}
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void set_last_kernel_ms(double ms) { g_last_ms = ms; }

// ---- staging slots ------------------------------------------------------------------------------------------------------
namespace {
struct StagePool {
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::unique_ptr<StageSlot>> slots;
    size_t pinned_total = 0;
};
StagePool &stage_pool() {
    static StagePool p;
    return p;
}
size_t pinned_budget() {          // bytes of pinned host memory all slots together may hold (DYD_PINNED_POOL_MB, default 1024)
    static size_t cached = 0;
    if (!cached) {
        size_t mb = 1024;
        if (const char *e = getenv("DYD_PINNED_POOL_MB")) { const long v = atol(e); if (v >= 0) mb = (size_t)v; }
        cached = (mb << 20) + 1;
    }
    return cached - 1;
}
constexpr size_t kMaxSlots = 64;
}  // namespace

int stage_acquire(size_t pin_bytes, StageSlot **out) {
    {
        std::lock_guard<std::recursive_mutex> lock(api_mutex());
        const int rc = ensure_init();
        if (rc != DYD_OK) return rc;
    }
    StagePool &P = stage_pool();
    StageSlot *slot = nullptr;
    {
        std::unique_lock<std::mutex> lk(P.m);
        for (;;) {
            // the free slot whose pinned arena fits best (smallest sufficient, else the largest there is)
            StageSlot *best = nullptr;
            for (auto &q : P.slots) {
                if (q->busy) continue;
                if (!best) { best = q.get(); continue; }
                const bool qf = q->pin_cap >= pin_bytes, bf = best->pin_cap >= pin_bytes;
                if ((qf && !bf) || (qf && bf && q->pin_cap < best->pin_cap) || (!qf && !bf && q->pin_cap > best->pin_cap)) best = q.get();
            }
            if (best) { slot = best; break; }
            if (P.slots.size() < kMaxSlots) {
                P.slots.emplace_back(new StageSlot());
                slot = P.slots.back().get();
                break;
            }
            P.cv.wait(lk);
        }
        slot->busy = true;
    }
    auto fail = [&](int rc) {
        stage_release(slot);
        return rc;
    };
    if (!slot->s) {
        if (hipStreamCreateWithFlags(&slot->s, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&slot->e0) != hipSuccess ||
            hipEventCreate(&slot->e1) != hipSuccess) {
            set_error("staging slot: stream / event creation failed");
            return fail(DYD_ERR_HIP);
        }
    }
    if (pin_bytes > slot->pin_cap) {
        const size_t want = pin_bytes + (pin_bytes >> 3);
        bool allowed;
        {
            std::lock_guard<std::mutex> lk(P.m);
            allowed = P.pinned_total - slot->pin_cap + want <= pinned_budget();
            if (allowed) P.pinned_total += want - slot->pin_cap;
        }
        if (allowed) {
            if (slot->pin) (void)hipHostFree(slot->pin);
            slot->pin = nullptr;
            const size_t old_cap = slot->pin_cap;
            slot->pin_cap = 0;
            if (hipHostMalloc(&slot->pin, want, hipHostMallocDefault) == hipSuccess) {
                slot->pin_cap = want;
            } else {                                   // no pinned memory to be had: the caller stages from pageable memory
                slot->pin = nullptr;
                (void)hipGetLastError();
                std::lock_guard<std::mutex> lk(P.m);
                P.pinned_total -= want;
                (void)old_cap;
            }
        }
    }
    *out = slot;
    return DYD_OK;
}

int stage_device(StageSlot *slot, size_t bytes) {
    if (bytes <= slot->dev_cap) return DYD_OK;
    if (slot->dev) {
        (void)hipStreamSynchronize(slot->s);
        (void)hipFree(slot->dev);
        slot->dev = nullptr;
        slot->dev_cap = 0;
    }
    const size_t want = bytes + (bytes >> 2);
    const hipError_t e = hipMalloc(&slot->dev, want);
    if (e != hipSuccess) {
        slot->dev = nullptr;
        set_error("staging slot: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return DYD_ERR_OOM;
    }
    slot->dev_cap = want;
    return DYD_OK;
}

void stage_release(StageSlot *slot) {
    if (!slot) return;
    StagePool &P = stage_pool();
    {
        std::lock_guard<std::mutex> lk(P.m);
        slot->busy = false;
    }
    P.cv.notify_one();
}

void stage_free_all() {
    StagePool &P = stage_pool();
    std::lock_guard<std::mutex> lk(P.m);
    for (auto &q : P.slots) {
        if (q->busy) continue;                         // a pass in flight keeps its slot; it is dropped with the next call
        if (q->s) (void)hipStreamSynchronize(q->s);
        if (q->dev) (void)hipFree(q->dev);
        if (q->pin) (void)hipHostFree(q->pin);
        if (q->e0) (void)hipEventDestroy(q->e0);
        if (q->e1) (void)hipEventDestroy(q->e1);
        if (q->s) (void)hipStreamDestroy(q->s);
        P.pinned_total -= q->pin_cap;
        *q = StageSlot();
    }
    P.slots.erase(std::remove_if(P.slots.begin(), P.slots.end(), [](const std::unique_ptr<StageSlot> &q) { return !q->busy; }), P.slots.end());
}

static int init_locked(int device) {
    Context &c = ctx();
    if (c.ready && (device < 0 || device == c.device)) return DYD_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return DYD_ERR_NO_DEVICE;
    }
    if (device < 0) {
        int cur = 0;
        device = (hipGetDevice(&cur) == hipSuccess) ? cur : 0;
    }
    if (device >= n) {
        set_error("device %d requested but only %d visible", device, n);
        return DYD_ERR_NO_DEVICE;
    }
    if (c.ready) {  // rebinding to another device: drop the old context first
        (void)hipSetDevice(c.device);
        stage_free_all();
        if (c.scratch) (void)hipFree(c.scratch);
        if (c.dev_status) (void)hipFree(c.dev_status);
        if (c.bigq) (void)hipFree(c.bigq);
        if (c.bigq_ev) (void)hipEventDestroy(c.bigq_ev);
        if (c.ev0) (void)hipEventDestroy(c.ev0);
        if (c.ev1) (void)hipEventDestroy(c.ev1);
        if (c.scratch_ev) (void)hipEventDestroy(c.scratch_ev);
        if (c.stream) (void)hipStreamDestroy(c.stream);
        c = Context();
    }
    DYD_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    DYD_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libdyd_gfx950.so carries gfx950 (MI355X) code only", device,
                  prop.gcnArchName);
        return DYD_ERR_NO_DEVICE;
    }
    c.device = device;
    c.num_cu = prop.multiProcessorCount;
    snprintf(c.name, sizeof(c.name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, c.num_cu);
    DYD_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    DYD_HIP(hipEventCreate(&c.ev0));
    DYD_HIP(hipEventCreate(&c.ev1));
    DYD_HIP(hipEventCreateWithFlags(&c.scratch_ev, hipEventDisableTiming));
    {   // freed DevBuf memory stays in the pool for the next call (dyd_shutdown trims it)
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
            uint64_t keep = ~(uint64_t)0;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    DYD_HIP(hipMalloc(reinterpret_cast<void **>(&c.dev_status), 16));
    DYD_HIP(hipMemset(c.dev_status, 0, 16));
    c.ready = true;
    return DYD_OK;
}

int ensure_init() {
    if (ctx().ready) {
        // HIP's current device is per thread; bind the calling thread to the context's device
        hipError_t e = hipSetDevice(ctx().device);
        if (e != hipSuccess) {
            set_error("hipSetDevice(%d) failed: %s", ctx().device, hipGetErrorString(e));
            return DYD_ERR_HIP;
        }
        return DYD_OK;
    }
    return init_locked(-1);
}

void release_scratch(hipStream_t st) {
    Context &c = ctx();
    if (hipEventRecord(c.scratch_ev, st) == hipSuccess) c.scratch_used = true;
}

int get_scratch(size_t bytes, void **out, hipStream_t st) {
    Context &c = ctx();
    if (c.scratch_used) DYD_HIP(hipStreamWaitEvent(st, c.scratch_ev, 0));
    if (bytes > c.scratch_bytes) {
        if (c.scratch) {
            DYD_HIP(hipDeviceSynchronize());
            (void)hipFree(c.scratch);
            c.scratch = nullptr;
            c.scratch_bytes = 0;
        }
        size_t want = bytes + (bytes >> 2);
        hipError_t e = hipMalloc(&c.scratch, want);
        if (e != hipSuccess) {
            c.scratch = nullptr;
            set_error("scratch hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return DYD_ERR_OOM;
        }
        c.scratch_bytes = want;
    }
    *out = c.scratch;
    return DYD_OK;
}

int take_device_status(hipStream_t st, const char *what) {
    Context &c = ctx();
    int bits = 0;
    DYD_HIP(hipMemcpyAsync(&bits, c.dev_status, 4, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    if (bits == 0) return DYD_OK;
    DYD_HIP(hipMemsetAsync(c.dev_status, 0, 4, st));
    DYD_HIP(hipStreamSynchronize(st));
    set_error("%s: device-side failure%s%s (status 0x%x) — the result must not be used", what,
              (bits & 1) ? ", hash table full" : "", (bits & 2) ? ", an inserted key was not found again" : "", bits);
    return DYD_ERR_HIP;
}

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_device_status(void *stream) {
    DYD_API_ENTER();
    return take_device_status(pick_stream(stream), "dyd_device_status");
}

int dyd_init(int device_or_minus1) {
    std::lock_guard<std::recursive_mutex> lock(api_mutex());
    return init_locked(device_or_minus1);
}

void dyd_shutdown(void) {
    std::lock_guard<std::recursive_mutex> lock(api_mutex());
    Context &c = ctx();
    if (!c.ready) return;
    (void)hipSetDevice(c.device);
    (void)hipDeviceSynchronize();
    stage_free_all();
    {
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, c.device) == hipSuccess && pool) (void)hipMemPoolTrimTo(pool, 0);
        (void)hipGetLastError();
    }
    if (c.scratch) (void)hipFree(c.scratch);
    if (c.dev_status) (void)hipFree(c.dev_status);
    if (c.bigq) (void)hipFree(c.bigq);
    if (c.bigq_ev) (void)hipEventDestroy(c.bigq_ev);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    if (c.scratch_ev) (void)hipEventDestroy(c.scratch_ev);
    if (c.stream) (void)hipStreamDestroy(c.stream);
    c = Context();
}

const char *dyd_last_error(void) { return g_err; }

int dyd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *dyd_version(void) { return "dyd 0.1.0 (gfx950)"; }

const char *dyd_device_name(void) { return ctx().name; }

int dyd_malloc(void **dptr, size_t bytes) {
    DYD_API_ENTER();
    DYD_REQUIRE(dptr != nullptr, "dptr is null");
    *dptr = nullptr;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) {
        *dptr = nullptr;
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return DYD_ERR_OOM;
    }
    return DYD_OK;
}

int dyd_free(void *dptr) {
    DYD_API_ENTER();
    if (dptr) DYD_HIP(hipFree(dptr));
    return DYD_OK;
}

int dyd_h2d(void *dst_dev, const void *src_host, size_t bytes) {
    DYD_API_ENTER();
    if (bytes == 0) return DYD_OK;
    DYD_REQUIRE(dst_dev && src_host, "null pointer");
    DYD_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx().stream));
    DYD_HIP(hipStreamSynchronize(ctx().stream));
    return DYD_OK;
}

int dyd_d2h(void *dst_host, const void *src_dev, size_t bytes) {
    DYD_API_ENTER();
    if (bytes == 0) return DYD_OK;
    DYD_REQUIRE(dst_host && src_dev, "null pointer");
    DYD_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx().stream));
    DYD_HIP(hipStreamSynchronize(ctx().stream));
    return DYD_OK;
}

int dyd_memset(void *dst_dev, int byte, size_t bytes) {
    DYD_API_ENTER();
    if (bytes == 0) return DYD_OK;
    DYD_REQUIRE(dst_dev, "null pointer");
    DYD_HIP(hipMemsetAsync(dst_dev, byte, bytes, ctx().stream));
    DYD_HIP(hipStreamSynchronize(ctx().stream));
    return DYD_OK;
}

int dyd_sync(void *stream) {
    DYD_API_ENTER();
    DYD_HIP(hipStreamSynchronize(pick_stream(stream)));
    return DYD_OK;
}

double dyd_last_kernel_ms(void) { return g_last_ms; }

// ---- host code: numpy legacy RandomState(seed).permutation(n) ----------------------------
// DataFrame.sample(frac=1, random_state=seed) at processor.py:800 resolves to
// RandomState(seed).choice(n, n, replace=False) == permutation(n): MT19937 seeded by
// init_genrand, then a reversed Fisher-Yates whose index draw is a 32-bit output masked to
// the next 2^k-1 and rejected while it exceeds the bound.  Inherently sequential, so it runs
// on the host; the device only applies the permutation (dyd_split_ids).
namespace {
struct Mt {
    uint32_t s[624];
    int pos;
    explicit Mt(uint32_t seed) {
        s[0] = seed;
        for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    void refill() {
        constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
        int k = 0;
        for (; k < 624 - 397; ++k) {
            uint32_t y = (s[k] & UP) | (s[k + 1] & LO);
            s[k] = s[k + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        for (; k < 623; ++k) {
            uint32_t y = (s[k] & UP) | (s[k + 1] & LO);
            s[k] = s[k + 397 - 624] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        uint32_t y = (s[623] & UP) | (s[0] & LO);
        s[623] = s[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    uint32_t next() {
        if (pos >= 624) refill();
        uint32_t y = s[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};
}  // namespace

int dyd_mt19937_permutation(uint32_t seed, int64_t n, int64_t *out) {
    if (n < 0 || (n > 0 && !out)) {
        set_error("invalid argument: n < 0 or out is null");
        return DYD_ERR_INVALID;
    }
    Mt g(seed);
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    // The draws do not depend on the array, only the swaps do: the partner indices of a block of steps are drawn
    // first and their cache lines requested, then the swaps are applied in order.  On a table beyond the caches
    // (165 M records = 1.3 GB) the swap partner is a DRAM miss every step; with the misses of a block in flight
    // together the loop runs ~3x faster than one miss at a time.
    constexpr int BLOCK = 64;
    int64_t partner[BLOCK];
    int64_t i = n - 1;
    while (i >= 1) {
        const int m = (int)((i < BLOCK) ? i : BLOCK);           // steps i, i-1, ..., i-m+1
        for (int k = 0; k < m; ++k) {
            const uint64_t bound = (uint64_t)(i - k);
            uint64_t mask = bound;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
            mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
            uint64_t j;
            if (bound <= 0xffffffffull) {
                do { j = g.next() & mask; } while (j > bound);
            } else {
                do {
                    uint64_t hi = g.next(), lo = g.next();
                    j = ((hi << 32) | lo) & mask;
                } while (j > bound);
            }
            partner[k] = (int64_t)j;
            __builtin_prefetch(out + j, 1, 0);
        }
        for (int k = 0; k < m; ++k) {
            const int64_t a = i - k, j = partner[k];
            const int64_t t = out[a];
            out[a] = out[j];
            out[j] = t;
        }
        i -= m;
    }
    return DYD_OK;
}

}  // extern "C"
