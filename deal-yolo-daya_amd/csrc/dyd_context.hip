// dyd_context.hip — context, error state, device memory and the host-side MT19937
// permutation of libdyd_gfx950.so (C ABI: include/dyd.h).
#include <cstring>

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
