// k4_dedup.hip — K4 / K5: occurrence masks and set membership over 128-bit hash keys.
//
// K4 replaces DataFrame.drop_duplicates(subset=["source"], keep=first|last|False)
//    (reference core/processor.py:140-144);
// K5 replaces Series.isin(set(ref)) (reference core/processor.py:194-199);
// dyd_dedup_global_dev is K4 over the keys of all ranks after the one allgather of the
//    multi-GPU path: every rank inserts all keys, resolves only its own rows.
//
// Data structure: an open-addressing table of ROW INDICES (int64, -1 = empty, capacity =
// power of two >= 2n, linear probing).  A slot is claimed with one 64-bit CAS; the key a slot
// stands for is read back from the immutable key array through the index it holds, so no
// 128-bit atomic is needed and a reader can never observe a half-written key.  keep=first /
// last fold the winning row with atomicMin / atomicMax on the slot (the replacement has the
// same key, so the slot's meaning never changes); keep=False counts occurrences per slot.  A
// second launch resolves each row against the finished table, so results do not depend on
// the order in which waves ran.
//
// Algorithmic bytes: K4 16*N keys + N mask + 48*U (one 24-B slot write and read per distinct
// key); K5 16*N + N + 16*R.  Bound: HBM / Infinity-Cache random access, not bandwidth.
#include "dyd_common.h"

namespace dyd {

constexpr int K4_BLOCK = 256;
constexpr long long K4_EMPTY = -1;

__device__ __forceinline__ bool key_eq(ulonglong2 a, ulonglong2 b) { return a.x == b.x && a.y == b.y; }

// mode: DYD_KEEP_FIRST / DYD_KEEP_LAST / DYD_KEEP_NONE
// Every row first finds its key's slot (claiming it if it is the first to come); the update of the slot — atomicMin /
// atomicMax of the row index, or the occurrence count — is issued afterwards, with the lanes of a wave that met in the
// same slot folded into ONE atomic (up to K4_FOLD distinct slots per wave, the rest one by one).  On ordinary tables lanes
// seldom share a slot and the fold costs a few ballots; on a constant column (or 10 M NaN) it turns 10 M atomics on one
// word into 160 k: keep=False 116 -> 2 ms.
constexpr int K4_FOLD = 4;

__global__ __launch_bounds__(K4_BLOCK) void k4_insert(const ulonglong2 *__restrict__ keys, int64_t n,
                                                      long long *tab, unsigned int *cnt, uint64_t mask,
                                                      int mode, int *err) {
    const int64_t i = (int64_t)blockIdx.x * K4_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool update = false;   // the slot needs this row folded in
    uint64_t slot = 0;
    if (i < n) {
        const ulonglong2 k = keys[i];
        slot = k.x & mask;
        bool placed = false;
        for (uint64_t probe = 0; probe <= mask; ++probe) {  // bounded: every wave exits
            long long cur = __hip_atomic_load(&tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == K4_EMPTY) {
                const unsigned long long prev =
                    atomicCAS(reinterpret_cast<unsigned long long *>(&tab[slot]),
                              (unsigned long long)K4_EMPTY, (unsigned long long)i);
                if (prev == (unsigned long long)K4_EMPTY) {
                    update = (mode == DYD_KEEP_NONE);   // the claim itself recorded the index; the count still wants it
                    placed = true;
                    break;
                }
                cur = (long long)prev;
            }
            if (key_eq(keys[cur], k)) {
                // the slot only ever moves towards the winner, so a value already at least as good as this row makes the
                // atomic a no-op: skipped (rows run roughly in index order, so with keep=first most duplicates skip it)
                update = (mode == DYD_KEEP_NONE) || (mode == DYD_KEEP_FIRST ? cur > (long long)i : cur < (long long)i);
                placed = true;
                break;
            }
            slot = (slot + 1) & mask;
        }
        if (!placed) atomicOr(err, 1);  // table full: cannot happen with capacity >= 2n
    }
    // fold the lanes that share a slot (row indices grow with the lane: the lowest / highest lane of a group holds its min / max)
    unsigned long long todo = __ballot(update);
    for (int round = 0; round < K4_FOLD && todo; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint64_t s0 = __shfl(slot, leader);
        const unsigned long long same = __ballot(update && slot == s0);
        const int last = 63 - __clzll((long long)same);
        if (mode == DYD_KEEP_FIRST) { if (lane == leader) atomicMin(&tab[slot], (long long)i); }
        else if (mode == DYD_KEEP_LAST) { if (lane == last) atomicMax(&tab[slot], (long long)i); }
        else if (lane == leader) atomicAdd(&cnt[slot], (unsigned int)__popcll(same));
        if ((same >> lane) & 1ull) update = false;
        todo &= ~same;
    }
    if (update) {
        if (mode == DYD_KEEP_FIRST) atomicMin(&tab[slot], (long long)i);
        else if (mode == DYD_KEEP_LAST) atomicMax(&tab[slot], (long long)i);
        else atomicAdd(&cnt[slot], 1u);
    }
}

__global__ __launch_bounds__(K4_BLOCK) void k4_resolve(const ulonglong2 *__restrict__ keys, int64_t first,
                                                       int64_t n_local, const long long *__restrict__ tab,
                                                       const unsigned int *__restrict__ cnt, uint64_t mask,
                                                       int mode, uint8_t *__restrict__ out_keep, int *err) {
    const int64_t t = (int64_t)blockIdx.x * K4_BLOCK + threadIdx.x;
    if (t >= n_local) return;
    const int64_t i = first + t;
    const ulonglong2 k = keys[i];
    uint64_t slot = k.x & mask;
    if (mode != DYD_KEEP_NONE) {
        // keep=first / last: the row is kept iff the table holds ITS index, and the only slot that can hold it lies on its
        // key's probe chain before the first empty slot — so the walk compares indices only and never loads another
        // row's key (the dependent second gather of every step): 10 M rows 0.79 -> 0.66 ms
        uint8_t kept = 0;
        for (uint64_t probe = 0; probe <= mask; ++probe) {
            const long long cur = tab[slot];
            if (cur == (long long)i) {
                kept = 1;
                break;
            }
            if (cur == K4_EMPTY) break;
            slot = (slot + 1) & mask;
        }
        out_keep[t] = kept;
        return;
    }
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const long long cur = tab[slot];
        if (cur == K4_EMPTY) break;
        if (key_eq(keys[cur], k)) {
            out_keep[t] = (uint8_t)(cnt[slot] == 1u);
            return;
        }
        slot = (slot + 1) & mask;
    }
    out_keep[t] = 0;
    atomicOr(err, 2);  // a key that was inserted must be found
}

__global__ __launch_bounds__(K4_BLOCK) void k5_probe(const ulonglong2 *__restrict__ keys, int64_t n,
                                                     const ulonglong2 *__restrict__ ref_keys,
                                                     const long long *__restrict__ tab, uint64_t mask,
                                                     uint8_t *__restrict__ out_mask) {
    const int64_t i = (int64_t)blockIdx.x * K4_BLOCK + threadIdx.x;
    if (i >= n) return;
    const ulonglong2 k = keys[i];
    uint64_t slot = k.x & mask;
    uint8_t found = 0;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const long long cur = tab[slot];
        if (cur == K4_EMPTY) break;
        if (key_eq(ref_keys[cur], k)) {
            found = 1;
            break;
        }
        slot = (slot + 1) & mask;
    }
    out_mask[i] = found;
}

// ---- who stands for whom (verification of hash equality) ------------------------------------------------------------------
// Equality in K4 / K5 is equality of 128-bit hashes.  To PROVE that it was equality of values the host compares the bytes of
// every row that was matched with the bytes of the row it was matched to — which needs that row's index, not just the flag:
// k4_partner walks the (keep=first) table by key and writes the FIRST row holding the same hash (the row itself for a first
// occurrence), k5_partner the reference row a main row hit (-1: no hit).  One extra gather per row; run only when asked for.
__global__ __launch_bounds__(K4_BLOCK) void k4_partner(const ulonglong2 *__restrict__ keys, int64_t n, const long long *__restrict__ tab,
                                                       uint64_t mask, long long *__restrict__ out_partner, int *err) {
    const int64_t i = (int64_t)blockIdx.x * K4_BLOCK + threadIdx.x;
    if (i >= n) return;
    const ulonglong2 k = keys[i];
    uint64_t slot = k.x & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const long long cur = tab[slot];
        if (cur == K4_EMPTY) break;
        if (key_eq(keys[cur], k)) { out_partner[i] = cur; return; }
        slot = (slot + 1) & mask;
    }
    out_partner[i] = -1;
    atomicOr(err, 2);  // a key that was inserted must be found
}

__global__ __launch_bounds__(K4_BLOCK) void k5_partner(const ulonglong2 *__restrict__ keys, int64_t n, const ulonglong2 *__restrict__ ref_keys,
                                                       const long long *__restrict__ tab, uint64_t mask, long long *__restrict__ out_partner) {
    const int64_t i = (int64_t)blockIdx.x * K4_BLOCK + threadIdx.x;
    if (i >= n) return;
    const ulonglong2 k = keys[i];
    uint64_t slot = k.x & mask;
    long long found = -1;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const long long cur = tab[slot];
        if (cur == K4_EMPTY) break;
        if (key_eq(ref_keys[cur], k)) { found = cur; break; }
        slot = (slot + 1) & mask;
    }
    out_partner[i] = found;
}

// dyd_set_option("k4_capacity_shift", k): the table is made 2^k times SMALLER than it should be — only so that a test can
// watch the failure path (a full table must surface as an error, never as a wrong mask)
static int g_k4_capacity_shift = 0;
void set_k4_capacity_shift(int v) { g_k4_capacity_shift = v < 0 ? 0 : v; }

static uint64_t table_capacity(int64_t n) {
    uint64_t cap = 1024;
    while (cap < 2 * (uint64_t)n) cap <<= 1;
    cap >>= g_k4_capacity_shift;
    return cap < 2 ? 2 : cap;
}

// scratch layout: [tab: cap x i64][cnt: cap x u32]; failures go to the context's device status word
static int dedup_launch(const uint64_t *h, int64_t n_all, int64_t first, int64_t n_local, int keep_mode,
                        uint8_t *out_keep, hipStream_t st) {
    const uint64_t cap = table_capacity(n_all);
    const size_t tab_bytes = cap * 8, cnt_bytes = (keep_mode == DYD_KEEP_NONE) ? cap * 4 : 0;
    void *scr = nullptr;
    int rc = get_scratch(tab_bytes + cnt_bytes + 16, &scr, st);
    if (rc) return rc;
    long long *tab = static_cast<long long *>(scr);
    unsigned int *cnt = reinterpret_cast<unsigned int *>(static_cast<char *>(scr) + tab_bytes);
    int *err = ctx().dev_status;
    DYD_HIP(hipMemsetAsync(tab, 0xFF, tab_bytes, st));
    if (cnt_bytes) DYD_HIP(hipMemsetAsync(cnt, 0, cnt_bytes, st));
    const ulonglong2 *keys = reinterpret_cast<const ulonglong2 *>(h);
    hipLaunchKernelGGL(k4_insert, dim3((unsigned)ceil_div(n_all, K4_BLOCK)), dim3(K4_BLOCK), 0, st, keys, n_all,
                       tab, cnt, cap - 1, keep_mode, err);
    DYD_HIP(hipGetLastError());
    if (n_local > 0) {
        hipLaunchKernelGGL(k4_resolve, dim3((unsigned)ceil_div(n_local, K4_BLOCK)), dim3(K4_BLOCK), 0, st, keys,
                           first, n_local, tab, cnt, cap - 1, keep_mode, out_keep, err);
        DYD_HIP(hipGetLastError());
    }
    release_scratch(st);
    return DYD_OK;
}

static int isin_launch(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r, uint8_t *out_mask,
                       hipStream_t st) {
    if (r == 0) {
        DYD_HIP(hipMemsetAsync(out_mask, 0, (size_t)n, st));
        return DYD_OK;
    }
    // quarter load: a probe that misses (most do) walks 1.4 slots on average instead of the 2.5 of a half-full table
    const uint64_t cap = table_capacity(2 * r);
    void *scr = nullptr;
    int rc = get_scratch(cap * 8 + 16, &scr, st);
    if (rc) return rc;
    long long *tab = static_cast<long long *>(scr);
    int *err = ctx().dev_status;
    DYD_HIP(hipMemsetAsync(tab, 0xFF, cap * 8, st));
    const ulonglong2 *rk = reinterpret_cast<const ulonglong2 *>(ref_h);
    hipLaunchKernelGGL(k4_insert, dim3((unsigned)ceil_div(r, K4_BLOCK)), dim3(K4_BLOCK), 0, st, rk, r, tab,
                       (unsigned int *)nullptr, cap - 1, DYD_KEEP_FIRST, err);
    DYD_HIP(hipGetLastError());
    hipLaunchKernelGGL(k5_probe, dim3((unsigned)ceil_div(n, K4_BLOCK)), dim3(K4_BLOCK), 0, st,
                       reinterpret_cast<const ulonglong2 *>(h), n, rk, tab, cap - 1, out_mask);
    DYD_HIP(hipGetLastError());
    release_scratch(st);
    return DYD_OK;
}

static int dedup_partner_launch(const uint64_t *h, int64_t n, long long *out_partner, hipStream_t st) {
    const uint64_t cap = table_capacity(n);
    void *scr = nullptr;
    int rc = get_scratch(cap * 8 + 16, &scr, st);
    if (rc) return rc;
    long long *tab = static_cast<long long *>(scr);
    int *err = ctx().dev_status;
    DYD_HIP(hipMemsetAsync(tab, 0xFF, cap * 8, st));
    const ulonglong2 *keys = reinterpret_cast<const ulonglong2 *>(h);
    hipLaunchKernelGGL(k4_insert, dim3((unsigned)ceil_div(n, K4_BLOCK)), dim3(K4_BLOCK), 0, st, keys, n, tab, (unsigned int *)nullptr, cap - 1,
                       DYD_KEEP_FIRST, err);
    DYD_HIP(hipGetLastError());
    hipLaunchKernelGGL(k4_partner, dim3((unsigned)ceil_div(n, K4_BLOCK)), dim3(K4_BLOCK), 0, st, keys, n, tab, cap - 1, out_partner, err);
    DYD_HIP(hipGetLastError());
    release_scratch(st);
    return DYD_OK;
}

static int isin_partner_launch(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r, long long *out_partner, hipStream_t st) {
    if (r == 0) {
        DYD_HIP(hipMemsetAsync(out_partner, 0xFF, 8 * (size_t)n, st));
        return DYD_OK;
    }
    const uint64_t cap = table_capacity(2 * r);
    void *scr = nullptr;
    int rc = get_scratch(cap * 8 + 16, &scr, st);
    if (rc) return rc;
    long long *tab = static_cast<long long *>(scr);
    int *err = ctx().dev_status;
    DYD_HIP(hipMemsetAsync(tab, 0xFF, cap * 8, st));
    const ulonglong2 *rk = reinterpret_cast<const ulonglong2 *>(ref_h);
    hipLaunchKernelGGL(k4_insert, dim3((unsigned)ceil_div(r, K4_BLOCK)), dim3(K4_BLOCK), 0, st, rk, r, tab, (unsigned int *)nullptr, cap - 1,
                       DYD_KEEP_FIRST, err);
    DYD_HIP(hipGetLastError());
    hipLaunchKernelGGL(k5_partner, dim3((unsigned)ceil_div(n, K4_BLOCK)), dim3(K4_BLOCK), 0, st, reinterpret_cast<const ulonglong2 *>(h), n, rk, tab,
                       cap - 1, out_partner);
    DYD_HIP(hipGetLastError());
    release_scratch(st);
    return DYD_OK;
}

static int check_dedup_args(const uint64_t *h, int64_t n, int keep_mode, const uint8_t *out) {
    DYD_REQUIRE(n >= 0, "n < 0");
    DYD_REQUIRE(keep_mode == DYD_KEEP_FIRST || keep_mode == DYD_KEEP_LAST || keep_mode == DYD_KEEP_NONE,
                "keep_mode must be DYD_KEEP_FIRST/LAST/NONE");
    DYD_REQUIRE(n == 0 || (h && out), "null pointer");
    DYD_REQUIRE(n < (1LL << 38), "n too large");
    DYD_REQUIRE((reinterpret_cast<uintptr_t>(h) & 15) == 0, "keys must be 16-byte aligned");
    return DYD_OK;
}

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_dedup_dev(const uint64_t *h, int64_t n, int keep_mode, uint8_t *out_keep, void *stream) {
    DYD_API_ENTER();
    int rc = check_dedup_args(h, n, keep_mode, out_keep);
    if (rc || n == 0) return rc;
    return dedup_launch(h, n, 0, n, keep_mode, out_keep, pick_stream(stream));
}

int dyd_dedup_global_dev(const uint64_t *all_h, int64_t n_all, int64_t first_global, int64_t n_local,
                         int keep_mode, uint8_t *out_keep, void *stream) {
    DYD_API_ENTER();
    int rc = check_dedup_args(all_h, n_all, keep_mode, out_keep);
    if (rc) return rc;
    DYD_REQUIRE(first_global >= 0 && n_local >= 0 && first_global + n_local <= n_all,
                "local row range outside the gathered keys");
    if (n_all == 0 || n_local == 0) return DYD_OK;
    return dedup_launch(all_h, n_all, first_global, n_local, keep_mode, out_keep, pick_stream(stream));
}

int dyd_dedup(const uint64_t *h, int64_t n, int keep_mode, uint8_t *out_keep) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0, "n < 0");
    DYD_REQUIRE(keep_mode == DYD_KEEP_FIRST || keep_mode == DYD_KEEP_LAST || keep_mode == DYD_KEEP_NONE,
                "keep_mode must be DYD_KEEP_FIRST/LAST/NONE");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(h && out_keep, "null pointer");
    DevBuf d_h, d_keep;
    int rc;
    if ((rc = d_h.alloc(16 * (size_t)n)) || (rc = d_keep.alloc((size_t)n))) return rc;
    hipStream_t st = ctx().stream;
    DYD_HIP(hipMemcpyAsync(d_h.p, h, 16 * (size_t)n, hipMemcpyHostToDevice, st));
    KernelTimer t(st);
    rc = dedup_launch(d_h.as<uint64_t>(), n, 0, n, keep_mode, d_keep.as<uint8_t>(), st);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_keep, d_keep.p, (size_t)n, hipMemcpyDeviceToHost, st));
    return take_device_status(st, "dyd_dedup");
}

int dyd_isin_dev(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r, uint8_t *out_mask, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0 && r >= 0, "negative size");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(h && out_mask && (r == 0 || ref_h), "null pointer");
    DYD_REQUIRE(((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(ref_h)) & 15) == 0,
                "keys must be 16-byte aligned");
    return isin_launch(h, n, ref_h, r, out_mask, pick_stream(stream));
}

int dyd_isin(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r, uint8_t *out_mask) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0 && r >= 0, "negative size");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(h && out_mask && (r == 0 || ref_h), "null pointer");
    DevBuf d_h, d_r, d_m;
    int rc;
    if ((rc = d_h.alloc(16 * (size_t)n)) || (rc = d_r.alloc(16 * (size_t)r)) || (rc = d_m.alloc((size_t)n)))
        return rc;
    hipStream_t st = ctx().stream;
    DYD_HIP(hipMemcpyAsync(d_h.p, h, 16 * (size_t)n, hipMemcpyHostToDevice, st));
    if (r) DYD_HIP(hipMemcpyAsync(d_r.p, ref_h, 16 * (size_t)r, hipMemcpyHostToDevice, st));
    KernelTimer t(st);
    rc = isin_launch(d_h.as<uint64_t>(), n, d_r.as<uint64_t>(), r, d_m.as<uint8_t>(), st);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_mask, d_m.p, (size_t)n, hipMemcpyDeviceToHost, st));
    return take_device_status(st, "dyd_isin");
}


// Host-pointer entries for the verification of hash equality: out_partner[i] = the first row whose hash equals row i's (i itself
// for a first occurrence) / the reference row main row i hit (-1: none).  The caller compares the bytes of the pairs.
int dyd_dedup_partner(const uint64_t *h, int64_t n, int64_t *out_partner) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(h && out_partner, "null pointer");
    DYD_REQUIRE(n < (1LL << 38), "n too large");
    DevBuf d_h, d_p;
    int rc;
    if ((rc = d_h.alloc(16 * (size_t)n)) || (rc = d_p.alloc(8 * (size_t)n))) return rc;
    hipStream_t st = ctx().stream;
    DYD_HIP(hipMemcpyAsync(d_h.p, h, 16 * (size_t)n, hipMemcpyHostToDevice, st));
    rc = dedup_partner_launch(d_h.as<uint64_t>(), n, d_p.as<long long>(), st);
    if (rc) return rc;
    DYD_HIP(hipMemcpyAsync(out_partner, d_p.p, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    return take_device_status(st, "dyd_dedup_partner");
}

int dyd_isin_partner(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r, int64_t *out_partner) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0 && r >= 0, "negative size");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(h && out_partner && (r == 0 || ref_h), "null pointer");
    DevBuf d_h, d_r, d_p;
    int rc;
    if ((rc = d_h.alloc(16 * (size_t)n)) || (rc = d_r.alloc(16 * (size_t)r)) || (rc = d_p.alloc(8 * (size_t)n))) return rc;
    hipStream_t st = ctx().stream;
    DYD_HIP(hipMemcpyAsync(d_h.p, h, 16 * (size_t)n, hipMemcpyHostToDevice, st));
    if (r) DYD_HIP(hipMemcpyAsync(d_r.p, ref_h, 16 * (size_t)r, hipMemcpyHostToDevice, st));
    rc = isin_partner_launch(d_h.as<uint64_t>(), n, d_r.as<uint64_t>(), r, d_p.as<long long>(), st);
    if (rc) return rc;
    DYD_HIP(hipMemcpyAsync(out_partner, d_p.p, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    return take_device_status(st, "dyd_isin_partner");
}

}  // extern "C"
