// k6_split.hip — K6: train/val/test split ids per category.
//
// Replaces, per category, DataFrame.sample(frac=1, random_state=seed).reset_index(drop=True)
// followed by the iloc cuts at int(n*train_ratio) and int(n*val_ratio)
// (reference core/processor.py:796-806).  The permutation itself is numpy's legacy MT19937
// stream and is produced on the host (dyd_mt19937_permutation); the device (1) ranks every
// expanded row inside its category in row order — a stable multi-way prefix count, which is
// the order category_rows[category] is appended in at :773 — (2) inverts the permutation and
// (3) turns rank -> shuffled position -> split id.
//
// Layout in HBM: cat = E int32 (-1 = unclassified), perm = sum n_c int64, outputs split E u8 and
// pos E int64.  Algorithmic bytes per launch: 4*E + 8*E + 9*E = 21*E.  Bound: HBM.
//
// Mapping: a wave owns a tile of K6_TILE consecutive rows and keeps one running counter per
// category in LDS.  Pass A counts the tile per category; pass B is an exclusive scan over tiles
// per category (chunked: shuffle scan inside 4096-tile chunks, then the few chunk totals); pass C replays the tile 64 rows at a time: lanes
// holding the same category are found with ballots, a lane's rank is counter + popcount of the
// lower lanes of its ballot.  Categories are processed in windows of K6_CATS so any n_cat fits.
#include <vector>

#include "dyd_common.h"

namespace dyd {

constexpr int K6_BLOCK = 256;                // 4 waves
constexpr int K6_TILE = 2048;                // rows per wave tile
constexpr int K6_CATS = 1024;                // categories per window (4 KiB of LDS counters per wave)
constexpr int K6_WAVES = K6_BLOCK / kWave;

__device__ __forceinline__ unsigned long long lanemask_lt() {
    const unsigned lane = threadIdx.x & 63;
    return lane ? (~0ull >> (64 - lane)) : 0ull;
}

// pass A: hist[(c - c0) * n_tiles + tile] = number of rows of category c in the tile
__global__ __launch_bounds__(K6_BLOCK) void k6_count(const int32_t *__restrict__ cat, int64_t n, int32_t c0,
                                                     int32_t c1, int64_t n_tiles, unsigned int *__restrict__ hist) {
    __shared__ unsigned int s_cnt[K6_WAVES][K6_CATS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * K6_WAVES + wave;
    const int32_t nc = c1 - c0;
    for (int c = lane; c < nc; c += kWave) s_cnt[wave][c] = 0;
    __builtin_amdgcn_wave_barrier();
    if (tile < n_tiles) {
        const int64_t r0 = tile * K6_TILE;
        const int64_t r1 = (r0 + K6_TILE < n) ? r0 + K6_TILE : n;
        for (int64_t r = r0 + lane; r < r1; r += kWave) {
            const int32_t c = cat[r];
            if (c >= c0 && c < c1) atomicAdd(&s_cnt[wave][c - c0], 1u);
        }
    }
    __syncthreads();
    if (tile < n_tiles)
        for (int c = lane; c < nc; c += kWave) hist[(int64_t)c * n_tiles + tile] = s_cnt[wave][c];
}

// pass B1: exclusive scan of hist[c][...] inside chunks of K6_SCAN_CHUNK tiles (one workgroup per
// (chunk, category)); the chunk's total goes to chunk_tot[c][chunk]
constexpr int K6_SCAN_PER_THREAD = 16;
constexpr int K6_SCAN_CHUNK = K6_BLOCK * K6_SCAN_PER_THREAD;

__global__ __launch_bounds__(K6_BLOCK) void k6_scan_chunks(unsigned int *hist, int64_t n_tiles, int64_t n_chunks,
                                                           unsigned long long *__restrict__ chunk_tot) {
    __shared__ unsigned int s_wave[K6_WAVES];
    const int c = blockIdx.y;
    const int64_t chunk = blockIdx.x;
    unsigned int *row = hist + (int64_t)c * n_tiles;
    const int64_t first = chunk * K6_SCAN_CHUNK + (int64_t)threadIdx.x * K6_SCAN_PER_THREAD;
    unsigned int v[K6_SCAN_PER_THREAD];
    unsigned int sum = 0;
#pragma unroll
    for (int k = 0; k < K6_SCAN_PER_THREAD; ++k) {
        v[k] = (first + k < n_tiles) ? row[first + k] : 0u;
        sum += v[k];
    }
    // inclusive scan of `sum` across the wave, then across the workgroup's four waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int incl = sum;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == kWave - 1) s_wave[wave] = incl;
    __syncthreads();
    unsigned int wave_base = 0, total = 0;
    for (int w = 0; w < K6_WAVES; ++w) {
        if (w < wave) wave_base += s_wave[w];
        total += s_wave[w];
    }
    unsigned int run = wave_base + incl - sum;  // exclusive prefix of this thread inside the chunk
#pragma unroll
    for (int k = 0; k < K6_SCAN_PER_THREAD; ++k) {
        if (first + k < n_tiles) row[first + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) chunk_tot[(int64_t)c * n_chunks + chunk] = total;
}

// pass B2: exclusive scan of the chunk totals of every category (n_chunks is tiny: tiles / 4096)
__global__ void k6_scan_totals(unsigned long long *chunk_tot, int64_t n_chunks) {
    unsigned long long *row = chunk_tot + (int64_t)blockIdx.x * n_chunks;
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int64_t i = 0; i < n_chunks; ++i) {
            const unsigned long long v = row[i];
            row[i] = run;
            run += v;
        }
    }
}

// inverse permutation: inv[cat_off[c] + perm[k]] = k - cat_off[c] for k in category c's range
__global__ __launch_bounds__(K6_BLOCK) void k6_invert(const int64_t *__restrict__ perm,
                                                      const int64_t *__restrict__ cat_off, int32_t n_cat,
                                                      int64_t total, int64_t *__restrict__ inv) {
    const int64_t k = (int64_t)blockIdx.x * K6_BLOCK + threadIdx.x;
    if (k >= total) return;
    int lo = 0, hi = n_cat;  // largest c with cat_off[c] <= k
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cat_off[mid] <= k) lo = mid; else hi = mid;
    }
    const int64_t base = cat_off[lo];
    const int64_t size = cat_off[lo + 1] - base;
    const int64_t p = perm[k];
    if (p >= 0 && p < size) inv[base + p] = k - base;  // guarded: a malformed perm cannot write out of range
}

// The same into a table of 32-bit positions (n < 2^32): half the table, half of it again resident in the 256 MiB Infinity
// Cache — a random scatter runs at 86 G words/s while its target fits that cache and at 27 G words/s into a 1 GiB table,
// whatever the word size (k0_membench modes 6 / 9).  At 165 M records 4.87 -> 2.77 ms.  (Sweeping the permutation once per
// 256 MiB window of the table, so that every scatter stays inside the cache, was measured too: the extra sweeps cost
// more than the scatter gains — 6.4 ms for K6 with 256 MiB windows, 5.2 with 512 MiB, 4.8 with one sweep.)
__global__ __launch_bounds__(K6_BLOCK) void k6_invert32(const int64_t *__restrict__ perm,
                                                        const int64_t *__restrict__ cat_off, int32_t n_cat,
                                                        int64_t total, uint32_t *__restrict__ inv) {
    const int64_t k = (int64_t)blockIdx.x * K6_BLOCK + threadIdx.x;
    if (k >= total) return;
    int lo = 0, hi = n_cat;  // largest c with cat_off[c] <= k
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cat_off[mid] <= k) lo = mid; else hi = mid;
    }
    const int64_t base = cat_off[lo];
    const int64_t size = cat_off[lo + 1] - base;
    const int64_t p = perm[k];
    if (p >= 0 && p < size) inv[base + p] = (uint32_t)(k - base);  // guarded as above
}

// pass C: ranks -> positions -> split ids for the categories of the window
template <class INV>
__global__ __launch_bounds__(K6_BLOCK) void k6_assign(const int32_t *__restrict__ cat, int64_t n, int32_t c0,
                                                      int32_t c1, int64_t n_tiles,
                                                      const unsigned int *__restrict__ hist,
                                                      const unsigned long long *__restrict__ chunk_off,
                                                      int64_t n_chunks, const INV *__restrict__ inv,
                                                      const int64_t *__restrict__ cat_off,
                                                      const int64_t *__restrict__ n_train,
                                                      const int64_t *__restrict__ n_val,
                                                      const int64_t *__restrict__ rank_base, int32_t n_cat,
                                                      uint8_t *__restrict__ out_split, int64_t *__restrict__ out_pos) {
    __shared__ unsigned int s_cnt[K6_WAVES][K6_CATS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * K6_WAVES + wave;
    if (tile >= n_tiles) return;  // no workgroup barrier below: waves are independent
    const int32_t nc = c1 - c0;
    // running in-category rank at the start of this tile; rank_base (multi-GPU) = rows of the
    // category held by lower ranks
    for (int c = lane; c < nc; c += kWave)
        s_cnt[wave][c] = hist[(int64_t)c * n_tiles + tile] +
                         (unsigned int)chunk_off[(int64_t)c * n_chunks + tile / K6_SCAN_CHUNK] +
                         (rank_base ? (unsigned int)rank_base[c0 + c] : 0u);
    __builtin_amdgcn_wave_barrier();
    const int64_t r0 = tile * K6_TILE;
    const int64_t r1 = (r0 + K6_TILE < n) ? r0 + K6_TILE : n;
    const unsigned long long lt = lanemask_lt();
    for (int64_t rb = r0; rb < r1; rb += kWave) {
        const int64_t r = rb + lane;
        const int32_t c = (r < r1) ? cat[r] : -1;
        const bool mine = (c >= c0 && c < c1);
        unsigned long long todo = __ballot(mine);
        int64_t rank = -1;
        while (todo) {  // one round per distinct category among the 64 rows
            const int leader = __ffsll((long long)todo) - 1;
            const int32_t cl = __shfl(c, leader);
            const unsigned long long same = __ballot(mine && c == cl);
            const unsigned int before = s_cnt[wave][cl - c0];
            if (mine && c == cl) rank = (int64_t)before + __popcll(same & lt);
            __builtin_amdgcn_wave_barrier();
            if (lane == leader) s_cnt[wave][cl - c0] = before + (unsigned int)__popcll(same);
            __builtin_amdgcn_wave_barrier();
            todo &= ~same;
        }
        if (mine) {
            const int64_t base = cat_off[c];
            const int64_t size = cat_off[c + 1] - base;
            int64_t pos = -1;
            uint8_t sp = 255;
            if (rank < size) {  // guarded: cat_off must cover the category's rows
                pos = (int64_t)inv[base + rank];
                const int64_t a = n_train[c], b = n_val[c];
                sp = pos < a ? 0 : (pos < a + b ? 1 : 2);
            }
            out_pos[r] = pos;
            out_split[r] = sp;
        } else if (c0 == 0 && r < r1 && (c < 0 || c >= n_cat)) {   // unclassified rows ride along with the first window
            out_pos[r] = -1;
            out_split[r] = 255;
        }
    }
}

// rows whose category is outside [0, n_cat): unclassified (own launch only when there is no category at all)
__global__ __launch_bounds__(K6_BLOCK) void k6_unclassified(const int32_t *__restrict__ cat, int64_t n,
                                                            int32_t n_cat, uint8_t *__restrict__ out_split,
                                                            int64_t *__restrict__ out_pos) {
    const int64_t r = (int64_t)blockIdx.x * K6_BLOCK + threadIdx.x;
    if (r >= n) return;
    const int32_t c = cat[r];
    if (c < 0 || c >= n_cat) {
        out_split[r] = 255;
        out_pos[r] = -1;
    }
}

// dyd_set_option("k6_variant"): 1 = 32-bit inverse-permutation table (default), 0 = 64-bit (A/B)
static int g_k6_variant = 1;
void set_k6_variant(int v) { g_k6_variant = v ? 1 : 0; }

// inv_ready: the inverse permutations are already there as 32-bit positions (K8 writes them directly); then `perm` is unused
static int split_launch(const int32_t *cat, int64_t n, const int64_t *perm, const int64_t *cat_off,
                        const int64_t *n_train, const int64_t *n_val, int32_t n_cat, int64_t total,
                        const int64_t *rank_base, uint8_t *out_split, int64_t *out_pos, hipStream_t st,
                        const uint32_t *inv_ready = nullptr) {
    const int64_t n_tiles = ceil_div(n, K6_TILE);
    const int64_t blocks = ceil_div(n_tiles, K6_WAVES);
    const int32_t win = n_cat < K6_CATS ? n_cat : K6_CATS;
    const bool narrow = g_k6_variant != 0 || inv_ready;   // 32-bit inverse table (n < 2^32 is required at the entry points)
    const size_t inv_bytes = inv_ready ? 16 : (((size_t)(total > 0 ? total : 1) * (narrow ? 4 : 8)) + 15) & ~(size_t)15;
    const size_t hist_bytes = (((size_t)(win > 0 ? win : 1) * (size_t)n_tiles * 4) + 15) & ~(size_t)15;
    const int64_t n_chunks = ceil_div(n_tiles, K6_SCAN_CHUNK);
    const size_t tot_bytes = (size_t)(win > 0 ? win : 1) * (size_t)n_chunks * 8;
    void *scr = nullptr;
    int rc = get_scratch(inv_bytes + hist_bytes + tot_bytes, &scr, st);
    if (rc) return rc;
    int64_t *inv = static_cast<int64_t *>(scr);
    unsigned int *hist = reinterpret_cast<unsigned int *>(static_cast<char *>(scr) + inv_bytes);
    unsigned long long *chunk_tot =
        reinterpret_cast<unsigned long long *>(static_cast<char *>(scr) + inv_bytes + hist_bytes);
    if (n_cat == 0) {
        hipLaunchKernelGGL(k6_unclassified, dim3((unsigned)ceil_div(n, K6_BLOCK)), dim3(K6_BLOCK), 0, st, cat, n, n_cat,
                           out_split, out_pos);
        DYD_HIP(hipGetLastError());
    }
    if (inv_ready) inv = reinterpret_cast<int64_t *>(const_cast<uint32_t *>(inv_ready));
    if (total > 0 && !inv_ready) {
        DYD_HIP(hipMemsetAsync(inv, 0, inv_bytes, st));
        if (narrow)
            hipLaunchKernelGGL(k6_invert32, dim3((unsigned)ceil_div(total, K6_BLOCK)), dim3(K6_BLOCK), 0, st, perm, cat_off, n_cat,
                               total, reinterpret_cast<uint32_t *>(inv));
        else
            hipLaunchKernelGGL(k6_invert, dim3((unsigned)ceil_div(total, K6_BLOCK)), dim3(K6_BLOCK), 0, st, perm, cat_off,
                               n_cat, total, inv);
        DYD_HIP(hipGetLastError());
    }
    for (int32_t c0 = 0; c0 < n_cat; c0 += K6_CATS) {
        const int32_t c1 = (n_cat - c0 < K6_CATS) ? n_cat : c0 + K6_CATS;
        hipLaunchKernelGGL(k6_count, dim3((unsigned)blocks), dim3(K6_BLOCK), 0, st, cat, n, c0, c1, n_tiles, hist);
        DYD_HIP(hipGetLastError());
        hipLaunchKernelGGL(k6_scan_chunks, dim3((unsigned)n_chunks, (unsigned)(c1 - c0)), dim3(K6_BLOCK), 0, st, hist,
                           n_tiles, n_chunks, chunk_tot);
        DYD_HIP(hipGetLastError());
        hipLaunchKernelGGL(k6_scan_totals, dim3((unsigned)(c1 - c0)), dim3(kWave), 0, st, chunk_tot, n_chunks);
        DYD_HIP(hipGetLastError());
        if (narrow)
            hipLaunchKernelGGL(k6_assign<uint32_t>, dim3((unsigned)blocks), dim3(K6_BLOCK), 0, st, cat, n, c0, c1, n_tiles, hist,
                               chunk_tot, n_chunks, reinterpret_cast<const uint32_t *>(inv), cat_off, n_train, n_val, rank_base,
                               n_cat, out_split, out_pos);
        else
            hipLaunchKernelGGL(k6_assign<int64_t>, dim3((unsigned)blocks), dim3(K6_BLOCK), 0, st, cat, n, c0, c1, n_tiles, hist,
                               chunk_tot, n_chunks, inv, cat_off, n_train, n_val, rank_base, n_cat, out_split, out_pos);
        DYD_HIP(hipGetLastError());
    }
    release_scratch(st);
    return DYD_OK;
}

int k8_permutations(uint32_t seed, const int64_t *sizes, int n_sizes, uint32_t *const *inv32, int64_t *const *inv64,
                    int64_t *const *perm64, hipStream_t st);

// K6 with the permutations made on the device (K8): sizes / cuts are HOST arrays (a handful of numbers the caller has anyway:
// it computed the cuts from the sizes, reference :801-802), everything else stays on the device.
static int split_seeded(const int32_t *cat, int64_t n, uint32_t seed, const int64_t *sizes, const int64_t *n_train, const int64_t *n_val,
                        int32_t n_cat, const int64_t *rank_base_host, uint8_t *out_split, int64_t *out_pos, hipStream_t st) {
    std::vector<int64_t> off((size_t)n_cat + 1, 0);
    for (int32_t c = 0; c < n_cat; ++c) {
        if (sizes[c] < 0) { set_error("invalid argument: negative category size"); return DYD_ERR_INVALID; }
        off[(size_t)c + 1] = off[(size_t)c] + sizes[c];
    }
    const int64_t total = off[(size_t)n_cat];
    if (total >= (1LL << 32)) { set_error("invalid argument: more than 2^32 records"); return DYD_ERR_INVALID; }
    DevBuf d_inv, d_small;
    int rc;
    if ((rc = d_inv.alloc(4 * (size_t)(total > 0 ? total : 1), st)) || (rc = d_small.alloc(8 * (size_t)(4 * n_cat + 1), st))) return rc;
    uint32_t *inv = d_inv.as<uint32_t>();
    int64_t *d_off = d_small.as<int64_t>(), *d_tr = d_off + n_cat + 1, *d_va = d_tr + n_cat, *d_rb = d_va + n_cat;
    DYD_HIP(hipMemcpyAsync(d_off, off.data(), 8 * (size_t)(n_cat + 1), hipMemcpyHostToDevice, st));
    if (n_cat) {
        DYD_HIP(hipMemcpyAsync(d_tr, n_train, 8 * (size_t)n_cat, hipMemcpyHostToDevice, st));
        DYD_HIP(hipMemcpyAsync(d_va, n_val, 8 * (size_t)n_cat, hipMemcpyHostToDevice, st));
        if (rank_base_host) DYD_HIP(hipMemcpyAsync(d_rb, rank_base_host, 8 * (size_t)n_cat, hipMemcpyHostToDevice, st));
    }
    // big categories on the device (one shared generator stream: every category uses the same seed), small ones with the
    // sequential host loop (a few dozen kernel launches would cost more than they do)
    constexpr int64_t kHostBelow = 1 << 15;
    std::vector<int64_t> big_sizes;
    std::vector<uint32_t *> big_inv;
    std::vector<int64_t> perm;
    std::vector<uint32_t> small_inv;
    for (int32_t c = 0; c < n_cat; ++c) {
        if (sizes[c] >= kHostBelow) {
            big_sizes.push_back(sizes[c]);
            big_inv.push_back(inv + off[(size_t)c]);
        } else if (sizes[c] > 0) {
            perm.resize((size_t)sizes[c]);
            small_inv.resize((size_t)sizes[c]);
            rc = dyd_mt19937_permutation(seed, sizes[c], perm.data());
            if (rc) return rc;
            for (int64_t k = 0; k < sizes[c]; ++k) small_inv[(size_t)perm[(size_t)k]] = (uint32_t)k;
            DYD_HIP(hipMemcpyAsync(inv + off[(size_t)c], small_inv.data(), 4 * (size_t)sizes[c], hipMemcpyHostToDevice, st));
            DYD_HIP(hipStreamSynchronize(st));   // small_inv is reused
        }
    }
    if (!big_sizes.empty()) {
        rc = k8_permutations(seed, big_sizes.data(), (int)big_sizes.size(), big_inv.data(), nullptr, nullptr, st);
        if (rc) return rc;
    }
    rc = split_launch(cat, n, nullptr, d_off, d_tr, d_va, n_cat, total, rank_base_host ? d_rb : nullptr, out_split, out_pos, st, inv);
    if (rc) return rc;
    DYD_HIP(hipStreamSynchronize(st));   // the tables above are freed on return
    return DYD_OK;
}

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_split_ids_seeded_dev(const int32_t *cat, int64_t n, uint32_t seed, const int64_t *cat_sizes_host, const int64_t *n_train_host,
                             const int64_t *n_val_host, int32_t n_cat, const int64_t *cat_rank_base_host_or_null, uint8_t *out_split,
                             int64_t *out_pos, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0 && n_cat >= 0, "negative size");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(cat && out_split && out_pos, "null pointer");
    DYD_REQUIRE(n_cat == 0 || (cat_sizes_host && n_train_host && n_val_host), "null pointer");
    DYD_REQUIRE(n < (1LL << 32), "n too large");
    return split_seeded(cat, n, seed, cat_sizes_host, n_train_host, n_val_host, n_cat, cat_rank_base_host_or_null, out_split, out_pos,
                        pick_stream(stream));
}

int dyd_split_ids_seeded(const int32_t *cat, int64_t n, uint32_t seed, const int64_t *cat_sizes, const int64_t *n_train,
                         const int64_t *n_val, int32_t n_cat, uint8_t *out_split, int64_t *out_pos) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0 && n_cat >= 0, "negative size");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(cat && out_split && out_pos, "null pointer");
    DYD_REQUIRE(n_cat == 0 || (cat_sizes && n_train && n_val), "null pointer");
    DYD_REQUIRE(n < (1LL << 32), "n too large");
    DevBuf d_cat, d_split, d_pos;
    int rc;
    if ((rc = d_cat.alloc(4 * (size_t)n)) || (rc = d_split.alloc((size_t)n)) || (rc = d_pos.alloc(8 * (size_t)n))) return rc;
    hipStream_t st = ctx().stream;
    DYD_HIP(hipMemcpyAsync(d_cat.p, cat, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    KernelTimer t(st);
    rc = split_seeded(d_cat.as<int32_t>(), n, seed, cat_sizes, n_train, n_val, n_cat, nullptr, d_split.as<uint8_t>(), d_pos.as<int64_t>(), st);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_split, d_split.p, (size_t)n, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(out_pos, d_pos.p, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    return DYD_OK;
}

int dyd_split_ids_dev(const int32_t *cat, int64_t n, const int64_t *cat_perm_concat, const int64_t *cat_off,
                      const int64_t *n_train, const int64_t *n_val, int32_t n_cat, uint8_t *out_split,
                      int64_t *out_pos, void *stream) {
    // cat_off lives on the device here; its last entry (the permutation length) is needed on the
    // host to size the scratch, so this twin reads it back once (8 bytes).
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0 && n_cat >= 0, "negative size");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(cat && out_split && out_pos, "null pointer");
    DYD_REQUIRE(n_cat == 0 || (cat_perm_concat && cat_off && n_train && n_val), "null pointer");
    DYD_REQUIRE(n < (1LL << 32), "n too large");
    hipStream_t st = pick_stream(stream);
    int64_t total = 0;
    if (n_cat > 0) {
        DYD_HIP(hipMemcpyAsync(&total, cat_off + n_cat, 8, hipMemcpyDeviceToHost, st));
        DYD_HIP(hipStreamSynchronize(st));
        DYD_REQUIRE(total >= 0, "cat_off[n_cat] < 0");
    }
    return split_launch(cat, n, cat_perm_concat, cat_off, n_train, n_val, n_cat, total, nullptr, out_split, out_pos,
                        st);
}

int dyd_split_ids_sharded_dev(const int32_t *cat, int64_t n, const int64_t *cat_perm_concat, const int64_t *cat_off,
                              const int64_t *n_train, const int64_t *n_val, int32_t n_cat,
                              const int64_t *cat_rank_base, uint8_t *out_split, int64_t *out_pos, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0 && n_cat >= 0, "negative size");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(cat && out_split && out_pos, "null pointer");
    DYD_REQUIRE(n_cat == 0 || (cat_perm_concat && cat_off && n_train && n_val && cat_rank_base), "null pointer");
    DYD_REQUIRE(n < (1LL << 32), "n too large");
    hipStream_t st = pick_stream(stream);
    int64_t total = 0;
    if (n_cat > 0) {
        DYD_HIP(hipMemcpyAsync(&total, cat_off + n_cat, 8, hipMemcpyDeviceToHost, st));
        DYD_HIP(hipStreamSynchronize(st));
        DYD_REQUIRE(total >= 0, "cat_off[n_cat] < 0");
    }
    return split_launch(cat, n, cat_perm_concat, cat_off, n_train, n_val, n_cat, total, cat_rank_base, out_split,
                        out_pos, st);
}

int dyd_split_ids(const int32_t *cat, int64_t n, const int64_t *cat_perm_concat, const int64_t *cat_off,
                  const int64_t *n_train, const int64_t *n_val, int32_t n_cat, uint8_t *out_split, int64_t *out_pos) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0 && n_cat >= 0, "negative size");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(cat && out_split && out_pos, "null pointer");
    DYD_REQUIRE(n_cat == 0 || (cat_off && n_train && n_val), "null pointer");
    int64_t total = 0;
    if (n_cat > 0) {
        DYD_REQUIRE(cat_off[0] == 0, "cat_off[0] != 0");
        for (int32_t c = 0; c < n_cat; ++c) DYD_REQUIRE(cat_off[c + 1] >= cat_off[c], "cat_off not monotone");
        total = cat_off[n_cat];
        DYD_REQUIRE(total == 0 || cat_perm_concat, "cat_perm_concat is null");
    }
    DevBuf d_cat, d_perm, d_off, d_tr, d_va, d_split, d_pos;
    int rc;
    if ((rc = d_cat.alloc(4 * (size_t)n)) || (rc = d_perm.alloc(8 * (size_t)total)) ||
        (rc = d_off.alloc(8 * (size_t)(n_cat + 1))) || (rc = d_tr.alloc(8 * (size_t)n_cat)) ||
        (rc = d_va.alloc(8 * (size_t)n_cat)) || (rc = d_split.alloc((size_t)n)) || (rc = d_pos.alloc(8 * (size_t)n)))
        return rc;
    hipStream_t st = ctx().stream;
    DYD_HIP(hipMemcpyAsync(d_cat.p, cat, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    if (total) DYD_HIP(hipMemcpyAsync(d_perm.p, cat_perm_concat, 8 * (size_t)total, hipMemcpyHostToDevice, st));
    if (n_cat) {
        DYD_HIP(hipMemcpyAsync(d_off.p, cat_off, 8 * (size_t)(n_cat + 1), hipMemcpyHostToDevice, st));
        DYD_HIP(hipMemcpyAsync(d_tr.p, n_train, 8 * (size_t)n_cat, hipMemcpyHostToDevice, st));
        DYD_HIP(hipMemcpyAsync(d_va.p, n_val, 8 * (size_t)n_cat, hipMemcpyHostToDevice, st));
    }
    KernelTimer t(st);
    rc = split_launch(d_cat.as<int32_t>(), n, d_perm.as<int64_t>(), d_off.as<int64_t>(), d_tr.as<int64_t>(),
                      d_va.as<int64_t>(), n_cat, total, nullptr, d_split.as<uint8_t>(), d_pos.as<int64_t>(), st);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_split, d_split.p, (size_t)n, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(out_pos, d_pos.p, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    return DYD_OK;
}

}  // extern "C"
