/* pyhelpers.c — CPython glue between a pandas object column and the flat buffers of libdyd_gfx950.so
 * (module deal_yolo_daya_amd._dydpy; built with the interpreter's own headers, see the Makefile).
 *
 * The annotation column of a DataFrame is an array of str objects of ~4 KB each.  Joining and encoding them in
 * Python (native_json.cells_to_buffers) costs more than scanning them, and a million 2 KB result strings created
 * one by one cost more than emitting them.  The functions take raw addresses (numpy `.ctypes.data`); the two oldest:
 *
 *   str_views(objs, n, ptr_out, len_out, missing_out[, n_threads])
 *       per element: a str -> the address and length of its UTF-8 form (the object's own buffer for ASCII text, its
 *       cached UTF-8 copy otherwise — no per-call copy), anything else -> missing (reference processor.py:264, :344:
 *       `not isinstance(json_str, str)`).  The views live as long as the str objects do.
 *   strs_from_utf8(text, off, n, na, objs_out, n_threads)
 *       per element with na == 0: a new str of text[off[i]:off[i+1]] stored into the object array (which must hold
 *       None everywhere).  See "The str builder" below for who allocates.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <limits.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <stdint.h>
#include <string.h>

/* pass 1 of str_views, on worker threads: elements that are compact ASCII str objects (the usual annotation cell) are viewed by reading
 * the object header — immutable memory, no Python API, the caller keeps the GIL so nothing is reassigned meanwhile; everything else is
 * left for the serial pass (todo[i] = 1).  A million cells are a million cache misses on scattered headers: 0.11 s on one thread. */
typedef struct {
    PyObject **objs;
    const char **ptr;
    int64_t *len;
    uint8_t *missing, *todo;
    int64_t lo, hi;
} views_t;

static void *views_worker(void *arg) {
    views_t *w = (views_t *)arg;
    for (int64_t i = w->lo; i < w->hi; ++i) {
        PyObject *o = w->objs[i];
        if (o != NULL && Py_TYPE(o) == &PyUnicode_Type && PyUnicode_IS_COMPACT_ASCII(o)) {
            w->ptr[i] = (const char *)(((PyASCIIObject *)o) + 1);
            w->len[i] = (int64_t)PyUnicode_GET_LENGTH(o);
            w->missing[i] = 0;
            w->todo[i] = 0;
        } else {
            w->todo[i] = 1;
        }
    }
    return NULL;
}

static PyObject *str_views(PyObject *self, PyObject *args) {
    unsigned long long a_objs, a_ptr, a_len, a_missing;
    Py_ssize_t n;
    int n_threads = 1;
    if (!PyArg_ParseTuple(args, "KnKKK|i", &a_objs, &n, &a_ptr, &a_len, &a_missing, &n_threads)) return NULL;
    PyObject **objs = (PyObject **)(uintptr_t)a_objs;
    const char **ptr = (const char **)(uintptr_t)a_ptr;
    int64_t *len = (int64_t *)(uintptr_t)a_len;
    uint8_t *missing = (uint8_t *)(uintptr_t)a_missing;
    if (n == 0) return PyLong_FromSsize_t(0);
    uint8_t *todo = (uint8_t *)PyMem_RawMalloc((size_t)n);
    if (!todo) return PyErr_NoMemory();
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (n < 65536) n_threads = 1;
    {
        views_t w[64];
        pthread_t th[64];
        int started[64];
        for (int t = 0; t < n_threads; ++t) {
            w[t].objs = objs; w[t].ptr = ptr; w[t].len = len; w[t].missing = missing; w[t].todo = todo;
            w[t].lo = (int64_t)n * t / n_threads;
            w[t].hi = (int64_t)n * (t + 1) / n_threads;
            started[t] = (t > 0) && pthread_create(&th[t], NULL, views_worker, &w[t]) == 0;
        }
        for (int t = 0; t < n_threads; ++t)
            if (!started[t]) views_worker(&w[t]);
        for (int t = 0; t < n_threads; ++t)
            if (started[t]) pthread_join(th[t], NULL);
    }
    Py_ssize_t n_str = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (!todo[i]) { ++n_str; continue; }
        PyObject *o = objs[i];
        if (o != NULL && PyUnicode_CheckExact(o)) {
            Py_ssize_t k = 0;
            const char *p = PyUnicode_AsUTF8AndSize(o, &k);
            if (p == NULL) { PyMem_RawFree(todo); return NULL; } /* lone surrogate: UnicodeEncodeError, as "".encode() would raise */
            ptr[i] = p;
            len[i] = (int64_t)k;
            missing[i] = 0;
            ++n_str;
        } else {
            ptr[i] = "";
            len[i] = 0;
            missing[i] = 1;
        }
    }
    PyMem_RawFree(todo);
    return PyLong_FromSsize_t(n_str);
}

/* all_exact_str(objs, n, n_threads): is every element of the object array an exact str?  (header reads on worker threads, the caller keeps
 * the GIL) — a column of str cells holds no missing value, which spares DataFrame.isna its per-object walk */
typedef struct {
    PyObject **objs;
    int64_t lo, hi;
    int all_str;
} allstr_t;

static void *allstr_worker(void *arg) {
    allstr_t *w = (allstr_t *)arg;
    int ok = 1;
    for (int64_t i = w->lo; i < w->hi && ok; ++i) {
        PyObject *o = w->objs[i];
        ok = (o != NULL && Py_TYPE(o) == &PyUnicode_Type);
    }
    w->all_str = ok;
    return NULL;
}

static PyObject *all_exact_str(PyObject *self, PyObject *args) {
    unsigned long long a_objs;
    Py_ssize_t n;
    int n_threads = 1;
    if (!PyArg_ParseTuple(args, "Kn|i", &a_objs, &n, &n_threads)) return NULL;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (n < 65536) n_threads = 1;
    allstr_t w[64];
    pthread_t th[64];
    int started[64];
    for (int t = 0; t < n_threads; ++t) {
        w[t].objs = (PyObject **)(uintptr_t)a_objs;
        w[t].lo = (int64_t)n * t / n_threads;
        w[t].hi = (int64_t)n * (t + 1) / n_threads;
        w[t].all_str = 1;
        started[t] = (t > 0) && pthread_create(&th[t], NULL, allstr_worker, &w[t]) == 0;
    }
    for (int t = 0; t < n_threads; ++t)
        if (!started[t]) allstr_worker(&w[t]);
    int ok = 1;
    for (int t = 0; t < n_threads; ++t) {
        if (started[t]) pthread_join(th[t], NULL);
        ok &= w[t].all_str;
    }
    return PyBool_FromLong(ok);
}

/* gather_utf8(ptr, len, off, n, out, n_threads): out[off[i] .. off[i] + len[i]) = the bytes at ptr[i] — the flat buffer K3 hashes,
 * copied by worker threads without the GIL (the views come from str_views; off is the caller's prefix sum of len) */
typedef struct {
    const char *const *ptr;
    const int64_t *len, *off;
    char *out;
    int64_t lo, hi;
} gather_t;

static void *gather_worker(void *arg) {
    gather_t *w = (gather_t *)arg;
    for (int64_t i = w->lo; i < w->hi; ++i)
        if (w->len[i]) memcpy(w->out + w->off[i], w->ptr[i], (size_t)w->len[i]);
    return NULL;
}

static PyObject *gather_utf8(PyObject *self, PyObject *args) {
    unsigned long long a_ptr, a_len, a_off, a_out;
    Py_ssize_t n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KKKnKi", &a_ptr, &a_len, &a_off, &n, &a_out, &n_threads)) return NULL;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (n < 65536) n_threads = 1;
    gather_t w[64];
    pthread_t th[64];
    int started[64];
    Py_BEGIN_ALLOW_THREADS
    for (int t = 0; t < n_threads; ++t) {
        w[t].ptr = (const char *const *)(uintptr_t)a_ptr;
        w[t].len = (const int64_t *)(uintptr_t)a_len;
        w[t].off = (const int64_t *)(uintptr_t)a_off;
        w[t].out = (char *)(uintptr_t)a_out;
        w[t].lo = (int64_t)n * t / n_threads;
        w[t].hi = (int64_t)n * (t + 1) / n_threads;
        started[t] = (t > 0) && pthread_create(&th[t], NULL, gather_worker, &w[t]) == 0;
    }
    for (int t = 0; t < n_threads; ++t)
        if (!started[t]) gather_worker(&w[t]);
    for (int t = 0; t < n_threads; ++t)
        if (started[t]) pthread_join(th[t], NULL);
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

/* =====================================================================================================================
 * Column builders of the split step (core/processor.py split_frames): 15 M records, each wanting a str of its JSON text, its
 * source row's cells, its label ... at the place the shuffle gave it.  All of them have the form
 *
 *     out[slot ? slot[i] : i] = f(idx ? idx[i] : i)           i = 0 .. n-1, on worker threads over ranges of i
 *
 * so the caller can walk the records in ROW order (idx absent or non-decreasing: sequential reads, the same source object
 * many times in a row) and scatter eight bytes to the shuffled place, instead of gathering from random places.
 * ===================================================================================================================== */
#include <stdlib.h>

#define MAXT 64
typedef void *(*worker_fn)(void *);

static int64_t min_parallel = -1;    /* >= 0: overrides every builder's "too small for threads" size (tests) */

static int clamp_threads(int n_threads, int64_t n, int64_t small) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > MAXT) n_threads = MAXT;
    if (n < (min_parallel >= 0 ? min_parallel : small)) n_threads = 1;
    if ((int64_t)n_threads > n) n_threads = n > 0 ? (int)n : 1;
    return n_threads;
}

static PyObject *set_min_parallel(PyObject *self, PyObject *args) {
    long long k;
    if (!PyArg_ParseTuple(args, "L", &k)) return NULL;
    min_parallel = (int64_t)k;
    Py_RETURN_NONE;
}

/* runs fn over `count` argument blocks of `stride` bytes: block 0 on this thread, the others on their own */
static void run_workers(worker_fn fn, void *blocks, size_t stride, int count) {
    pthread_t th[MAXT];
    int started[MAXT];
    for (int t = 1; t < count; ++t) started[t] = pthread_create(&th[t], NULL, fn, (char *)blocks + stride * (size_t)t) == 0;
    fn(blocks);
    for (int t = 1; t < count; ++t) {
        if (started[t]) pthread_join(th[t], NULL);
        else fn((char *)blocks + stride * (size_t)t);
    }
}

static inline int text_is_ascii(const unsigned char *s, int64_t n) {
    uint64_t acc = 0;
    int64_t j = 0;
    for (; j + 8 <= n; j += 8) {
        uint64_t v;
        memcpy(&v, s + j, 8);
        acc |= v;
    }
    for (; j < n; ++j) acc |= s[j];
    return !(acc & 0x8080808080808080ull);
}

/* Class of a UTF-8 text for the builders: 1 = ASCII; 2 = its widest character lies in U+0100..U+FFFF (Chinese labels and
 * separators: a 2-byte str whose characters the workers can write), *nch = its number of characters; 0 = anything else (Latin-1
 * only, or beyond the BMP, or too long): decoded by CPython on the calling thread.  The class must be exact — CPython keeps every
 * str in the narrowest form that holds it and compares forms before contents. */
static inline uint8_t text_class(const unsigned char *s, int64_t n, int32_t *nch) {
    if (text_is_ascii(s, n)) return 1;
    int64_t chars = 0;
    unsigned wide = 0, four = 0;
    for (int64_t j = 0; j < n; ++j) {
        const unsigned c = s[j];
        chars += (c & 0xC0u) != 0x80u;
        wide |= (unsigned)(c >= 0xC4u) & (unsigned)(c <= 0xEFu);
        four |= (unsigned)(c >= 0xF0u);
    }
    if (four || !wide || chars > 0x7fffffff) return 0;
    *nch = (int32_t)chars;
    return 2;
}

/* strict UTF-8 -> UCS-2 for a text of class 2 into exactly nch units: 0, or -1 for anything CPython's strict decoder would refuse
 * (or that does not come to nch characters) — the caller then lets CPython decode that text and raise what it raises */
static inline int utf8_to_ucs2(const unsigned char *s, int64_t n, Py_UCS2 *d, int64_t nch) {
    int64_t j = 0, k = 0;
    while (j < n) {
        const unsigned c = s[j];
        if (k >= nch) return -1;
        if (c < 0x80u) { d[k++] = (Py_UCS2)c; ++j; continue; }
        if (c >= 0xC2u && c <= 0xDFu) {
            if (j + 1 >= n || (s[j + 1] & 0xC0u) != 0x80u) return -1;
            d[k++] = (Py_UCS2)(((c & 0x1Fu) << 6) | (s[j + 1] & 0x3Fu));
            j += 2;
            continue;
        }
        if (c >= 0xE0u && c <= 0xEFu) {
            if (j + 2 >= n || (s[j + 1] & 0xC0u) != 0x80u || (s[j + 2] & 0xC0u) != 0x80u) return -1;
            const unsigned cp = ((c & 0x0Fu) << 12) | ((s[j + 1] & 0x3Fu) << 6) | (s[j + 2] & 0x3Fu);
            if (cp < 0x800u || (cp >= 0xD800u && cp <= 0xDFFFu)) return -1;
            d[k++] = (Py_UCS2)cp;
            j += 3;
            continue;
        }
        return -1;
    }
    return k == nch ? 0 : -1;
}

/* The str builder.  Text k is given either as a view (ptr[k], len[k]) or as base[off[k] .. off[k+1]); element i of the walk
 * takes text k = idx[i] (idx absent: k = i) and its str goes to objs[slot[i]] (slot absent: i); na[i] != 0 skips the element
 * (its slot keeps what it holds).  The output slots must be fresh (NULL / None).
 * PyUnicode_New needs the GIL, so allocating N str objects is one thread's work (55-60 ns each: for the 14 M records of a
 * 1 M-row table it IS the step) — everything else runs beside it: the calling thread allocates element after element, in walk
 * order, into a sequential array and publishes how far it got; the workers, chunk by chunk behind it, copy the text in and put
 * the object at its (scattered) place.  Non-ASCII texts are decoded by the calling thread.
 * (Tried and dropped: letting the workers allocate from malloc — CPython's free() accepts such blocks — creates 14 M small
 * strings in 0.6 s instead of 0.9 but frees them in 3.5 s instead of 1.0, and for 2 KB strings sixteen threads growing sixteen
 * malloc arenas at once were four times SLOWER than the one thread, on the GPU box's host.) */
#define VCHUNK_MAX 16384
typedef struct {
    const char *const *ptr;
    const int64_t *len;
    const char *base;
    const int64_t *off;
    const int64_t *idx, *slot;
    const uint8_t *na;
    PyObject **objs;             /* the output array */
    PyObject **seq;              /* element i's object, in walk order (== objs when there is no slot) */
    uint8_t *ascii;              /* per i: the text's class (text_class: 1 ASCII, 2 BMP, 0 other; 3 = found malformed while filling); NULL: all ASCII */
    int32_t *nch;                /* per i: characters of a class-2 text (between classification and allocation) */
    int bad_text;                /* a worker met malformed UTF-8 (that element's class is 3 now) */
    int64_t n;
    int64_t chunk;               /* elements per chunk: a power of two, small enough for every thread to get several */
    int64_t next_chunk;          /* atomic: next chunk a worker takes */
    int64_t ready;               /* atomic: elements [0, ready) are allocated */
    int mode;                    /* 0 classify, 1 fill + place */
    int abort_fill;
} vshared_t;

static inline const unsigned char *vtext(const vshared_t *w, int64_t i, int64_t *n) {
    const int64_t k = w->idx ? w->idx[i] : i;
    if (w->off) { *n = w->off[k + 1] - w->off[k]; return (const unsigned char *)w->base + w->off[k]; }
    *n = w->len[k];
    return (const unsigned char *)w->ptr[k];
}

static void *vworker(void *arg) {
    vshared_t *w = (vshared_t *)arg;
    const int64_t n_chunks = (w->n + w->chunk - 1) / w->chunk;
    for (;;) {
        const int64_t c = __atomic_fetch_add(&w->next_chunk, 1, __ATOMIC_RELAXED);
        if (c >= n_chunks) break;
        const int64_t lo = c * w->chunk, hi = (lo + w->chunk < w->n) ? lo + w->chunk : w->n;
        if (w->mode == 1) {
            while (__atomic_load_n(&w->ready, __ATOMIC_ACQUIRE) < hi) {      /* asleep, not spinning: under a CPU quota a spinning */
                if (__atomic_load_n(&w->abort_fill, __ATOMIC_RELAXED)) return NULL;   /* waiter is charged like a working thread */
                const struct timespec nap = {0, 40000};
                nanosleep(&nap, NULL);
            }
        }
        for (int64_t i = lo; i < hi; ++i) {
            if (w->na && w->na[i]) continue;
            int64_t n;
            const unsigned char *s = vtext(w, i, &n);
            if (w->mode == 0) {
                int32_t chars = 0;
                w->ascii[i] = text_class(s, n, &chars);
                if (w->nch) w->nch[i] = chars;
                else if (w->ascii[i] == 2) w->ascii[i] = 0;
                continue;
            }
            PyObject *o = w->seq[i];
            if (!w->ascii || w->ascii[i] == 1) memcpy(PyUnicode_1BYTE_DATA(o), s, (size_t)n);
            else if (w->ascii[i] == 2 && utf8_to_ucs2(s, n, PyUnicode_2BYTE_DATA(o), (int64_t)PyUnicode_GET_LENGTH(o)) != 0) {
                w->ascii[i] = 3;
                __atomic_store_n(&w->bad_text, 1, __ATOMIC_RELAXED);
            }
            if (w->slot) w->objs[w->slot[i]] = o;
        }
    }
    return NULL;
}

static void vrun_all(vshared_t *w, int n_threads, int mode) {   /* without the GIL: every thread works through the chunks */
    pthread_t th[MAXT];
    int k = 0;
    w->mode = mode;
    w->next_chunk = 0;
    for (int t = 1; t < n_threads; ++t)
        if (pthread_create(&th[k], NULL, vworker, w) == 0) ++k;
    vworker(w);
    for (int t = 0; t < k; ++t) pthread_join(th[t], NULL);
}

/* w: the texts, idx / slot / na, objs and n filled in.  Returns 0, or -1 with an exception set (nothing is left in the array
 * then). */
/* ---- fresh arenas for a known amount of small str objects --------------------------------------------------------------------
 * Creating millions of str objects is serial where it allocates (the GIL), and for the split step's records — ~300 bytes, i.e.
 * pymalloc — most of an allocation's cost is not the allocator: pymalloc maps a new 256 KiB arena for every ~800 records, and
 * every new 4 KiB page of it is a first-touch fault taken by the one thread everybody waits for.  On the GPU box's host (a VM)
 * such faults do not even scale with threads: 12 GB/s for all sixteen together, against 260 GB/s for transparent huge pages
 * (profiles/r03_first_touch.log).  Measured there for 14 M strings of 300 bytes: 0.85-0.91 s from cold arenas, 0.27-0.29 s from
 * prepared ones.
 *
 * So the builders size what they are about to allocate and prepare it first, on all cores, on huge pages: a slab is mapped,
 * advised MADV_HUGEPAGE and touched, and for the duration of the allocation loop pymalloc's ARENA allocator
 * (PyObject_SetArenaAllocator, a public hook) hands out the slab's 256 KiB slots.  Afterwards the default allocator is back and
 * the unused tail of the slab is unmapped.  pymalloc releases an arena with munmap(arena, size) whoever mapped it, so nothing of
 * ours outlives the call: the slots are ordinary arenas.  Skipped below 4 MB (DYD_PREFAULT_MIN_MB) and with DYD_PREFAULT=0; a
 * failure on the way just means the strings are allocated the ordinary way.
 *
 * (The same was built for strings beyond pymalloc's 512 bytes — the replace step's 2 KB bbox texts, which come from malloc: grow
 * the main heap's top in 16 MiB blocks, advise and touch it, free the blocks into one chunk below a kept guard block.  Alone in a
 * process it takes 1 M x 2.5 KB from 0.37 to 0.08 s; inside the real step, whose heap holds gigabytes of live and freed cells,
 * glibc spends 175 ns per malloc whatever the pages' state and the reservation only added its own 0.06-0.1 s.  Dropped.) */
#include <sys/mman.h>

#define PF_HUGE ((size_t)2 << 20)

typedef struct {
    int arena_on;
    char *map; size_t map_bytes;       /* the slab's mapping; slab = its 2 MiB-aligned part */
    char *slab; size_t slab_bytes, slab_next;
    PyObjectArenaAllocator orig;
} prefault_t;

static prefault_t *g_prefault = NULL;  /* the slab the installed arena allocator serves from (touched under the GIL only) */
static int g_prefault_off = 0;         /* set when preparing turned out slower than what it saves (see prefault_begin) */

typedef struct { char *base; size_t lo, hi; } pf_touch_t;
static void *pf_touch(void *arg) {
    pf_touch_t *w = (pf_touch_t *)arg;
    for (size_t o = w->lo; o < w->hi; o += 4096) ((volatile char *)w->base)[o] = 0;
    return NULL;
}

static void *pf_arena_alloc(void *ctx, size_t size) {
    prefault_t *pf = (prefault_t *)ctx;
    if (size && (size & 4095) == 0) {
        const size_t at = (pf->slab_next + size - 1) / size * size;        /* a slot is aligned to its size */
        if (at + size <= pf->slab_bytes) { pf->slab_next = at + size; return pf->slab + at; }
    }
    return pf->orig.alloc(pf->orig.ctx, size);
}
static void pf_arena_free(void *ctx, void *ptr, size_t size) {
    prefault_t *pf = (prefault_t *)ctx;
    pf->orig.free(pf->orig.ctx, ptr, size);                                /* munmap(ptr, size): fine for a slot of the slab too */
}

static size_t pf_min_bytes(void) {
    const char *e = getenv("DYD_PREFAULT_MIN_MB");
    return (size_t)(e ? atol(e) : 4) << 20;
}
static int pf_enabled(void) {
    const char *e = getenv("DYD_PREFAULT");
    return !(e && e[0] == '0');
}

/* what a str object of this length takes from pymalloc (pool overhead included); nothing for one beyond its 512 bytes */
static inline void pf_count(int64_t len, size_t *small) {
    const size_t bs = ((size_t)len + 49 + 15) & ~(size_t)15;               /* PyASCIIObject (48) + text + NUL, 16-byte classes */
    if (bs <= 512) *small += 4096 / ((4096 - 48) / bs);                    /* a 4 KiB pool holds (4096 - 48) / bs blocks */
}

/* GIL held (released while the pages are touched) */
static void prefault_begin(prefault_t *pf, size_t small, int n_threads) {
    memset(pf, 0, sizeof(*pf));
    if (!pf_enabled() || g_prefault_off || g_prefault || small == 0 || small < pf_min_bytes()) return;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > MAXT) n_threads = MAXT;
    const size_t bytes = ((small + small / 16 + ((size_t)1 << 20)) + PF_HUGE - 1) & ~(PF_HUGE - 1);
    char *map = (char *)mmap(NULL, bytes + PF_HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (map == MAP_FAILED) return;
    pf->map = map;
    pf->map_bytes = bytes + PF_HUGE;
    pf->slab = (char *)(((uintptr_t)map + PF_HUGE - 1) & ~(uintptr_t)(PF_HUGE - 1));
    pf->slab_bytes = bytes;
    (void)madvise(pf->slab, bytes, MADV_HUGEPAGE);
    pf_touch_t w[MAXT];
    for (int t = 0; t < n_threads; ++t) {
        w[t].base = pf->slab;
        w[t].lo = (bytes / 4096 * (size_t)t / (size_t)n_threads) * 4096;
        w[t].hi = (bytes / 4096 * (size_t)(t + 1) / (size_t)n_threads) * 4096;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    Py_BEGIN_ALLOW_THREADS
    run_workers(pf_touch, w, sizeof(w[0]), n_threads);
    Py_END_ALLOW_THREADS
    clock_gettime(CLOCK_MONOTONIC, &t1);
    /* where huge pages have to be made by compacting a fragmented machine first, touching can take longer than the faults it was
     * meant to save (seen once: 1.7 s for 640 MB on a long-running 64 GB container, against 0.03 s afterwards): below 1 GB/s on a
     * slab of 64 MB or more the preparation is switched off for the rest of the process */
    const double touch_s = (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-9;
    if (bytes >= ((size_t)64 << 20) && touch_s > (double)bytes / 1e9) g_prefault_off = 1;
    if (g_prefault) {                                                       /* somebody else got here while the GIL was away */
        munmap(pf->map, pf->map_bytes);
        pf->map = NULL;
        return;
    }
    PyObject_GetArenaAllocator(&pf->orig);
    PyObjectArenaAllocator mine = {pf, pf_arena_alloc, pf_arena_free};
    PyObject_SetArenaAllocator(&mine);
    g_prefault = pf;
    pf->arena_on = 1;
}

static void prefault_end(prefault_t *pf) {
    if (!pf->arena_on) return;
    PyObject_SetArenaAllocator(&pf->orig);
    g_prefault = NULL;
    pf->arena_on = 0;
    char *used_end = pf->slab + ((pf->slab_next + 4095) & ~(size_t)4095), *map_end = pf->map + pf->map_bytes;
    if (pf->slab > pf->map) munmap(pf->map, (size_t)(pf->slab - pf->map));
    if (map_end > used_end) munmap(used_end, (size_t)(map_end - used_end));
    pf->map = NULL;
}

static int build_strs(vshared_t *w, int n_threads, int all_ascii) {
    const int64_t n = w->n;
    if (n == 0) return 0;
    n_threads = clamp_threads(n_threads, n, 4096);
    w->chunk = VCHUNK_MAX;
    while (w->chunk > 256 && w->chunk * 4 * n_threads > n) w->chunk >>= 1;
    for (int64_t i = 0; i < n; ++i) {             /* fresh slots only; the None references they hold are released here */
        if (w->na && w->na[i]) continue;
        PyObject **q = &w->objs[w->slot ? w->slot[i] : i];
        if (*q == NULL) continue;
        if (*q != Py_None) { PyErr_SetString(PyExc_ValueError, "str builder: the output array must be freshly allocated"); return -1; }
        *q = NULL;
        Py_DECREF(Py_None);
    }
    w->seq = w->objs;
    if (w->slot) {
        w->seq = (PyObject **)PyMem_RawMalloc((size_t)n * sizeof(PyObject *));
        if (!w->seq) { PyErr_NoMemory(); return -1; }
    }
    if (!all_ascii) {
        w->ascii = (uint8_t *)PyMem_RawMalloc((size_t)n);
        w->nch = (int32_t *)PyMem_RawMalloc((size_t)n * sizeof(int32_t));
        if (!w->ascii || !w->nch) {
            if (w->slot) PyMem_RawFree(w->seq);
            PyMem_RawFree(w->ascii); PyMem_RawFree(w->nch);
            w->ascii = NULL; w->nch = NULL;
            PyErr_NoMemory();
            return -1;
        }
        Py_BEGIN_ALLOW_THREADS
        vrun_all(w, n_threads, 0);
        Py_END_ALLOW_THREADS
    }
    int64_t made = 0;
    int failed = 0;
    pthread_t th[MAXT];
    int fillers = 0;
    w->mode = 1;
    w->next_chunk = 0;
    /* Small strings (the split step's 160-byte records): a few fillers keep up with the allocating thread — a copy is cheaper
     * than an allocation; more would only nap in turns.  Large ones (the 2 KB bbox text of a row: malloc'ed, not pymalloc'ed):
     * copying beside the allocating thread slowed it down more than the overlap gained (GPU box's host, 1 M x 2 KB: 0.27 s
     * against 0.19 s for allocate-all-then-fill-with-every-thread), so those are done in two phases. */
    int64_t sample_bytes = 0, sample_n = 0;
    for (int64_t i = 0; i < n; i += (n / 1024) + 1) {
        if (w->na && w->na[i]) continue;
        int64_t k;
        (void)vtext(w, i, &k);
        sample_bytes += k;
        ++sample_n;
    }
    const int two_phases = sample_n > 0 && sample_bytes / sample_n >= 464;
    const int want_fillers = two_phases ? 0 : (n_threads - 1 < 6 ? n_threads - 1 : 6);
    prefault_t pf;
    {
        size_t small = 0;
        if (pf_enabled() && !two_phases && (n >= 65536 || pf_min_bytes() == 0))
            for (int64_t i = 0; i < n; ++i) {
                if ((w->na && w->na[i]) || (w->ascii && !w->ascii[i])) continue;
                int64_t k;
                (void)vtext(w, i, &k);
                if (w->ascii && w->ascii[i] == 2) pf_count(24 + 2 * (int64_t)w->nch[i] + 1, &small);   /* 72-byte header, 2 bytes a character */
                else pf_count(k, &small);
            }
        prefault_begin(&pf, small, n_threads);
    }
    for (int t = 0; t < want_fillers; ++t)
        if (pthread_create(&th[fillers], NULL, vworker, w) == 0) ++fillers;
    for (int64_t i = 0; i < n; ++i) {
        if (!(w->na && w->na[i])) {
            int64_t k;
            const unsigned char *s = vtext(w, i, &k);
            const int cls = w->ascii ? w->ascii[i] : 1;
            PyObject *o = cls == 1 ? PyUnicode_New((Py_ssize_t)k, 127)
                        : cls == 2 ? PyUnicode_New((Py_ssize_t)w->nch[i], 0xFFFF)
                                   : PyUnicode_DecodeUTF8((const char *)s, (Py_ssize_t)k, "strict");
            if (!o) { failed = 1; break; }
            w->seq[i] = o;
        }
        made = i + 1;
        if ((made & (w->chunk - 1)) == 0) __atomic_store_n(&w->ready, made, __ATOMIC_RELEASE);
    }
    prefault_end(&pf);
    if (failed) __atomic_store_n(&w->abort_fill, 1, __ATOMIC_RELAXED);
    else __atomic_store_n(&w->ready, n, __ATOMIC_RELEASE);
    Py_BEGIN_ALLOW_THREADS
    if (!failed && two_phases) vrun_all(w, n_threads, 1);
    else if (!failed) vworker(w);                  /* help with what is left */
    for (int t = 0; t < fillers; ++t) pthread_join(th[t], NULL);
    Py_END_ALLOW_THREADS
    if (!failed && w->bad_text) {                  /* malformed UTF-8 behind a class-2 text: CPython decodes it and raises what it raises */
        for (int64_t i = 0; i < n && !failed; ++i) {
            if ((w->na && w->na[i]) || !w->ascii || w->ascii[i] != 3) continue;
            int64_t k;
            const unsigned char *t = vtext(w, i, &k);
            PyObject *o = PyUnicode_DecodeUTF8((const char *)t, (Py_ssize_t)k, "strict");
            if (!o) { failed = 1; break; }
            Py_DECREF(w->seq[i]);                   /* (it decoded after all: that is the string then) */
            w->seq[i] = o;
            w->objs[w->slot ? w->slot[i] : i] = o;
        }
    }
    if (failed) {                                  /* hand everything back: the array is as fresh as it came */
        for (int64_t i = 0; i < made; ++i) {
            if (w->na && w->na[i]) continue;
            PyObject *o = w->seq[i];                /* (seq IS objs when there is no slot: read before the slot is cleared) */
            w->objs[w->slot ? w->slot[i] : i] = NULL;
            Py_XDECREF(o);
        }
    }
    if (w->slot) PyMem_RawFree(w->seq);
    if (w->ascii) { PyMem_RawFree(w->ascii); w->ascii = NULL; }
    if (w->nch) { PyMem_RawFree(w->nch); w->nch = NULL; }
    return failed ? -1 : 0;
}

/* The builder in two calls, for callers that know the texts' lengths long before they know where the strings go (the split step:
 * the records of a batch of rows are final as soon as the batch is parsed, their places only after K8 + K6 have seen ALL records).
 *   alloc_strs(ptr, len, n, seq, n_threads, all_ascii, ascii_out)   seq[i] = an ASCII str of len[i] characters with its text still
 *       unwritten, or — for a text that is not ASCII — the finished str; ascii_out[i] says which (may be 0 with all_ascii != 0).
 *       This is the part that needs the GIL; the caller runs it while other batches are still being parsed by native threads.
 *   fill_strs(ptr, len, n, seq, slot, out, n_threads, ascii)   writes the ASCII texts and MOVES seq[i] to out[slot[i]] (seq[i]
 *       becomes NULL; slot == 0 and out == 0: filled in place).  No Python API: runs on worker threads without the GIL. */
static PyObject *alloc_strs(PyObject *self, PyObject *args) {
    unsigned long long a_ptr, a_len, a_seq, a_ascii = 0;
    Py_ssize_t n;
    int n_threads, all_ascii = 0;
    if (!PyArg_ParseTuple(args, "KKnKi|iK", &a_ptr, &a_len, &n, &a_seq, &n_threads, &all_ascii, &a_ascii)) return NULL;
    vshared_t w;
    memset(&w, 0, sizeof(w));
    w.ptr = (const char *const *)(uintptr_t)a_ptr;
    w.len = (const int64_t *)(uintptr_t)a_len;
    w.objs = (PyObject **)(uintptr_t)a_seq;
    w.n = (int64_t)n;
    if (n == 0) Py_RETURN_NONE;
    if (!all_ascii && !a_ascii) { PyErr_SetString(PyExc_ValueError, "alloc_strs: ascii_out is needed unless all_ascii"); return NULL; }
    n_threads = clamp_threads(n_threads, (int64_t)n, 4096);
    w.chunk = VCHUNK_MAX;
    while (w.chunk > 256 && w.chunk * 4 * n_threads > (int64_t)n) w.chunk >>= 1;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (w.objs[i] == NULL) continue;
        if (w.objs[i] != Py_None) { PyErr_SetString(PyExc_ValueError, "alloc_strs: the output array must be freshly allocated"); return NULL; }
        w.objs[i] = NULL;
        Py_DECREF(Py_None);
    }
    if (!all_ascii) {
        w.ascii = (uint8_t *)(uintptr_t)a_ascii;
        w.nch = (int32_t *)PyMem_RawMalloc((size_t)n * sizeof(int32_t));
        if (!w.nch) return PyErr_NoMemory();
        Py_BEGIN_ALLOW_THREADS
        vrun_all(&w, n_threads, 0);
        Py_END_ALLOW_THREADS
    }
    prefault_t pf;
    {
        size_t small = 0;
        if (pf_enabled() && (n >= 65536 || pf_min_bytes() == 0))
            for (Py_ssize_t i = 0; i < n; ++i) {
                if (!w.ascii || w.ascii[i] == 1) pf_count(w.len[i], &small);
                else if (w.ascii[i] == 2) pf_count(24 + 2 * (int64_t)w.nch[i] + 1, &small);
            }
        prefault_begin(&pf, small, n_threads);
    }
    for (Py_ssize_t i = 0; i < n; ++i) {
        const int cls = w.ascii ? w.ascii[i] : 1;
        PyObject *o = cls == 1 ? PyUnicode_New((Py_ssize_t)w.len[i], 127)
                    : cls == 2 ? PyUnicode_New((Py_ssize_t)w.nch[i], 0xFFFF)
                               : PyUnicode_DecodeUTF8(w.ptr[i], (Py_ssize_t)w.len[i], "strict");
        if (!o) {
            prefault_end(&pf);
            for (Py_ssize_t j = 0; j < i; ++j) { Py_DECREF(w.objs[j]); w.objs[j] = NULL; }
            PyMem_RawFree(w.nch);
            return NULL;
        }
        w.objs[i] = o;
    }
    prefault_end(&pf);
    PyMem_RawFree(w.nch);
    Py_RETURN_NONE;
}

typedef struct {
    const char *const *ptr;
    const int64_t *len, *slot;
    PyObject **seq, **out;
    const uint8_t *ascii;
    int64_t lo, hi;
    int64_t bad;                 /* -1, or the first element whose class-2 text turned out malformed */
} fillw_t;

static void *fill_worker(void *arg) {
    fillw_t *w = (fillw_t *)arg;
    w->bad = -1;
    for (int64_t i = w->lo; i < w->hi; ++i) {
        PyObject *o = w->seq[i];
        if (!w->ascii || w->ascii[i] == 1) memcpy(PyUnicode_1BYTE_DATA(o), w->ptr[i], (size_t)w->len[i]);
        else if (w->ascii[i] == 2 && utf8_to_ucs2((const unsigned char *)w->ptr[i], w->len[i], PyUnicode_2BYTE_DATA(o), (int64_t)PyUnicode_GET_LENGTH(o)) != 0 && w->bad < 0)
            w->bad = i;
        if (w->out) {
            w->out[w->slot ? w->slot[i] : i] = o;
            w->seq[i] = NULL;
        }
    }
    return NULL;
}

static PyObject *fill_strs(PyObject *self, PyObject *args) {
    unsigned long long a_ptr, a_len, a_seq, a_slot, a_out, a_ascii = 0;
    Py_ssize_t n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KKnKKKi|K", &a_ptr, &a_len, &n, &a_seq, &a_slot, &a_out, &n_threads, &a_ascii)) return NULL;
    if (n == 0) Py_RETURN_NONE;
    n_threads = clamp_threads(n_threads, (int64_t)n, 4096);
    PyObject **seq = (PyObject **)(uintptr_t)a_seq, **out = (PyObject **)(uintptr_t)a_out;
    const int64_t *slot = (const int64_t *)(uintptr_t)a_slot;
    for (Py_ssize_t i = 0; i < n; ++i)
        if (seq[i] == NULL) { PyErr_SetString(PyExc_ValueError, "fill_strs: the strings were already handed on"); return NULL; }
    if (out) {                                    /* the target slots must be fresh: release the None references they hold */
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject **q = &out[slot ? slot[i] : i];
            if (*q == NULL) continue;
            if (*q != Py_None) { PyErr_SetString(PyExc_ValueError, "fill_strs: the output array must be freshly allocated"); return NULL; }
            *q = NULL;
            Py_DECREF(Py_None);
        }
    }
    fillw_t w[MAXT];
    for (int t = 0; t < n_threads; ++t) {
        w[t].ptr = (const char *const *)(uintptr_t)a_ptr;
        w[t].len = (const int64_t *)(uintptr_t)a_len;
        w[t].slot = slot;
        w[t].seq = seq;
        w[t].out = out;
        w[t].ascii = (const uint8_t *)(uintptr_t)a_ascii;
        w[t].lo = (int64_t)n * t / n_threads;
        w[t].hi = (int64_t)n * (t + 1) / n_threads;
    }
    Py_BEGIN_ALLOW_THREADS
    run_workers(fill_worker, w, sizeof(w[0]), n_threads);
    Py_END_ALLOW_THREADS
    for (int t = 0; t < n_threads; ++t)
        if (w[t].bad >= 0) {                           /* malformed UTF-8: CPython's decoder says what is wrong with it */
            PyObject *o = PyUnicode_DecodeUTF8(w[t].ptr[w[t].bad], (Py_ssize_t)w[t].len[w[t].bad], "strict");
            if (o) { Py_DECREF(o); PyErr_SetString(PyExc_ValueError, "fill_strs: a text changed between alloc_strs and fill_strs"); }
            return NULL;
        }
    Py_RETURN_NONE;
}

/* map_strs(ptr, len, idx, slot, n, out, n_threads[, all_ascii[, na]]): out[slot[i]] = str(the len[k] bytes at ptr[k]), k = idx[i];
 * all_ascii != 0: the caller vouches that every text is ASCII; na[i] != 0 skips element i */
static PyObject *map_strs(PyObject *self, PyObject *args) {
    unsigned long long a_ptr, a_len, a_idx, a_slot, a_objs, a_na = 0;
    Py_ssize_t n;
    int n_threads, all_ascii = 0;
    if (!PyArg_ParseTuple(args, "KKKKnKi|iK", &a_ptr, &a_len, &a_idx, &a_slot, &n, &a_objs, &n_threads, &all_ascii, &a_na)) return NULL;
    vshared_t w;
    memset(&w, 0, sizeof(w));
    w.ptr = (const char *const *)(uintptr_t)a_ptr;
    w.len = (const int64_t *)(uintptr_t)a_len;
    w.idx = (const int64_t *)(uintptr_t)a_idx;
    w.slot = (const int64_t *)(uintptr_t)a_slot;
    w.na = (const uint8_t *)(uintptr_t)a_na;
    w.objs = (PyObject **)(uintptr_t)a_objs;
    w.n = (int64_t)n;
    if (build_strs(&w, n_threads, all_ascii) < 0) return NULL;
    Py_RETURN_NONE;
}

/* strs_from_utf8(text, off, n, na, objs_out, n_threads): objs_out[i] = str(text[off[i] : off[i+1]]) where na[i] == 0 (na may be 0) */
static PyObject *strs_from_utf8(PyObject *self, PyObject *args) {
    unsigned long long a_text, a_off, a_na, a_objs;
    Py_ssize_t n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KKnKKi", &a_text, &a_off, &n, &a_na, &a_objs, &n_threads)) return NULL;
    vshared_t w;
    memset(&w, 0, sizeof(w));
    w.base = (const char *)(uintptr_t)a_text;
    w.off = (const int64_t *)(uintptr_t)a_off;
    w.na = (const uint8_t *)(uintptr_t)a_na;
    w.objs = (PyObject **)(uintptr_t)a_objs;
    w.n = (int64_t)n;
    if (build_strs(&w, n_threads, 0) < 0) return NULL;
    Py_RETURN_NONE;
}

/* map_objects(src, idx, slot, n, out, n_threads): out[slot[i]] = src[idx[i]] for object arrays.  The caller KEEPS the GIL (no
 * other Python thread runs) and the workers count references with atomic adds — one per RUN of equal objects, so a row-ordered
 * walk (the ~15 records of a source row follow each other) touches each source object's header once instead of bouncing its
 * cache line between cores once per record.  out must be freshly allocated (NULL / None slots). */
typedef struct {
    PyObject **src, **out;
    const int64_t *idx, *slot;
    int64_t lo, hi;
} mobj_t;

static void *mobj_worker(void *arg) {
    mobj_t *w = (mobj_t *)arg;
    PyObject *run = NULL;
    Py_ssize_t count = 0;
    for (int64_t i = w->lo; i < w->hi; ++i) {
        PyObject *o = w->src[w->idx ? w->idx[i] : i];
        if (o == NULL) o = Py_None;
        if (o != run) {
            if (run) __atomic_fetch_add(&run->ob_refcnt, count, __ATOMIC_RELAXED);
            run = o;
            count = 0;
        }
        ++count;
        w->out[w->slot ? w->slot[i] : i] = o;
    }
    if (run) __atomic_fetch_add(&run->ob_refcnt, count, __ATOMIC_RELAXED);
    return NULL;
}

/* fresh output slots: NULL, or None whose references are handed back here */
static int release_fresh_slots(PyObject **out, Py_ssize_t n, const char *who) {
    Py_ssize_t nones = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (out[i] == Py_None) ++nones;
        else if (out[i] != NULL) { PyErr_Format(PyExc_ValueError, "%s: the output array must be freshly allocated", who); return -1; }
    }
    if (nones) {
        for (Py_ssize_t i = 0; i < n; ++i) out[i] = NULL;
        Py_SET_REFCNT(Py_None, Py_REFCNT(Py_None) - nones);
    }
    return 0;
}

static PyObject *map_objects(PyObject *self, PyObject *args) {
    unsigned long long a_src, a_idx, a_slot, a_out;
    Py_ssize_t n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KKKnKi", &a_src, &a_idx, &a_slot, &n, &a_out, &n_threads)) return NULL;
    n_threads = clamp_threads(n_threads, (int64_t)n, 65536);
    if (release_fresh_slots((PyObject **)(uintptr_t)a_out, n, "map_objects") < 0) return NULL;
    mobj_t w[MAXT];
    for (int t = 0; t < n_threads; ++t) {
        w[t].src = (PyObject **)(uintptr_t)a_src;
        w[t].out = (PyObject **)(uintptr_t)a_out;
        w[t].idx = (const int64_t *)(uintptr_t)a_idx;
        w[t].slot = (const int64_t *)(uintptr_t)a_slot;
        w[t].lo = (int64_t)n * t / n_threads;
        w[t].hi = (int64_t)n * (t + 1) / n_threads;
    }
    run_workers(mobj_worker, w, sizeof(w[0]), n_threads);
    Py_RETURN_NONE;
}

/* map_small(table, m, codes, idx, slot, n, out, n_threads): out[slot[i]] = table[codes[idx[i]]] for a SMALL table of m objects
 * (the labels of the rules, the category names): sixteen threads adding to twenty reference counts one element at a time would
 * pass those cache lines around 15 M times, so every worker counts its uses per entry and adds each sum once.  codes are int32
 * in [0, m) (checked by the caller); out must be freshly allocated.  The caller keeps the GIL. */
typedef struct {
    PyObject **table, **out;
    const int32_t *codes;
    const int64_t *idx, *slot;
    int64_t m, lo, hi;
    int64_t *count;
} msmall_t;

static void *msmall_worker(void *arg) {
    msmall_t *w = (msmall_t *)arg;
    for (int64_t i = w->lo; i < w->hi; ++i) {
        const int32_t c = w->codes[w->idx ? w->idx[i] : i];
        PyObject *o = w->table[c];
        ++w->count[c];
        w->out[w->slot ? w->slot[i] : i] = o ? o : Py_None;
    }
    for (int64_t c = 0; c < w->m; ++c)
        if (w->count[c]) {
            PyObject *o = w->table[c] ? w->table[c] : Py_None;
            __atomic_fetch_add(&o->ob_refcnt, (Py_ssize_t)w->count[c], __ATOMIC_RELAXED);
        }
    return NULL;
}

static PyObject *map_small(PyObject *self, PyObject *args) {
    unsigned long long a_table, a_codes, a_idx, a_slot, a_out;
    Py_ssize_t m, n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KnKKKnKi", &a_table, &m, &a_codes, &a_idx, &a_slot, &n, &a_out, &n_threads)) return NULL;
    n_threads = clamp_threads(n_threads, (int64_t)n, 65536);
    if (release_fresh_slots((PyObject **)(uintptr_t)a_out, n, "map_small") < 0) return NULL;
    const size_t mm = (size_t)(m > 0 ? m : 1);
    int64_t *counts = (int64_t *)PyMem_RawCalloc((size_t)n_threads * mm, sizeof(int64_t));
    if (!counts) return PyErr_NoMemory();
    msmall_t w[MAXT];
    for (int t = 0; t < n_threads; ++t) {
        w[t].table = (PyObject **)(uintptr_t)a_table;
        w[t].codes = (const int32_t *)(uintptr_t)a_codes;
        w[t].idx = (const int64_t *)(uintptr_t)a_idx;
        w[t].slot = (const int64_t *)(uintptr_t)a_slot;
        w[t].out = (PyObject **)(uintptr_t)a_out;
        w[t].m = (int64_t)m;
        w[t].lo = (int64_t)n * t / n_threads;
        w[t].hi = (int64_t)n * (t + 1) / n_threads;
        w[t].count = counts + (size_t)t * mm;
    }
    run_workers(msmall_worker, w, sizeof(w[0]), n_threads);
    PyMem_RawFree(counts);
    Py_RETURN_NONE;
}

/* map_fixed(src, itemsize, idx, slot, n, out, n_threads): out[slot[i]] = src[idx[i]] for items of 1 / 2 / 4 / 8 bytes, by worker
 * threads without the GIL (numpy's take is one thread; the split step moves a dozen 15 M-element columns) */
typedef struct {
    const char *src;
    char *out;
    const int64_t *idx, *slot;
    int64_t itemsize, lo, hi;
} mfix_t;

#define MFIX_LOOP(T)                                                                          \
    {                                                                                         \
        const T *s = (const T *)w->src;                                                       \
        T *o = (T *)w->out;                                                                   \
        if (idx && slot) for (int64_t i = w->lo; i < w->hi; ++i) o[slot[i]] = s[idx[i]];      \
        else if (idx) for (int64_t i = w->lo; i < w->hi; ++i) o[i] = s[idx[i]];               \
        else if (slot) for (int64_t i = w->lo; i < w->hi; ++i) o[slot[i]] = s[i];             \
        else for (int64_t i = w->lo; i < w->hi; ++i) o[i] = s[i];                             \
    }

static void *mfix_worker(void *arg) {
    mfix_t *w = (mfix_t *)arg;
    const int64_t *idx = w->idx, *slot = w->slot;
    switch (w->itemsize) {
        case 8: MFIX_LOOP(uint64_t) break;
        case 4: MFIX_LOOP(uint32_t) break;
        case 2: MFIX_LOOP(uint16_t) break;
        default: MFIX_LOOP(uint8_t) break;
    }
    return NULL;
}

static PyObject *map_fixed(PyObject *self, PyObject *args) {
    unsigned long long a_src, a_idx, a_slot, a_out;
    Py_ssize_t itemsize, n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KnKKnKi", &a_src, &itemsize, &a_idx, &a_slot, &n, &a_out, &n_threads)) return NULL;
    if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8) { PyErr_SetString(PyExc_ValueError, "map_fixed: itemsize 1, 2, 4 or 8"); return NULL; }
    n_threads = clamp_threads(n_threads, (int64_t)n, 65536);
    mfix_t w[MAXT];
    for (int t = 0; t < n_threads; ++t) {
        w[t].src = (const char *)(uintptr_t)a_src;
        w[t].out = (char *)(uintptr_t)a_out;
        w[t].idx = (const int64_t *)(uintptr_t)a_idx;
        w[t].slot = (const int64_t *)(uintptr_t)a_slot;
        w[t].itemsize = (int64_t)itemsize;
        w[t].lo = (int64_t)n * t / n_threads;
        w[t].hi = (int64_t)n * (t + 1) / n_threads;
    }
    Py_BEGIN_ALLOW_THREADS
    run_workers(mfix_worker, w, sizeof(w[0]), n_threads);
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

/* category_slots(cat, pos, cat_off, n, out, n_threads): out[e] = cat_off[cat[e]] + pos[e] — where record e stands once the
 * categories are laid out one after the other, each in its shuffled order.  cat int32, the rest int64. */
typedef struct {
    const int32_t *cat;
    const int64_t *pos, *cat_off;
    int64_t *out;
    int64_t lo, hi;
} cslot_t;

static void *cslot_worker(void *arg) {
    cslot_t *w = (cslot_t *)arg;
    for (int64_t e = w->lo; e < w->hi; ++e) w->out[e] = w->cat_off[w->cat[e]] + w->pos[e];
    return NULL;
}

static PyObject *category_slots(PyObject *self, PyObject *args) {
    unsigned long long a_cat, a_pos, a_off, a_out;
    Py_ssize_t n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KKKnKi", &a_cat, &a_pos, &a_off, &n, &a_out, &n_threads)) return NULL;
    n_threads = clamp_threads(n_threads, (int64_t)n, 65536);
    cslot_t w[MAXT];
    for (int t = 0; t < n_threads; ++t) {
        w[t].cat = (const int32_t *)(uintptr_t)a_cat;
        w[t].pos = (const int64_t *)(uintptr_t)a_pos;
        w[t].cat_off = (const int64_t *)(uintptr_t)a_off;
        w[t].out = (int64_t *)(uintptr_t)a_out;
        w[t].lo = (int64_t)n * t / n_threads;
        w[t].hi = (int64_t)n * (t + 1) / n_threads;
    }
    Py_BEGIN_ALLOW_THREADS
    run_workers(cslot_worker, w, sizeof(w[0]), n_threads);
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

/* map_text(ptr, len, idx, slot, n, out_off, out_text, n_threads): the len[k] bytes at ptr[k], k = idx[i], copied to
 * out_text + out_off[slot[i]] — out_off is the caller's prefix sum of the lengths in OUTPUT order: one contiguous UTF-8 buffer
 * in the wanted order, for an Arrow string column laid over it */
typedef struct {
    const char *const *ptr;
    const int64_t *len, *idx, *slot, *off;
    char *out;
    int64_t lo, hi;
} mtext_t;

static void *mtext_worker(void *arg) {
    mtext_t *w = (mtext_t *)arg;
    for (int64_t i = w->lo; i < w->hi; ++i) {
        const int64_t k = w->idx ? w->idx[i] : i;
        if (w->len[k]) memcpy(w->out + w->off[w->slot ? w->slot[i] : i], w->ptr[k], (size_t)w->len[k]);
    }
    return NULL;
}

static PyObject *map_text(PyObject *self, PyObject *args) {
    unsigned long long a_ptr, a_len, a_idx, a_slot, a_off, a_out;
    Py_ssize_t n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KKKKnKKi", &a_ptr, &a_len, &a_idx, &a_slot, &n, &a_off, &a_out, &n_threads)) return NULL;
    n_threads = clamp_threads(n_threads, (int64_t)n, 65536);
    mtext_t w[MAXT];
    for (int t = 0; t < n_threads; ++t) {
        w[t].ptr = (const char *const *)(uintptr_t)a_ptr;
        w[t].len = (const int64_t *)(uintptr_t)a_len;
        w[t].idx = (const int64_t *)(uintptr_t)a_idx;
        w[t].slot = (const int64_t *)(uintptr_t)a_slot;
        w[t].off = (const int64_t *)(uintptr_t)a_off;
        w[t].out = (char *)(uintptr_t)a_out;
        w[t].lo = (int64_t)n * t / n_threads;
        w[t].hi = (int64_t)n * (t + 1) / n_threads;
    }
    Py_BEGIN_ALLOW_THREADS
    run_workers(mtext_worker, w, sizeof(w[0]), n_threads);
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"all_exact_str", all_exact_str, METH_VARARGS, "is every element of an object array an exact str"},
    {"gather_utf8", gather_utf8, METH_VARARGS, "copy (pointer, length) views into one flat buffer at given offsets"},
    {"str_views", str_views, METH_VARARGS, "UTF-8 views of the str elements of an object array"},
    {"strs_from_utf8", strs_from_utf8, METH_VARARGS, "str objects from flat UTF-8 + offsets into an object array"},
    {"alloc_strs", alloc_strs, METH_VARARGS, "str objects of given lengths, ASCII ones with their text still unwritten"},
    {"fill_strs", fill_strs, METH_VARARGS, "write the texts of alloc_strs' strings and move them to their places"},
    {"map_strs", map_strs, METH_VARARGS, "out[slot[i]] = str(text idx[i]) from (pointer, length) views"},
    {"map_objects", map_objects, METH_VARARGS, "out[slot[i]] = src[idx[i]] for object arrays, on worker threads"},
    {"map_small", map_small, METH_VARARGS, "out[slot[i]] = table[codes[idx[i]]] for a small table of objects"},
    {"map_fixed", map_fixed, METH_VARARGS, "out[slot[i]] = src[idx[i]] for 1/2/4/8-byte items, on worker threads"},
    {"map_text", map_text, METH_VARARGS, "texts given as views -> one contiguous buffer at given offsets"},
    {"category_slots", category_slots, METH_VARARGS, "out[e] = cat_off[cat[e]] + pos[e]"},
    {"set_min_parallel", set_min_parallel, METH_VARARGS, "tests: element count from which the builders use their threads (-1 = defaults)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_dydpy", "pandas object column <-> flat UTF-8 buffers", -1, methods};

PyMODINIT_FUNC PyInit__dydpy(void) { return PyModule_Create(&module); }
