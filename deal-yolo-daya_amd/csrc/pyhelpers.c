/* pyhelpers.c — CPython glue between a pandas object column and the flat buffers of libdyd_gfx950.so
 * (module deal_yolo_daya_amd._dydpy; built with the interpreter's own headers, see the Makefile).
 *
 * The annotation column of a DataFrame is an array of str objects of ~4 KB each.  Joining and encoding them in
 * Python (native_json.cells_to_buffers) costs more than scanning them, and a million 2 KB result strings created
 * one by one cost more than emitting them.  Two functions, both taking raw addresses (numpy `.ctypes.data`):
 *
 *   str_views(objs, n, ptr_out, len_out, missing_out[, n_threads])
 *       per element: a str -> the address and length of its UTF-8 form (the object's own buffer for ASCII text, its
 *       cached UTF-8 copy otherwise — no per-call copy), anything else -> missing (reference processor.py:264, :344:
 *       `not isinstance(json_str, str)`).  The views live as long as the str objects do.
 *   strs_from_utf8(text, off, n, na, objs_out, n_threads)
 *       per element with na == 0: a new str of text[off[i]:off[i+1]] stored into the object array (which must hold
 *       None everywhere).  ASCII cells are allocated with the GIL held and filled by worker threads without it.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
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

typedef struct {
    const char *text;
    const int64_t *off;
    const uint8_t *na;
    PyObject **objs;
    uint8_t *ascii; /* per cell: 1 = pure ASCII */
    int64_t lo, hi;
    int phase;
} work_t;

static void *worker(void *arg) {
    work_t *w = (work_t *)arg;
    if (w->phase == 0) { /* classify */
        for (int64_t i = w->lo; i < w->hi; ++i) {
            if (w->na && w->na[i]) { w->ascii[i] = 0; continue; }
            const unsigned char *s = (const unsigned char *)w->text + w->off[i];
            const int64_t k = w->off[i + 1] - w->off[i];
            uint64_t acc = 0;
            int64_t j = 0;
            for (; j + 8 <= k; j += 8) {
                uint64_t v;
                memcpy(&v, s + j, 8);
                acc |= v;
            }
            for (; j < k; ++j) acc |= s[j];
            w->ascii[i] = (acc & 0x8080808080808080ull) ? 0 : 1;
        }
    } else { /* fill the ASCII objects */
        for (int64_t i = w->lo; i < w->hi; ++i) {
            if (!w->ascii[i]) continue;
            memcpy(PyUnicode_1BYTE_DATA(w->objs[i]), w->text + w->off[i], (size_t)(w->off[i + 1] - w->off[i]));
        }
    }
    return NULL;
}

static void run_phase(work_t *proto, int64_t n, int n_threads, int phase) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (n < 4096) n_threads = 1;
    pthread_t th[64];
    work_t w[64];
    int started[64];
    for (int t = 0; t < n_threads; ++t) {
        w[t] = *proto;
        w[t].lo = n * t / n_threads;
        w[t].hi = n * (t + 1) / n_threads;
        w[t].phase = phase;
        started[t] = (t > 0) && pthread_create(&th[t], NULL, worker, &w[t]) == 0;
    }
    for (int t = 0; t < n_threads; ++t)
        if (!started[t]) worker(&w[t]);
    for (int t = 0; t < n_threads; ++t)
        if (started[t]) pthread_join(th[t], NULL);
}

static PyObject *strs_from_utf8(PyObject *self, PyObject *args) {
    unsigned long long a_text, a_off, a_na, a_objs;
    Py_ssize_t n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "KKnKKi", &a_text, &a_off, &n, &a_na, &a_objs, &n_threads)) return NULL;
    work_t w;
    memset(&w, 0, sizeof(w));
    w.text = (const char *)(uintptr_t)a_text;
    w.off = (const int64_t *)(uintptr_t)a_off;
    w.na = (const uint8_t *)(uintptr_t)a_na;
    w.objs = (PyObject **)(uintptr_t)a_objs;
    if (n == 0) Py_RETURN_NONE;
    w.ascii = (uint8_t *)PyMem_RawMalloc((size_t)n);
    if (!w.ascii) return PyErr_NoMemory();
    Py_BEGIN_ALLOW_THREADS
    run_phase(&w, (int64_t)n, n_threads, 0);
    Py_END_ALLOW_THREADS
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (w.na && w.na[i]) continue;
        const int64_t k = w.off[i + 1] - w.off[i];
        PyObject *s = w.ascii[i] ? PyUnicode_New((Py_ssize_t)k, 127)
                                 : PyUnicode_DecodeUTF8(w.text + w.off[i], (Py_ssize_t)k, "strict");
        if (!s) {
            /* objects created so far stay in the array (owned by it); the ASCII ones among them are still unfilled, so
             * fill them before reporting */
            for (Py_ssize_t j = i; j < n; ++j) w.ascii[j] = 0;
            run_phase(&w, (int64_t)n, 1, 1);
            PyMem_RawFree(w.ascii);
            return NULL;
        }
        PyObject *old = w.objs[i];
        w.objs[i] = s;
        Py_XDECREF(old);
    }
    Py_BEGIN_ALLOW_THREADS
    run_phase(&w, (int64_t)n, n_threads, 1);
    Py_END_ALLOW_THREADS
    PyMem_RawFree(w.ascii);
    Py_RETURN_NONE;
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

static PyMethodDef methods[] = {
    {"all_exact_str", all_exact_str, METH_VARARGS, "is every element of an object array an exact str"},
    {"gather_utf8", gather_utf8, METH_VARARGS, "copy (pointer, length) views into one flat buffer at given offsets"},
    {"str_views", str_views, METH_VARARGS, "UTF-8 views of the str elements of an object array"},
    {"strs_from_utf8", strs_from_utf8, METH_VARARGS, "str objects from flat UTF-8 + offsets into an object array"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_dydpy", "pandas object column <-> flat UTF-8 buffers", -1, methods};

PyMODINIT_FUNC PyInit__dydpy(void) { return PyModule_Create(&module); }
