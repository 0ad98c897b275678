/* spl_pyshim.c -- CPython side of splintr_amd.Tokenizer: what PyO3 does for the reference's
 * PyTokenizer (src/python/bindings.rs:254-350: extract Vec<String> from the argument, call the core,
 * convert Vec<Vec<u32>> to list[list[int]]), as ONE C call around spl_encode_batch.
 *
 *   encode_batch(handle, texts, flags)     -> list[list[int]]
 *   encode_batch_csr(handle, texts, flags) -> (ids as u32 bytes, offsets as u64 bytes)
 *   encode(handle, text, flags)            -> list[int]
 *   pack_bytes(texts) / lists_from_csr(ids, off): the two halves on their own (CPU-side tests, measurements)
 *
 * The texts are UTF-8 encoded straight into a pinned staging buffer (spl_host_alloc: the GPU's
 * DMA engine reads it without another host copy); the result lists are filled from the pinned CSR with
 * cached int objects (ints are immutable, so sharing one object per id is unobservable).
 * The GIL is held throughout, as in the reference (no allow_threads in src/python/bindings.rs): the staging
 * buffers below are process-wide and safe for exactly that reason -- no entry point releases the GIL between
 * staging a batch and copying its result out.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#include "../../include/splintr_hip.h"

static uint8_t* g_text = NULL;      /* pinned */
static size_t g_text_cap = 0;
static uint64_t* g_off = NULL;      /* pinned */
static size_t g_off_cap = 0;
static PyObject** g_ints = NULL;    /* id -> int object (lazily created, owned) */
static size_t g_ints_cap = 0;

/* g_pageable: pack_bytes() on a box without a GPU (CPU-side tests of the packing code) -- staging then comes from
 * malloc.  The encode entry points never set it: without a GPU they fail in spl_create long before. */
static int g_pageable = 0, g_text_malloced = 0, g_off_malloced = 0;
static int ensure(void** p, size_t* cap, size_t need, int* malloced) {
    if (*p && *cap >= need) return 0;
    size_t c = *cap ? *cap : (1u << 16);
    while (c < need) c += c / 2 + 4096;
    int m = 0;
    void* q = spl_host_alloc(c);
    if (!q && g_pageable) { q = malloc(c); m = 1; }
    if (!q) { PyErr_SetString(PyExc_MemoryError, "splintr_amd: pinned host allocation failed"); return -1; }
    if (*malloced) free(*p); else spl_host_free(*p);
    *p = q;
    *cap = c;
    *malloced = m;
    return 0;
}

/* UTF-8 length of a str; -1 if it holds a lone surrogate (then the standard codec raises).
 * Text that is not pure ASCII is still MOSTLY ASCII (a few accented letters, dashes, quotes per document): the
 * one- and two-byte representations are walked eight / four characters at a time while those are all ASCII. */
static Py_ssize_t utf8_size(PyObject* s) {
    const Py_ssize_t n = PyUnicode_GET_LENGTH(s);
    if (PyUnicode_IS_ASCII(s)) return n;
    const int kind = PyUnicode_KIND(s);
    const void* d = PyUnicode_DATA(s);
    Py_ssize_t t = 0, i = 0;
    if (kind == PyUnicode_1BYTE_KIND) {
        const Py_UCS1* p = (const Py_UCS1*)d;
        for (; i + 8 <= n; i += 8) {
            uint64_t w;
            memcpy(&w, p + i, 8);
            t += 8 + (Py_ssize_t)__builtin_popcountll(w & 0x8080808080808080ull);
        }
        for (; i < n; i++) t += 1 + (p[i] >> 7);
    } else if (kind == PyUnicode_2BYTE_KIND) {
        const Py_UCS2* p = (const Py_UCS2*)d;
        for (; i < n; i++) {
            if (i + 4 <= n) {
                uint64_t w;
                memcpy(&w, p + i, 8);
                if (!(w & 0xFF80FF80FF80FF80ull)) { t += 4; i += 3; continue; }
            }
            const Py_UCS2 c = p[i];
            if (c >= 0xD800 && c <= 0xDFFF) return -1;
            t += 1 + (c >= 0x80) + (c >= 0x800);
        }
    } else {
        const Py_UCS4* p = (const Py_UCS4*)d;
        for (; i < n; i++) {
            const Py_UCS4 c = p[i];
            if (c >= 0xD800 && c <= 0xDFFF) return -1;
            t += 1 + (c >= 0x80) + (c >= 0x800) + (c >= 0x10000);
        }
    }
    return t;
}

static uint8_t* put_cp(uint8_t* o, uint32_t c) {
    if (c < 0x80) { *o++ = (uint8_t)c; }
    else if (c < 0x800) { *o++ = (uint8_t)(0xC0 | (c >> 6)); *o++ = (uint8_t)(0x80 | (c & 0x3F)); }
    else if (c < 0x10000) { *o++ = (uint8_t)(0xE0 | (c >> 12)); *o++ = (uint8_t)(0x80 | ((c >> 6) & 0x3F)); *o++ = (uint8_t)(0x80 | (c & 0x3F)); }
    else { *o++ = (uint8_t)(0xF0 | (c >> 18)); *o++ = (uint8_t)(0x80 | ((c >> 12) & 0x3F)); *o++ = (uint8_t)(0x80 | ((c >> 6) & 0x3F)); *o++ = (uint8_t)(0x80 | (c & 0x3F)); }
    return o;
}

static void utf8_write(PyObject* s, uint8_t* o) {
    const Py_ssize_t n = PyUnicode_GET_LENGTH(s);
    const void* d = PyUnicode_DATA(s);
    if (PyUnicode_IS_ASCII(s)) { memcpy(o, d, (size_t)n); return; }
    const int kind = PyUnicode_KIND(s);
    Py_ssize_t i = 0;
    if (kind == PyUnicode_1BYTE_KIND) {
        const Py_UCS1* p = (const Py_UCS1*)d;
        for (; i < n; i++) {
            if (i + 8 <= n) {
                uint64_t w;
                memcpy(&w, p + i, 8);
                if (!(w & 0x8080808080808080ull)) { memcpy(o, &w, 8); o += 8; i += 7; continue; }     /* eight ASCII characters */
            }
            o = put_cp(o, p[i]);
        }
    } else if (kind == PyUnicode_2BYTE_KIND) {
        const Py_UCS2* p = (const Py_UCS2*)d;
        for (; i < n; i++) {
            if (i + 4 <= n) {
                uint64_t w;
                memcpy(&w, p + i, 8);
                if (!(w & 0xFF80FF80FF80FF80ull)) {                                                    /* four ASCII characters */
                    o[0] = (uint8_t)w; o[1] = (uint8_t)(w >> 16); o[2] = (uint8_t)(w >> 32); o[3] = (uint8_t)(w >> 48);
                    o += 4; i += 3;
                    continue;
                }
            }
            o = put_cp(o, p[i]);
        }
    } else { const Py_UCS4* p = (const Py_UCS4*)d; for (; i < n; i++) o = put_cp(o, p[i]); }
}

/* ---- packing: list[str] -> packed UTF-8 + offsets in the pinned staging buffers ------------------------------
 * Two passes over the strings (sizes, then bytes).  A batch of a few hundred thousand strings is tens of
 * megabytes scattered over the heap: one thread moves ~1 GB/s of it (a cache miss per object), so big batches
 * are packed by a handful of helper threads.  They only READ immutable str objects that the calling thread
 * keeps alive while it holds the GIL for the whole call; anything unusual (a non-str item, a str that is not in
 * canonical form, a lone surrogate) is left to the calling thread, which raises what PyO3's extraction raises. */
#include <pthread.h>
#include <unistd.h>

#define PACK_THREADS_MAX 8
#define PACK_PAR_MIN_ITEMS 8192          /* sizes in parallel from this many strings on */
#define PACK_PAR_MIN_BYTES (2u << 20)    /* bytes in parallel from this many bytes on */

typedef struct {
    PyObject** items; Py_ssize_t lo, hi;
    uint64_t* off;          /* sizes pass: off[i + 1] = size of item i;  write pass: off[i] = its place */
    uint8_t* text;
    int write;              /* 0 sizes, 1 bytes */
    int bad;                /* sizes pass: an item this thread must not judge */
} pack_job;

static void* pack_worker(void* arg) {
    pack_job* j = (pack_job*)arg;
    if (!j->write) {
        for (Py_ssize_t i = j->lo; i < j->hi; i++) {
            PyObject* s = j->items[i];
            if (!PyUnicode_Check(s) || !PyUnicode_IS_READY(s)) { j->bad = 1; return NULL; }
            const Py_ssize_t l = utf8_size(s);
            if (l < 0) { j->bad = 1; return NULL; }
            j->off[i + 1] = (uint64_t)l;
        }
    } else {
        for (Py_ssize_t i = j->lo; i < j->hi; i++) utf8_write(j->items[i], j->text + j->off[i]);
    }
    return NULL;
}

static int pack_threads(void) {
    static int n = 0;
    if (!n) {
        long c = sysconf(_SC_NPROCESSORS_ONLN);
        n = c < 2 ? 1 : c > PACK_THREADS_MAX ? PACK_THREADS_MAX : (int)c;
    }
    return n;
}

/* run `jobs` (the first one on this thread); returns 0, or -1 if a thread could not be started (then all of
 * the remaining jobs ran here) */
static void run_jobs(pack_job* jobs, int nj) {
    pthread_t th[PACK_THREADS_MAX];
    int started[PACK_THREADS_MAX] = {0};
    for (int k = 1; k < nj; k++) started[k] = pthread_create(&th[k], NULL, pack_worker, &jobs[k]) == 0;
    pack_worker(&jobs[0]);
    for (int k = 1; k < nj; k++) {
        if (started[k]) pthread_join(th[k], NULL);
        else pack_worker(&jobs[k]);
    }
}

/* Returns the document count or -1 (exception set). */
static Py_ssize_t pack_texts(PyObject* texts, const char* argname) {
    if (PyUnicode_Check(texts) || PyBytes_Check(texts)) {
        /* PyO3 refuses a bare str for Vec<String> (src/python/bindings.rs:337) */
        PyErr_SetString(PyExc_TypeError, "Can't extract `str` to `Vec`");
        return -1;
    }
    PyObject* seq = PySequence_Fast(texts, "argument 'texts': object cannot be converted to 'Sequence'");
    if (!seq) { PyErr_Clear(); PyErr_Format(PyExc_TypeError, "argument '%s': '%s' object cannot be converted to 'Sequence'", argname, Py_TYPE(texts)->tp_name); return -1; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject** items = PySequence_Fast_ITEMS(seq);
    if (ensure((void**)&g_off, &g_off_cap, ((size_t)n + 1) * 8, &g_off_malloced)) { Py_DECREF(seq); return -1; }
    const int nt = pack_threads();
    pack_job jobs[PACK_THREADS_MAX];
    /* ---- sizes ---- */
    int need_serial = 1;
    if (nt > 1 && n >= PACK_PAR_MIN_ITEMS) {
        for (int k = 0; k < nt; k++) {
            jobs[k].items = items; jobs[k].lo = n * k / nt; jobs[k].hi = n * (k + 1) / nt;
            jobs[k].off = g_off; jobs[k].text = NULL; jobs[k].write = 0; jobs[k].bad = 0;
        }
        run_jobs(jobs, nt);
        need_serial = 0;
        for (int k = 0; k < nt; k++) need_serial |= jobs[k].bad;
    }
    if (need_serial) {
        for (Py_ssize_t i = 0; i < n; i++) {
            PyObject* s = items[i];
            if (!PyUnicode_Check(s)) {
                PyErr_Format(PyExc_TypeError, "argument '%s': '%s' object cannot be converted to 'PyString'", argname, Py_TYPE(s)->tp_name);
                Py_DECREF(seq);
                return -1;
            }
            if (PyUnicode_READY(s) < 0) { Py_DECREF(seq); return -1; }
            const Py_ssize_t l = utf8_size(s);
            if (l < 0) {                         /* lone surrogate: let the codec raise UnicodeEncodeError, as PyO3's extraction does */
                PyObject* b = PyUnicode_AsUTF8String(s);
                Py_XDECREF(b);
                if (!PyErr_Occurred()) PyErr_SetString(PyExc_UnicodeEncodeError, "surrogates not allowed");
                Py_DECREF(seq);
                return -1;
            }
            g_off[i + 1] = (uint64_t)l;
        }
    }
    uint64_t total = 0;
    g_off[0] = 0;
    for (Py_ssize_t i = 0; i < n; i++) { total += g_off[i + 1]; g_off[i + 1] = total; }
    if (ensure((void**)&g_text, &g_text_cap, (size_t)total + 64, &g_text_malloced)) { Py_DECREF(seq); return -1; }
    /* ---- bytes ---- */
    if (nt > 1 && total >= PACK_PAR_MIN_BYTES && n >= 2 * nt) {
        Py_ssize_t lo = 0;
        int nj = 0;
        for (int k = 0; k < nt && lo < n; k++) {                 /* ranges of about equal BYTES */
            const uint64_t target = total / (uint64_t)nt * (uint64_t)(k + 1);
            Py_ssize_t hi = lo;
            if (k == nt - 1) hi = n;
            else { Py_ssize_t a = lo, b = n; while (a < b) { const Py_ssize_t m = a + (b - a) / 2; if (g_off[m] < target) a = m + 1; else b = m; } hi = a > lo ? a : lo + 1; if (hi > n) hi = n; }
            jobs[nj].items = items; jobs[nj].lo = lo; jobs[nj].hi = hi; jobs[nj].off = g_off; jobs[nj].text = g_text;
            jobs[nj].write = 1; jobs[nj].bad = 0;
            nj++;
            lo = hi;
        }
        run_jobs(jobs, nj);
    } else {
        for (Py_ssize_t i = 0; i < n; i++) utf8_write(items[i], g_text + g_off[i]);
    }
    Py_DECREF(seq);
    return n;
}

/* ---- results: list[list[int]] from the CSR ---------------------------------------------------------------------
 * One shared int object per id, created on first use (ints are immutable: sharing is unobservable, CPython does
 * the same for -5..256).  A list element then costs an INCREF instead of an allocation.  The cyclic collector is
 * held off while the lists are built: every 700 new lists trigger a young collection and, with hundreds of
 * thousands of lists alive, full collections that walk every element of every list built so far -- on 250 000
 * short documents that was more than half of the call (the lists hold only ints: there is nothing to collect). */
static PyObject* int_of(uint32_t id) {      /* new reference */
    if ((size_t)id >= g_ints_cap) {
        size_t c = g_ints_cap ? g_ints_cap : 4096;
        while (c <= (size_t)id) c *= 2;
        if (c > ((size_t)1 << 24)) return PyLong_FromUnsignedLong(id);     /* (ids beyond 2^24: never from this library) */
        PyObject** q = (PyObject**)realloc(g_ints, c * sizeof(PyObject*));
        if (!q) return PyErr_NoMemory();
        memset(q + g_ints_cap, 0, (c - g_ints_cap) * sizeof(PyObject*));
        g_ints = q;
        g_ints_cap = c;
    }
    PyObject* o = g_ints[id];
    if (!o) {
        o = PyLong_FromUnsignedLong(id);
        if (!o) return NULL;
        g_ints[id] = o;                      /* the cache's own reference */
    }
    Py_INCREF(o);
    return o;
}

static PyObject* list_of(const uint32_t* ids, uint64_t n) {
    PyObject* l = PyList_New((Py_ssize_t)n);
    if (!l) return NULL;
    for (uint64_t k = 0; k < n; k++) {
        PyObject* o = int_of(ids[k]);
        if (!o) { Py_DECREF(l); return NULL; }
        PyList_SET_ITEM(l, (Py_ssize_t)k, o);
    }
    return l;
}

static PyObject* lists_of(const uint32_t* ids, const uint64_t* off, Py_ssize_t n) {
    const int gc_was_on = PyGC_Disable();
    PyObject* out = PyList_New(n);
    for (Py_ssize_t d = 0; out && d < n; d++) {
        PyObject* l = list_of(ids + off[d], off[d + 1] - off[d]);
        if (!l) { Py_CLEAR(out); break; }
        PyList_SET_ITEM(out, d, l);
    }
    if (gc_was_on) PyGC_Enable();
    return out;
}

static spl_result* run(unsigned long long handle, uint64_t n_docs, unsigned flags) {
    spl_result* res = NULL;
    const int rc = spl_encode_batch((spl_tokenizer*)(uintptr_t)handle, g_text, g_off, n_docs, flags, &res);
    if (rc != 0) { PyErr_Format(PyExc_RuntimeError, "spl_encode_batch failed (%d): %s", rc, spl_last_error()); return NULL; }
    return res;
}

static PyObject* py_encode_batch(PyObject* self, PyObject* args) {
    unsigned long long handle; PyObject* texts; unsigned flags;
    if (!PyArg_ParseTuple(args, "KOI", &handle, &texts, &flags)) return NULL;
    const Py_ssize_t n = pack_texts(texts, "texts");
    if (n < 0) return NULL;
    spl_result* res = run(handle, (uint64_t)n, flags);
    if (!res) return NULL;
    PyObject* out = lists_of(spl_result_tokens(res), spl_result_offsets(res), n);
    spl_result_free(res);
    return out;
}

/* The same call with the CSR itself as the result: (ids as bytes of u32, offsets as bytes of u64).  Staging,
 * encode and the copy out of the pinned result all happen inside this one call, with the GIL held: two
 * threads can never see each other's staging buffers (the earlier pack() + ctypes pair could). */
static PyObject* py_encode_batch_csr(PyObject* self, PyObject* args) {
    unsigned long long handle; PyObject* texts; unsigned flags;
    if (!PyArg_ParseTuple(args, "KOI", &handle, &texts, &flags)) return NULL;
    const Py_ssize_t n = pack_texts(texts, "texts");
    if (n < 0) return NULL;
    spl_result* res = run(handle, (uint64_t)n, flags);
    if (!res) return NULL;
    const uint64_t nt = spl_result_n_tokens(res);
    PyObject* ids = PyBytes_FromStringAndSize((const char*)spl_result_tokens(res), (Py_ssize_t)(nt * 4));
    PyObject* off = PyBytes_FromStringAndSize((const char*)spl_result_offsets(res), (Py_ssize_t)(((uint64_t)n + 1) * 8));
    spl_result_free(res);
    if (!ids || !off) { Py_XDECREF(ids); Py_XDECREF(off); return NULL; }
    PyObject* out = PyTuple_Pack(2, ids, off);
    Py_DECREF(ids); Py_DECREF(off);
    return out;
}

static PyObject* py_encode(PyObject* self, PyObject* args) {
    unsigned long long handle; PyObject* text; unsigned flags;
    if (!PyArg_ParseTuple(args, "KOI", &handle, &text, &flags)) return NULL;
    if (!PyUnicode_Check(text)) {
        PyErr_Format(PyExc_TypeError, "argument 'text': '%s' object cannot be converted to 'PyString'", Py_TYPE(text)->tp_name);
        return NULL;
    }
    PyObject* one = PyTuple_Pack(1, text);
    if (!one) return NULL;
    const Py_ssize_t n = pack_texts(one, "text");
    Py_DECREF(one);
    if (n < 0) return NULL;
    spl_result* res = run(handle, 1, flags);
    if (!res) return NULL;
    PyObject* l = list_of(spl_result_tokens(res), spl_result_n_tokens(res));
    spl_result_free(res);
    return l;
}

/* The two halves of the surface on their own, for tests and measurements that have no GPU:
 * pack_bytes(texts) -> (bytes, offsets bytes) as the staging buffers hold them; lists_from_csr(ids, off). */
static PyObject* py_pack_bytes(PyObject* self, PyObject* args) {
    PyObject* texts;
    if (!PyArg_ParseTuple(args, "O", &texts)) return NULL;
    g_pageable = spl_device_count() == 0;
    const Py_ssize_t n = pack_texts(texts, "texts");
    g_pageable = 0;
    if (n < 0) return NULL;
    return Py_BuildValue("y#y#", (const char*)g_text, (Py_ssize_t)g_off[n], (const char*)g_off, (Py_ssize_t)((n + 1) * 8));
}

static PyObject* py_lists_from_csr(PyObject* self, PyObject* args) {
    Py_buffer ids, off;
    if (!PyArg_ParseTuple(args, "y*y*", &ids, &off)) return NULL;
    PyObject* out = NULL;
    if (off.len < 8 || off.len % 8 || ids.len % 4) PyErr_SetString(PyExc_ValueError, "lists_from_csr: u32 ids, u64 offsets[N + 1]");
    else {
        const Py_ssize_t n = off.len / 8 - 1;
        const uint64_t* o = (const uint64_t*)off.buf;
        int ok = o[0] == 0 && o[n] * 4 == (uint64_t)ids.len;
        for (Py_ssize_t d = 0; ok && d < n; d++) ok = o[d] <= o[d + 1];
        if (!ok) PyErr_SetString(PyExc_ValueError, "lists_from_csr: offsets do not describe the ids");
        else out = lists_of((const uint32_t*)ids.buf, o, n);
    }
    PyBuffer_Release(&ids); PyBuffer_Release(&off);
    return out;
}

static PyMethodDef methods[] = {
    {"encode_batch", py_encode_batch, METH_VARARGS, "encode_batch(handle, texts, flags) -> list[list[int]]"},
    {"encode", py_encode, METH_VARARGS, "encode(handle, text, flags) -> list[int]"},
    {"encode_batch_csr", py_encode_batch_csr, METH_VARARGS, "encode_batch_csr(handle, texts, flags) -> (ids u32 bytes, offsets u64 bytes)"},
    {"pack_bytes", py_pack_bytes, METH_VARARGS, "pack_bytes(texts) -> (utf8 bytes, offsets u64 bytes)"},
    {"lists_from_csr", py_lists_from_csr, METH_VARARGS, "lists_from_csr(ids u32 bytes, offsets u64 bytes) -> list[list[int]]"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef mod = {PyModuleDef_HEAD_INIT, "_spl_py", "CPython front end of libsplintr_hip", -1, methods};

/* The lists of one large call are tens of megabytes of small heap blocks (C3: 10 000 lists, 12.5 M pointers, 100 MB), allocated and freed per
 * call.  With glibc's defaults the heap top goes back to the kernel on every free beyond 128 KB and is faulted in again by the next call: the
 * SAME list building measured 21.8 ms in one process state and 14.5 ms in another (whether some earlier free of a large block had happened to
 * raise the dynamic thresholds) -- BENCH_r04 1 600 MB/s against BENCH_r05 1 001 MB/s on the C3 batch with an unchanged shim
 * (profiles/r06_surface_bisect.txt).  The thresholds are therefore set once, at import: freed heap stays with the process (up to 1 GiB at the
 * top) and blocks of up to 32 MiB come from the heap.  SPLINTR_KEEP_MALLOC_DEFAULTS=1 leaves the allocator alone. */
#if defined(__GLIBC__)
#include <malloc.h>
#include <stdlib.h>
static void tune_malloc(void) {
    const char* e = getenv("SPLINTR_KEEP_MALLOC_DEFAULTS");
    if (e && e[0] == '1') return;
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 16 << 20);
}
#else
static void tune_malloc(void) {}
#endif

PyMODINIT_FUNC PyInit__spl_py(void) { tune_malloc(); return PyModule_Create(&mod); }
