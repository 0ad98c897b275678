/* spl_pyshim.c -- CPython side of splintr_amd.Tokenizer: what PyO3 does for the reference's
 * PyTokenizer (src/python/bindings.rs:254-350: extract Vec<String> from the argument, call the core,
 * convert Vec<Vec<u32>> to list[list[int]]), as ONE C call around spl_encode_batch.
 *
 *   encode_batch(handle, texts, flags) -> list[list[int]]
 *   encode(handle, text, flags)        -> list[int]
 *   pack(texts)                        -> (text_addr, n_bytes, off_addr, n_docs)   pinned, valid until the next call
 *
 * The texts are UTF-8 encoded straight into a pinned staging buffer (spl_host_alloc: the GPU's
 * DMA engine reads it without another host copy); the result lists are filled from the pinned CSR,
 * frequent ids with cached int objects (ints are immutable, so sharing one object per id is unobservable).
 * The GIL is held throughout, as in the reference (no allow_threads in src/python/bindings.rs).
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

static int ensure(void** p, size_t* cap, size_t need) {
    if (*p && *cap >= need) return 0;
    size_t c = *cap ? *cap : (1u << 16);
    while (c < need) c += c / 2 + 4096;
    void* q = spl_host_alloc(c);
    if (!q) { PyErr_SetString(PyExc_MemoryError, "splintr_amd: pinned host allocation failed"); return -1; }
    spl_host_free(*p);
    *p = q;
    *cap = c;
    return 0;
}

/* UTF-8 length of a str; -1 if it holds a lone surrogate (then the standard codec raises) */
static Py_ssize_t utf8_size(PyObject* s) {
    const Py_ssize_t n = PyUnicode_GET_LENGTH(s);
    if (PyUnicode_IS_ASCII(s)) return n;
    const int kind = PyUnicode_KIND(s);
    const void* d = PyUnicode_DATA(s);
    Py_ssize_t t = 0;
    if (kind == PyUnicode_1BYTE_KIND) {
        const Py_UCS1* p = (const Py_UCS1*)d;
        for (Py_ssize_t i = 0; i < n; i++) t += 1 + (p[i] >> 7);
    } else if (kind == PyUnicode_2BYTE_KIND) {
        const Py_UCS2* p = (const Py_UCS2*)d;
        for (Py_ssize_t i = 0; i < n; i++) {
            const Py_UCS2 c = p[i];
            if (c >= 0xD800 && c <= 0xDFFF) return -1;
            t += 1 + (c >= 0x80) + (c >= 0x800);
        }
    } else {
        const Py_UCS4* p = (const Py_UCS4*)d;
        for (Py_ssize_t i = 0; i < n; i++) {
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
    if (kind == PyUnicode_1BYTE_KIND) { const Py_UCS1* p = (const Py_UCS1*)d; for (Py_ssize_t i = 0; i < n; i++) o = put_cp(o, p[i]); }
    else if (kind == PyUnicode_2BYTE_KIND) { const Py_UCS2* p = (const Py_UCS2*)d; for (Py_ssize_t i = 0; i < n; i++) o = put_cp(o, p[i]); }
    else { const Py_UCS4* p = (const Py_UCS4*)d; for (Py_ssize_t i = 0; i < n; i++) o = put_cp(o, p[i]); }
}

/* list[str] -> packed UTF-8 + offsets in the pinned staging buffers.  Returns the document count or -1. */
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
    if (ensure((void**)&g_off, &g_off_cap, ((size_t)n + 1) * 8)) { Py_DECREF(seq); return -1; }
    uint64_t total = 0;
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
        g_off[i] = total;
        total += (uint64_t)l;
    }
    g_off[n] = total;
    if (ensure((void**)&g_text, &g_text_cap, (size_t)total + 64)) { Py_DECREF(seq); return -1; }
    for (Py_ssize_t i = 0; i < n; i++) utf8_write(items[i], g_text + g_off[i]);
    Py_DECREF(seq);
    return n;
}

/* Token ids are ranks, and ranks follow frequency: the low ids are the tokens a text is mostly made of.
 * Those get ONE shared int object each (a cache hit is an INCREF on a line that is hot in L2); a rare,
 * high id gets a fresh object -- sharing would turn every use into a cache miss on a cold object. */
#define INT_CACHE_MAX (1u << 16)
static PyObject* int_of(uint32_t id) {      /* new reference */
    if (id >= INT_CACHE_MAX) return PyLong_FromUnsignedLong(id);
    if ((size_t)id >= g_ints_cap) {
        size_t c = g_ints_cap ? g_ints_cap : 4096;
        while (c <= (size_t)id) c *= 2;
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
    const uint32_t* ids = spl_result_tokens(res);
    const uint64_t* off = spl_result_offsets(res);
    PyObject* out = PyList_New(n);
    for (Py_ssize_t d = 0; out && d < n; d++) {
        PyObject* l = list_of(ids + off[d], off[d + 1] - off[d]);
        if (!l) { Py_CLEAR(out); break; }
        PyList_SET_ITEM(out, d, l);
    }
    spl_result_free(res);
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

static PyObject* py_pack(PyObject* self, PyObject* args) {
    PyObject* texts;
    if (!PyArg_ParseTuple(args, "O", &texts)) return NULL;
    const Py_ssize_t n = pack_texts(texts, "texts");
    if (n < 0) return NULL;
    return Py_BuildValue("KKKn", (unsigned long long)(uintptr_t)g_text, (unsigned long long)g_off[n],
                         (unsigned long long)(uintptr_t)g_off, n);
}

static PyMethodDef methods[] = {
    {"encode_batch", py_encode_batch, METH_VARARGS, "encode_batch(handle, texts, flags) -> list[list[int]]"},
    {"encode", py_encode, METH_VARARGS, "encode(handle, text, flags) -> list[int]"},
    {"pack", py_pack, METH_VARARGS, "pack(texts) -> (text_addr, n_bytes, off_addr, n_docs) in pinned staging"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef mod = {PyModuleDef_HEAD_INIT, "_spl_py", "CPython front end of libsplintr_hip", -1, methods};

PyMODINIT_FUNC PyInit__spl_py(void) { return PyModule_Create(&mod); }
