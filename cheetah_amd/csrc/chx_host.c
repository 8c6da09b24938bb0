/* chx_host.c — CPython extension `cheetah_amd._chxhost`: the innermost host step of a merged `Segment.track`
 * (cheetah/accelerator/segment.py:545-574 for a run of scalar-setting elements) without ctypes and without Python-level
 * argument shuffling.
 *
 * A merged track of a control-loop lattice is ONE C call into libchx (chx_run_track: two launches, ~12 us of GPU at 1e6
 * particles, ~5 us at 1e4) — the Python around it cost more than the kernels: fifteen ctypes conversions, two torch.empty_like
 * calls through the Python dispatcher, stream and pointer look-ups. This module does that sequence in C against the CPython API:
 * it holds the plan (the packed kinds / parameter-pointer arrays of segment._FastRun) in a capsule and calls chx_run_track
 * through a function pointer handed over by the ctypes binding (the extension does not link libchx or torch).
 *
 * Python keeps what needs Python: deciding whether the fast path applies (epoch, dtypes, gradients) and building the beam.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>

typedef int (*run_track_fn)(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                            double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in, void* x_out, int64_t N,
                            const void* s_in, void* s_out, void* stream);

static run_track_fn p_run_track = NULL;
static PyObject* g_empty_like = NULL;    /* torch.empty_like */
static PyObject* g_raw_stream = NULL;    /* torch._C._cuda_getCurrentRawStream */
static PyObject* g_error = NULL;         /* cheetah_amd._lib.ChxError */
static PyObject* s_data_ptr = NULL;      /* interned "data_ptr" */

typedef struct {
    const int32_t* kinds;
    const void* const* ptrs;
    int64_t E;
    void* state;
    size_t state_bytes;
    int code;
} host_plan;

static void plan_free(PyObject* cap) { free(PyCapsule_GetPointer(cap, "chx.host_plan")); }

/* bind(chx_run_track address, torch.empty_like, raw-stream getter, error class) */
static PyObject* host_bind(PyObject* self, PyObject* args) {
    unsigned long long addr;
    PyObject *el, *rs, *err;
    if (!PyArg_ParseTuple(args, "KOOO", &addr, &el, &rs, &err)) return NULL;
    p_run_track = (run_track_fn)(uintptr_t)addr;
    Py_XDECREF(g_empty_like); Py_XDECREF(g_raw_stream); Py_XDECREF(g_error);
    Py_INCREF(el); Py_INCREF(rs); Py_INCREF(err);
    g_empty_like = el; g_raw_stream = rs; g_error = err;
    Py_RETURN_NONE;
}

/* plan(kinds address, pointer-table address, E, state address, state bytes, dtype code) -> capsule. The two arrays and the state
 * buffer belong to the _FastRun that asks for the capsule and outlive it. */
static PyObject* host_plan_new(PyObject* self, PyObject* args) {
    unsigned long long kinds, ptrs, state, state_bytes;
    long long E;
    int code;
    if (!PyArg_ParseTuple(args, "KKLKKi", &kinds, &ptrs, &E, &state, &state_bytes, &code)) return NULL;
    host_plan* p = (host_plan*)malloc(sizeof(host_plan));
    if (!p) return PyErr_NoMemory();
    p->kinds = (const int32_t*)(uintptr_t)kinds;
    p->ptrs = (const void* const*)(uintptr_t)ptrs;
    p->E = E;
    p->state = (void*)(uintptr_t)state;
    p->state_bytes = (size_t)state_bytes;
    p->code = code;
    return PyCapsule_New(p, "chx.host_plan", plan_free);
}

static int tensor_ptr(PyObject* t, void** out) {
    PyObject* v = PyObject_CallMethodNoArgs(t, s_data_ptr);
    if (!v) return -1;
    *out = PyLong_AsVoidPtr(v);
    Py_DECREF(v);
    return (*out == NULL && PyErr_Occurred()) ? -1 : 0;
}

/* run_track(plan, x, N, energy, s_in | None, mass_eV, n_charges, device_index) -> (out, s_out | None)
 * x: contiguous, 16-byte aligned (N, 7) tensor of the plan's dtype on the current device; energy, s_in: 0-d tensors of the same
 * dtype and device. out / s_out are fresh tensors (torch.empty_like). */
static PyObject* host_run_track(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 8) { PyErr_SetString(PyExc_TypeError, "run_track takes 8 arguments"); return NULL; }
    if (!p_run_track) { PyErr_SetString(PyExc_RuntimeError, "cheetah_amd._chxhost is not bound to libchx"); return NULL; }
    host_plan* p = (host_plan*)PyCapsule_GetPointer(args[0], "chx.host_plan");
    if (!p) return NULL;
    PyObject *x = args[1], *energy = args[3], *s_in = args[4];
    const long long N = PyLong_AsLongLong(args[2]);
    const double mass = PyFloat_AsDouble(args[5]), nq = PyFloat_AsDouble(args[6]);
    if (PyErr_Occurred()) return NULL;
    void *xp, *ep, *sp = NULL, *op, *sop = NULL, *stream;
    if (tensor_ptr(x, &xp) || tensor_ptr(energy, &ep)) return NULL;
    PyObject* out = PyObject_CallOneArg(g_empty_like, x);
    if (!out) return NULL;
    PyObject* s_out = Py_None;
    if (s_in != Py_None) {
        s_out = PyObject_CallOneArg(g_empty_like, s_in);
        if (!s_out || tensor_ptr(s_in, &sp) || tensor_ptr(s_out, &sop)) { Py_DECREF(out); Py_XDECREF(s_out); return NULL; }
    } else {
        Py_INCREF(Py_None);
    }
    PyObject* st = PyObject_CallOneArg(g_raw_stream, args[7]);
    if (!st || tensor_ptr(out, &op)) { Py_XDECREF(st); Py_DECREF(out); Py_DECREF(s_out); return NULL; }
    stream = PyLong_AsVoidPtr(st);
    Py_DECREF(st);
    const int rc = p_run_track(p->kinds, p->ptrs, p->E, ep, mass, nq, p->code, p->state, p->state_bytes, xp, op, (int64_t)N, sp, sop,
                               stream);
    if (rc != 0) {
        Py_DECREF(out); Py_DECREF(s_out);
        PyErr_Format(g_error ? g_error : PyExc_RuntimeError, "chx_run_track failed with status %d", rc);
        return NULL;
    }
    PyObject* res = PyTuple_Pack(2, out, s_out);
    Py_DECREF(out); Py_DECREF(s_out);
    return res;
}

/* ---- a stretch of lattice ([run | active cavity]+) in ONE C call: chx_lattice_track (two launches for the whole stretch) ------ */
/* chx_lattice_track_diag (include/chx.h); without monitors / apertures: survival = survival_out = readings = workspace = NULL */
typedef int (*lattice_track_fn)(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                                double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in,
                                void* x_out, int64_t N, int64_t B, int64_t Bx, int64_t Bm, int64_t Bw, int small_runs,
                                void* energy_out, const void* s_in, void* s_out, const void* survival, void* survival_out,
                                int64_t n_bpm, void* readings, void* workspace, size_t workspace_bytes, void* stream);
static lattice_track_fn p_lattice_track = NULL;

typedef struct {
    const int64_t* table;
    int64_t n_items, n_elems, n_ptrs;
    void* state;
    size_t state_bytes;
    int code;
} lattice_plan;

static void lattice_free(PyObject* cap) { free(PyCapsule_GetPointer(cap, "chx.lattice_plan")); }

static PyObject* host_bind_lattice(PyObject* self, PyObject* args) {
    unsigned long long addr;
    if (!PyArg_ParseTuple(args, "K", &addr)) return NULL;
    p_lattice_track = (lattice_track_fn)(uintptr_t)addr;
    Py_RETURN_NONE;
}

/* lattice_plan(table address (device), n_items, n_elems, n_ptrs, state address, state bytes, dtype code) -> capsule */
static PyObject* host_lattice_plan(PyObject* self, PyObject* args) {
    unsigned long long table, state, state_bytes;
    long long n_items, n_elems, n_ptrs;
    int code;
    if (!PyArg_ParseTuple(args, "KLLLKKi", &table, &n_items, &n_elems, &n_ptrs, &state, &state_bytes, &code)) return NULL;
    lattice_plan* p = (lattice_plan*)malloc(sizeof(lattice_plan));
    if (!p) return PyErr_NoMemory();
    p->table = (const int64_t*)(uintptr_t)table;
    p->n_items = n_items; p->n_elems = n_elems; p->n_ptrs = n_ptrs;
    p->state = (void*)(uintptr_t)state;
    p->state_bytes = (size_t)state_bytes;
    p->code = code;
    return PyCapsule_New(p, "chx.lattice_plan", lattice_free);
}

/* lattice_track(plan, x, N, energy, s_in | None, mass_eV, n_charges, device_index
 *               [, survival | None, survival_out | None, n_bpm, readings | None, workspace | None, workspace bytes, B])
 *   -> (out, energy_out, s_out | None)
 * the seven optional arguments: B beams of N particles (x: (B, N, 7) contiguous) and / or a stretch with active BPMs / apertures —
 * survival_out (B, N), readings (n_bpm, B, 2) and the workspace are the caller's tensors */
static PyObject* host_lattice_track(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 8 && nargs != 15 && nargs != 20 && nargs != 21) { PyErr_SetString(PyExc_TypeError, "lattice_track takes 8, 15, 20 or 21 arguments"); return NULL; }
    void *surv = NULL, *surv_out = NULL, *readings = NULL, *bws = NULL;
    long long n_bpm = 0, beams = 1, bx = -1, bm = 1, bw = -1, small_runs = 0;
    unsigned long long bws_bytes = 0;
    PyObject *out_given = NULL, *e_out_given = NULL;   /* (21st: the outgoing energies, when their shape is not the incoming one's) */
    if (nargs == 21 && args[20] != Py_None) e_out_given = args[20];
    if (nargs >= 20) {           /* ..., Bx, Bm, Bw, small_runs, out: vectorised lattice settings / one beam shared by the rows */
        bx = PyLong_AsLongLong(args[15]);
        bm = PyLong_AsLongLong(args[16]);
        bw = PyLong_AsLongLong(args[17]);
        small_runs = PyLong_AsLongLong(args[18]);
        if (PyErr_Occurred()) return NULL;
        if (args[19] != Py_None) out_given = args[19];
    }
    if (nargs >= 15) {
        n_bpm = PyLong_AsLongLong(args[10]);
        bws_bytes = PyLong_AsUnsignedLongLong(args[13]);
        beams = PyLong_AsLongLong(args[14]);
        if (PyErr_Occurred()) return NULL;
        if ((args[8] != Py_None && tensor_ptr(args[8], &surv)) || (args[9] != Py_None && tensor_ptr(args[9], &surv_out)) ||
            (args[11] != Py_None && tensor_ptr(args[11], &readings)) || (args[12] != Py_None && tensor_ptr(args[12], &bws)))
            return NULL;
    }
    if (!p_lattice_track) { PyErr_SetString(PyExc_RuntimeError, "cheetah_amd._chxhost is not bound to chx_lattice_track"); return NULL; }
    lattice_plan* p = (lattice_plan*)PyCapsule_GetPointer(args[0], "chx.lattice_plan");
    if (!p) return NULL;
    PyObject *x = args[1], *energy = args[3], *s_in = args[4];
    const long long N = PyLong_AsLongLong(args[2]);
    const double mass = PyFloat_AsDouble(args[5]), nq = PyFloat_AsDouble(args[6]);
    if (PyErr_Occurred()) return NULL;
    void *xp, *ep, *sp = NULL, *op, *eop, *sop = NULL, *stream;
    if (tensor_ptr(x, &xp) || tensor_ptr(energy, &ep)) return NULL;
    PyObject* out;
    if (out_given) { out = out_given; Py_INCREF(out); }
    else out = PyObject_CallOneArg(g_empty_like, x);
    if (!out) return NULL;
    PyObject* e_out;
    if (e_out_given) { e_out = e_out_given; Py_INCREF(e_out); }
    else e_out = PyObject_CallOneArg(g_empty_like, energy);
    if (!e_out) { Py_DECREF(out); return NULL; }
    PyObject* s_out = Py_None;
    if (s_in != Py_None) {
        s_out = PyObject_CallOneArg(g_empty_like, s_in);
        if (!s_out || tensor_ptr(s_in, &sp) || tensor_ptr(s_out, &sop)) { Py_DECREF(out); Py_DECREF(e_out); Py_XDECREF(s_out); return NULL; }
    } else {
        Py_INCREF(Py_None);
    }
    PyObject* st = PyObject_CallOneArg(g_raw_stream, args[7]);
    if (!st || tensor_ptr(out, &op) || tensor_ptr(e_out, &eop)) { Py_XDECREF(st); Py_DECREF(out); Py_DECREF(e_out); Py_DECREF(s_out); return NULL; }
    stream = PyLong_AsVoidPtr(st);
    Py_DECREF(st);
    const int rc = p_lattice_track(p->table, p->n_items, p->n_elems, p->n_ptrs, ep, mass, nq, p->code, p->state, p->state_bytes, xp, op,
                                   (int64_t)N, (int64_t)beams, (int64_t)(bx < 0 ? beams : bx), (int64_t)bm, (int64_t)(bw < 0 ? beams : bw), (int)small_runs, eop, sp, sop, surv, surv_out,
                                   (int64_t)n_bpm, readings, bws, (size_t)bws_bytes, stream);
    if (rc != 0) {
        Py_DECREF(out); Py_DECREF(e_out); Py_DECREF(s_out);
        PyErr_Format(g_error ? g_error : PyExc_RuntimeError, "chx_lattice_track failed with status %d", rc);
        return NULL;
    }
    PyObject* res = PyTuple_Pack(3, out, e_out, s_out);
    Py_DECREF(out); Py_DECREF(e_out); Py_DECREF(s_out);
    return res;
}

/* ---- a stretch's table from a host buffer to the device: chx_table_store (the words ride in a launch's arguments) ---------------- */
typedef int (*table_store_fn)(const int64_t* host_words, int64_t n, void* table, void* stream);
static table_store_fn p_table_store = NULL;
static long long g_table_store_max = 0;

static PyObject* host_bind_table_store(PyObject* self, PyObject* args) {
    unsigned long long addr;
    long long max_words;
    if (!PyArg_ParseTuple(args, "KL", &addr, &max_words)) return NULL;
    p_table_store = (table_store_fn)(uintptr_t)addr;
    g_table_store_max = max_words;
    Py_RETURN_NONE;
}

/* table_store(words (a contiguous buffer of int64: a numpy array), table address (device), device index) -> True, or False when the
 * table has more words than one launch carries (the caller then uploads it through a staging tensor as before) */
static PyObject* host_table_store(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 3) { PyErr_SetString(PyExc_TypeError, "table_store takes 3 arguments"); return NULL; }
    if (!p_table_store) { PyErr_SetString(PyExc_RuntimeError, "cheetah_amd._chxhost is not bound to chx_table_store"); return NULL; }
    Py_buffer view;
    if (PyObject_GetBuffer(args[0], &view, PyBUF_C_CONTIGUOUS) != 0) return NULL;
    const long long n = (long long)(view.len / 8);
    if (view.len % 8 != 0 || n < 1) { PyBuffer_Release(&view); PyErr_SetString(PyExc_ValueError, "table_store: a buffer of int64 words"); return NULL; }
    if (n > g_table_store_max) { PyBuffer_Release(&view); Py_RETURN_FALSE; }
    void* table = PyLong_AsVoidPtr(args[1]);
    if (!table && PyErr_Occurred()) { PyBuffer_Release(&view); return NULL; }
    PyObject* st = PyObject_CallOneArg(g_raw_stream, args[2]);
    if (!st) { PyBuffer_Release(&view); return NULL; }
    void* stream = PyLong_AsVoidPtr(st);
    Py_DECREF(st);
    const int rc = p_table_store((const int64_t*)view.buf, (int64_t)n, table, stream);
    PyBuffer_Release(&view);
    if (rc != 0) {
        PyErr_Format(g_error ? g_error : PyExc_RuntimeError, "chx_table_store failed with status %d", rc);
        return NULL;
    }
    Py_RETURN_TRUE;
}

static PyMethodDef methods[] = {
    {"bind_table_store", host_bind_table_store, METH_VARARGS, "bind_table_store(chx_table_store address, words per launch)"},
    {"table_store", (PyCFunction)(void (*)(void))host_table_store, METH_FASTCALL,
     "table_store(int64 words buffer, device table address, device index) -> bool (False: too many words for one launch)"},
    {"bind_lattice", host_bind_lattice, METH_VARARGS, "bind_lattice(chx_lattice_track_diag address)"},
    {"lattice_plan", host_lattice_plan, METH_VARARGS, "lattice_plan(table addr, n_items, n_elems, n_ptrs, state addr, state bytes, dtype code) -> capsule"},
    {"lattice_track", (PyCFunction)(void (*)(void))host_lattice_track, METH_FASTCALL,
     "lattice_track(plan, x, N, energy, s_in | None, mass_eV, n_charges, device index[, survival | None, survival_out | None, n_bpm, readings | None, workspace | None, workspace bytes, B]) -> (out, energy_out, s_out | None)"},
    {"bind", host_bind, METH_VARARGS, "bind(chx_run_track address, torch.empty_like, raw stream getter, error class)"},
    {"plan", host_plan_new, METH_VARARGS, "plan(kinds addr, pointer-table addr, E, state addr, state bytes, dtype code) -> capsule"},
    {"run_track", (PyCFunction)(void (*)(void))host_run_track, METH_FASTCALL,
     "run_track(plan, x, N, energy, s_in | None, mass_eV, n_charges, device index) -> (out, s_out | None)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_chxhost", "host-side fast path of cheetah_amd (see chx_host.c)", -1, methods};

PyMODINIT_FUNC PyInit__chxhost(void) {
    s_data_ptr = PyUnicode_InternFromString("data_ptr");
    return PyModule_Create(&module);
}
