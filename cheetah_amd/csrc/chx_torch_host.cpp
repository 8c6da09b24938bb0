// cheetah_amd._chxtorch — the host step of a lattice stretch against torch's C++ side, and the requires_grad scan.
//
// An eager RL control step (`Segment.track` on a small beam, then the screen's reading; cheetah/accelerator/segment.py:545-574,
// screen.py:187-344) is two launches of ~5 us here; what bounded it was the host: every output tensor allocated through the Python
// dispatcher (~2.5 us each: outgoing rows, energy, path length, the screen's record, its image), ctypes marshalling of thirty
// arguments, pointer look-ups. This module does that sequence in C++: tensors come from ATen directly (at::empty, ~0.7 us), the
// plan (table / state addresses, the screens' image shapes) sits in a capsule, and libchx's entry points are called through
// function pointers handed over by the ctypes binding (the extension links torch, not libchx). Python keeps what needs Python:
// whether the path applies (epoch, dtypes, gradients, sharding) and the beam objects.
//
// `any_requires_grad(tuple)`: `torch._C._any_requires_grad(*tensors)` goes through torch's generic argument parser (~23 ns per
// tensor, 7 us for the 300 setting tensors of a 100-element FODO); reading the flag from the THPVariable is ~2 ns per tensor.
#include <Python.h>

#include <ATen/ATen.h>
#include <torch/csrc/autograd/python_variable.h>

#include <cstdint>
#include <vector>

#include "chx.h"

namespace {

using track_screens_fn = decltype(&chx_lattice_track_screens);
using parameter_screens_fn = decltype(&chx_parameter_lattice_track_screens);

track_screens_fn p_track = nullptr;
parameter_screens_fn p_parameter = nullptr;
PyObject* g_raw_stream = nullptr;   // torch._C._cuda_getCurrentRawStream
PyObject* g_error = nullptr;        // cheetah_amd._lib.ChxError

struct ScreenShape {
    int deposit;          // the particle pass may deposit the cloud-in-cell image of this screen
    int64_t bins_x, bins_y;
};

struct StretchPlan {
    const int64_t* table;
    int64_t n_items, n_elems, n_ptrs;
    void* state;
    size_t state_bytes;
    int code;
    std::vector<ScreenShape> screens;
};

void plan_free(PyObject* cap) { delete static_cast<StretchPlan*>(PyCapsule_GetPointer(cap, "chx.stretch_plan")); }

PyObject* any_requires_grad(PyObject*, PyObject* seq) {
    if (!PyTuple_Check(seq)) {
        PyErr_SetString(PyExc_TypeError, "any_requires_grad expects a tuple of tensors");
        return nullptr;
    }
    const Py_ssize_t n = PyTuple_GET_SIZE(seq);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* o = PyTuple_GET_ITEM(seq, i);
        if (THPVariable_Check(o) && THPVariable_Unpack(o).requires_grad()) Py_RETURN_TRUE;
    }
    Py_RETURN_FALSE;
}

// bind(chx_lattice_track_screens address, chx_parameter_lattice_track_screens address, raw-stream getter, error class)
PyObject* host_bind(PyObject*, PyObject* args) {
    unsigned long long a, b;
    PyObject *rs, *err;
    if (!PyArg_ParseTuple(args, "KKOO", &a, &b, &rs, &err)) return nullptr;
    p_track = reinterpret_cast<track_screens_fn>(static_cast<uintptr_t>(a));
    p_parameter = reinterpret_cast<parameter_screens_fn>(static_cast<uintptr_t>(b));
    Py_XDECREF(g_raw_stream);
    Py_XDECREF(g_error);
    Py_INCREF(rs);
    Py_INCREF(err);
    g_raw_stream = rs;
    g_error = err;
    Py_RETURN_NONE;
}

// stretch_plan(table address (device), n_items, n_elems, n_ptrs, state address, state bytes, dtype code,
//              ((deposit, bins_x, bins_y), ...) per screen slot) -> capsule
PyObject* host_plan(PyObject*, PyObject* args) {
    unsigned long long table, state, state_bytes;
    long long n_items, n_elems, n_ptrs;
    int code;
    PyObject* screens;
    if (!PyArg_ParseTuple(args, "KLLLKKiO", &table, &n_items, &n_elems, &n_ptrs, &state, &state_bytes, &code, &screens)) return nullptr;
    if (!PyTuple_Check(screens) || PyTuple_GET_SIZE(screens) > CHX_LATTICE_MAX_SCREENS) {
        PyErr_SetString(PyExc_ValueError, "screens: a tuple of at most CHX_LATTICE_MAX_SCREENS (deposit, bins_x, bins_y) triples");
        return nullptr;
    }
    auto* p = new StretchPlan{reinterpret_cast<const int64_t*>(static_cast<uintptr_t>(table)), n_items, n_elems, n_ptrs,
                              reinterpret_cast<void*>(static_cast<uintptr_t>(state)), static_cast<size_t>(state_bytes), code, {}};
    for (Py_ssize_t k = 0; k < PyTuple_GET_SIZE(screens); ++k) {
        int deposit;
        long long bx, by;
        if (!PyArg_ParseTuple(PyTuple_GET_ITEM(screens, k), "iLL", &deposit, &bx, &by)) {
            delete p;
            return nullptr;
        }
        p->screens.push_back(ScreenShape{deposit, bx, by});
    }
    return PyCapsule_New(p, "chx.stretch_plan", plan_free);
}

bool current_stream(PyObject* device_index, void** stream) {
    PyObject* st = PyObject_CallOneArg(g_raw_stream, device_index);
    if (!st) return false;
    *stream = PyLong_AsVoidPtr(st);
    Py_DECREF(st);
    return !(*stream == nullptr && PyErr_Occurred());
}

inline const at::Tensor& unpack(PyObject* o) { return THPVariable_Unpack(o); }

PyObject* fail(int rc, const char* what) {
    PyErr_Format(g_error ? g_error : PyExc_RuntimeError, "%s failed with status %d", what, rc);
    return nullptr;
}

// lattice_track_screens(plan, x (N, 7), energy, s_in, charges (N,), survival (N,) | None, mass_eV, n_charges, device index,
//                       image_limit, survival_out | None, n_bpm, readings | None, workspace | None, workspace bytes)
//   -> (out, energy_out, s_out, (record, ...), (image | None, ...))
// record of screen k: ONE tensor of 9 N + 2 values [rows N x 7 | charges N | survival N | energy | s] of the beam AT the screen;
// image: (bins_y, bins_x), deposited by the particle pass when the screen allows it and N <= image_limit (else None: the caller
// forms it from the record when it is asked for).
PyObject* host_track(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 15) {
        PyErr_SetString(PyExc_TypeError, "lattice_track_screens takes 15 arguments");
        return nullptr;
    }
    if (!p_track) {
        PyErr_SetString(PyExc_RuntimeError, "cheetah_amd._chxtorch is not bound to libchx");
        return nullptr;
    }
    auto* p = static_cast<StretchPlan*>(PyCapsule_GetPointer(args[0], "chx.stretch_plan"));
    if (!p) return nullptr;
    const at::Tensor& x = unpack(args[1]);
    const at::Tensor& energy = unpack(args[2]);
    const at::Tensor& s_in = unpack(args[3]);
    const at::Tensor& charges = unpack(args[4]);
    const double mass = PyFloat_AsDouble(args[6]), nq = PyFloat_AsDouble(args[7]);
    const long long image_limit = PyLong_AsLongLong(args[9]);
    const long long n_bpm = PyLong_AsLongLong(args[11]);
    const unsigned long long ws_bytes = PyLong_AsUnsignedLongLong(args[14]);
    if (PyErr_Occurred()) return nullptr;
    const void* survival = args[5] == Py_None ? nullptr : unpack(args[5]).data_ptr();
    void* survival_out = args[10] == Py_None ? nullptr : unpack(args[10]).data_ptr();
    void* readings = args[12] == Py_None ? nullptr : unpack(args[12]).data_ptr();
    void* workspace = args[13] == Py_None ? nullptr : unpack(args[13]).data_ptr();
    void* stream;
    if (!current_stream(args[8], &stream)) return nullptr;
    const int64_t N = x.size(0);
    const auto opts = x.options();
    at::Tensor out = at::empty_like(x);
    at::Tensor e_out = at::empty_like(energy);
    at::Tensor s_out = at::empty_like(s_in);
    const size_t n_screens = p->screens.size();
    chx_lattice_screen scr[CHX_LATTICE_MAX_SCREENS] = {};
    at::Tensor recs[CHX_LATTICE_MAX_SCREENS], images[CHX_LATTICE_MAX_SCREENS];
    const size_t esize = x.element_size();
    for (size_t k = 0; k < n_screens; ++k) {
        recs[k] = at::empty({9 * N + 2}, opts);
        char* base = static_cast<char*>(recs[k].data_ptr());
        scr[k].rows = base;
        scr[k].charges = base + 7 * N * esize;
        scr[k].survival = base + 8 * N * esize;
        scr[k].energy = base + 9 * N * esize;
        scr[k].s = base + (9 * N + 1) * esize;
        if (p->screens[k].deposit && N <= image_limit) {
            images[k] = at::empty({p->screens[k].bins_y, p->screens[k].bins_x}, opts);
            scr[k].image = images[k].data_ptr();
            scr[k].image_bytes = static_cast<int64_t>(p->screens[k].bins_x * p->screens[k].bins_y * esize);
        }
    }
    const int rc = p_track(p->table, p->n_items, p->n_elems, p->n_ptrs, energy.data_ptr(), mass, nq, p->code, p->state, p->state_bytes,
                           x.data_ptr(), out.data_ptr(), N, 1, 1, 1, 1, 0, e_out.data_ptr(), s_in.data_ptr(), s_out.data_ptr(), survival,
                           survival_out, n_bpm, readings, workspace, static_cast<size_t>(ws_bytes), charges.data_ptr(), scr,
                           static_cast<int64_t>(n_screens), stream);
    if (rc != 0) return fail(rc, "chx_lattice_track_screens");
    PyObject* rec_t = PyTuple_New(static_cast<Py_ssize_t>(n_screens));
    PyObject* img_t = PyTuple_New(static_cast<Py_ssize_t>(n_screens));
    if (!rec_t || !img_t) {
        Py_XDECREF(rec_t);
        Py_XDECREF(img_t);
        return nullptr;
    }
    for (size_t k = 0; k < n_screens; ++k) {
        PyTuple_SET_ITEM(rec_t, k, THPVariable_Wrap(recs[k]));
        if (images[k].defined()) {
            PyTuple_SET_ITEM(img_t, k, THPVariable_Wrap(images[k]));
        } else {
            Py_INCREF(Py_None);
            PyTuple_SET_ITEM(img_t, k, Py_None);
        }
    }
    PyObject* res = PyTuple_New(5);
    if (!res) {
        Py_DECREF(rec_t);
        Py_DECREF(img_t);
        return nullptr;
    }
    PyTuple_SET_ITEM(res, 0, THPVariable_Wrap(out));
    PyTuple_SET_ITEM(res, 1, THPVariable_Wrap(e_out));
    PyTuple_SET_ITEM(res, 2, THPVariable_Wrap(s_out));
    PyTuple_SET_ITEM(res, 3, rec_t);
    PyTuple_SET_ITEM(res, 4, img_t);
    return res;
}

// parameter_lattice_track_screens(plan, mu (7,), cov (7, 7), energy, s_in, total_charge, mass_eV, n_charges, device index,
//                                 ((geom, shift, width, height) | None, ...) per screen, n_bpm, readings | None)
//   -> (mu_out, cov_out, energy_out, s_out, (record, ...), (image | None, ...))
// record of screen k: ONE tensor of 59 values [mu 7 | cov 49 | energy | s | total charge] of the beam AT the screen; image
// (height, width): the bivariate normal density of the recorded moments (screen.py:255-291), when geometry is given.
PyObject* host_parameter(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 12) {
        PyErr_SetString(PyExc_TypeError, "parameter_lattice_track_screens takes 12 arguments");
        return nullptr;
    }
    if (!p_parameter) {
        PyErr_SetString(PyExc_RuntimeError, "cheetah_amd._chxtorch is not bound to libchx");
        return nullptr;
    }
    auto* p = static_cast<StretchPlan*>(PyCapsule_GetPointer(args[0], "chx.stretch_plan"));
    if (!p) return nullptr;
    const at::Tensor& mu = unpack(args[1]);
    const at::Tensor& cov = unpack(args[2]);
    const at::Tensor& energy = unpack(args[3]);
    const at::Tensor& s_in = unpack(args[4]);
    const at::Tensor& q = unpack(args[5]);
    const double mass = PyFloat_AsDouble(args[6]), nq = PyFloat_AsDouble(args[7]);
    const long long n_bpm = PyLong_AsLongLong(args[10]);
    if (PyErr_Occurred()) return nullptr;
    void* readings = args[11] == Py_None ? nullptr : unpack(args[11]).data_ptr();
    void* stream;
    if (!current_stream(args[8], &stream)) return nullptr;
    PyObject* geoms = args[9];
    const size_t n_screens = p->screens.size();
    if (!PyTuple_Check(geoms) || static_cast<size_t>(PyTuple_GET_SIZE(geoms)) != n_screens) {
        PyErr_SetString(PyExc_ValueError, "one geometry entry per screen of the plan");
        return nullptr;
    }
    const auto opts = mu.options();
    at::Tensor mu_out = at::empty_like(mu), cov_out = at::empty_like(cov), e_out = at::empty_like(energy), s_out = at::empty_like(s_in);
    chx_lattice_screen scr[CHX_LATTICE_MAX_SCREENS] = {};
    at::Tensor recs[CHX_LATTICE_MAX_SCREENS], images[CHX_LATTICE_MAX_SCREENS];
    const size_t esize = mu.element_size();
    for (size_t k = 0; k < n_screens; ++k) {
        recs[k] = at::empty({59}, opts);
        char* base = static_cast<char*>(recs[k].data_ptr());
        scr[k].mu = base;
        scr[k].cov = base + 7 * esize;
        scr[k].energy = base + 56 * esize;
        scr[k].s = base + 57 * esize;
        scr[k].total_charge_out = base + 58 * esize;
        scr[k].total_charge = q.data_ptr();
        PyObject* g = PyTuple_GET_ITEM(geoms, k);
        if (g != Py_None) {
            PyObject *geom, *shift;
            int width, height;
            if (!PyArg_ParseTuple(g, "OOii", &geom, &shift, &width, &height)) return nullptr;
            images[k] = at::empty({height, width}, opts);
            scr[k].image = images[k].data_ptr();
            scr[k].geom = unpack(geom).data_ptr();
            scr[k].shift = unpack(shift).data_ptr();
            scr[k].width = width;
            scr[k].height = height;
        }
    }
    const int rc = p_parameter(p->table, p->n_items, p->n_elems, p->n_ptrs, energy.data_ptr(), mass, nq, p->code, p->state, p->state_bytes,
                               mu.data_ptr(), cov.data_ptr(), 1, 1, 1, 1, 0, mu_out.data_ptr(), cov_out.data_ptr(), e_out.data_ptr(),
                               s_in.data_ptr(), s_out.data_ptr(), n_bpm, readings, scr, static_cast<int64_t>(n_screens), stream);
    if (rc != 0) return fail(rc, "chx_parameter_lattice_track_screens");
    PyObject* rec_t = PyTuple_New(static_cast<Py_ssize_t>(n_screens));
    PyObject* img_t = PyTuple_New(static_cast<Py_ssize_t>(n_screens));
    if (!rec_t || !img_t) {
        Py_XDECREF(rec_t);
        Py_XDECREF(img_t);
        return nullptr;
    }
    for (size_t k = 0; k < n_screens; ++k) {
        PyTuple_SET_ITEM(rec_t, k, THPVariable_Wrap(recs[k]));
        if (images[k].defined()) {
            PyTuple_SET_ITEM(img_t, k, THPVariable_Wrap(images[k]));
        } else {
            Py_INCREF(Py_None);
            PyTuple_SET_ITEM(img_t, k, Py_None);
        }
    }
    PyObject* res = PyTuple_New(6);
    if (!res) {
        Py_DECREF(rec_t);
        Py_DECREF(img_t);
        return nullptr;
    }
    PyTuple_SET_ITEM(res, 0, THPVariable_Wrap(mu_out));
    PyTuple_SET_ITEM(res, 1, THPVariable_Wrap(cov_out));
    PyTuple_SET_ITEM(res, 2, THPVariable_Wrap(e_out));
    PyTuple_SET_ITEM(res, 3, THPVariable_Wrap(s_out));
    PyTuple_SET_ITEM(res, 4, rec_t);
    PyTuple_SET_ITEM(res, 5, img_t);
    return res;
}

PyMethodDef methods[] = {
    {"any_requires_grad", any_requires_grad, METH_O, "any_requires_grad(tuple_of_tensors) -> bool (non-tensor items count as False)"},
    {"bind", host_bind, METH_VARARGS, "bind(chx_lattice_track_screens address, chx_parameter_lattice_track_screens address, raw stream getter, error class)"},
    {"stretch_plan", host_plan, METH_VARARGS, "stretch_plan(table addr, n_items, n_elems, n_ptrs, state addr, state bytes, dtype code, screens) -> capsule"},
    {"lattice_track_screens", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(host_track)), METH_FASTCALL,
     "lattice_track_screens(plan, x, energy, s_in, charges, survival | None, mass_eV, n_charges, device index, image_limit, survival_out | None, "
     "n_bpm, readings | None, workspace | None, workspace bytes) -> (out, energy_out, s_out, records, images)"},
    {"parameter_lattice_track_screens", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(host_parameter)), METH_FASTCALL,
     "parameter_lattice_track_screens(plan, mu, cov, energy, s_in, total_charge, mass_eV, n_charges, device index, geometries, n_bpm, "
     "readings | None) -> (mu_out, cov_out, energy_out, s_out, records, images)"},
    {nullptr, nullptr, 0, nullptr}};

struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_chxtorch", "torch-side host step of cheetah_amd (see chx_torch_host.cpp)", -1, methods};

}  // namespace

PyMODINIT_FUNC PyInit__chxtorch(void) { return PyModule_Create(&moddef); }
