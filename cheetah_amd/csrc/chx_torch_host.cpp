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
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/csrc/autograd/python_variable.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "chx.h"

namespace {

using track_screens_fn = decltype(&chx_lattice_track_screens);
using parameter_screens_fn = decltype(&chx_parameter_lattice_track_screens);

track_screens_fn p_track = nullptr;
parameter_screens_fn p_parameter = nullptr;
// what the differentiable nodes below call (bound by name, `bind`)
decltype(&chx_run_build_compose) p_run_build_compose = nullptr;
decltype(&chx_run_vjp_masked) p_run_vjp_masked = nullptr;
decltype(&chx_run_vjp_workspace_bytes) p_run_vjp_workspace_bytes = nullptr;
decltype(&chx_run_vjp_entry) p_run_vjp_entry = nullptr;
decltype(&chx_run_vjp_entry_workspace_bytes) p_run_vjp_entry_workspace_bytes = nullptr;
decltype(&chx_apply_affine7_bwd) p_apply_bwd = nullptr;
decltype(&chx_apply_bwd_workspace_bytes) p_apply_bwd_workspace_bytes = nullptr;
decltype(&chx_moments_entry) p_moments_entry = nullptr;
decltype(&chx_moments_workspace_bytes) p_moments_workspace_bytes = nullptr;
decltype(&chx_moment_entry) p_moment_entry = nullptr;
decltype(&chx_moment_entry_mapped_bwd) p_moment_entry_mapped_bwd = nullptr;
decltype(&chx_lattice_moment_blocks) p_moment_blocks = nullptr;
decltype(&chx_lattice_screen_moments) p_screen_moments = nullptr;
PyObject* g_error = nullptr;        // cheetah_amd._lib.ChxError

// the stream torch's kernels of this thread go to on the tensor's device (also inside a backward pass: the engine restores the
// forward's stream) — libchx enqueues on the caller's stream
inline void* stream_of(const at::Tensor& t) { return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream()); }
inline int code_of(const at::Tensor& t) { return t.scalar_type() == at::kDouble ? CHX_F64 : CHX_F32; }

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

// bind({symbol name: address}, error class): the libchx entry points this module calls
PyObject* host_bind(PyObject*, PyObject* args) {
    PyObject *table, *err;
    if (!PyArg_ParseTuple(args, "O!O", &PyDict_Type, &table, &err)) return nullptr;
    auto take = [&](const char* name, auto& slot) -> bool {
        PyObject* v = PyDict_GetItemString(table, name);
        if (!v) {
            PyErr_Format(PyExc_KeyError, "bind: no address for %s", name);
            return false;
        }
        const unsigned long long a = PyLong_AsUnsignedLongLong(v);
        if (PyErr_Occurred()) return false;
        slot = reinterpret_cast<std::remove_reference_t<decltype(slot)>>(static_cast<uintptr_t>(a));
        return true;
    };
    if (!take("chx_lattice_track_screens", p_track) || !take("chx_parameter_lattice_track_screens", p_parameter) ||
        !take("chx_run_build_compose", p_run_build_compose) || !take("chx_run_vjp_masked", p_run_vjp_masked) ||
        !take("chx_run_vjp_workspace_bytes", p_run_vjp_workspace_bytes) || !take("chx_apply_affine7_bwd", p_apply_bwd) ||
        !take("chx_apply_bwd_workspace_bytes", p_apply_bwd_workspace_bytes) || !take("chx_moments_entry", p_moments_entry) ||
        !take("chx_moments_workspace_bytes", p_moments_workspace_bytes) || !take("chx_moment_entry", p_moment_entry) ||
        !take("chx_moment_entry_mapped_bwd", p_moment_entry_mapped_bwd) || !take("chx_lattice_moment_blocks", p_moment_blocks) ||
        !take("chx_lattice_screen_moments", p_screen_moments) || !take("chx_run_vjp_entry", p_run_vjp_entry) ||
        !take("chx_run_vjp_entry_workspace_bytes", p_run_vjp_entry_workspace_bytes))
        return nullptr;
    Py_XDECREF(g_error);
    Py_INCREF(err);
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

inline const at::Tensor& unpack(PyObject* o) { return THPVariable_Unpack(o); }

// What the Python caller promises, checked where a mismatch would otherwise write out of bounds on the device: `t` is a tensor on a
// ROCm device, contiguous, of the plan's dtype, with exactly `numel` elements (numel < 0: any). Sets a Python error and returns false.
bool tensor_ok(PyObject* o, const char* what, int code, int64_t numel, const at::Tensor* same_device_as = nullptr) {
    if (!THPVariable_Check(o)) {
        PyErr_Format(PyExc_TypeError, "%s: a tensor is expected", what);
        return false;
    }
    const at::Tensor& t = THPVariable_Unpack(o);
    const bool dtype_ok = code == CHX_F64 ? t.scalar_type() == at::kDouble : t.scalar_type() == at::kFloat;
    if (!t.is_cuda() || !t.is_contiguous() || !dtype_ok || (numel >= 0 && t.numel() != numel) ||
        (same_device_as && t.device() != same_device_as->device())) {
        PyErr_Format(PyExc_ValueError, "%s: a contiguous %s tensor of %lld elements on the beam's device is expected (got %lld elements, %s)", what,
                     code == CHX_F64 ? "float64" : "float32", static_cast<long long>(numel), static_cast<long long>(t.numel()),
                     t.is_cuda() ? (t.is_contiguous() ? "dtype or device differ" : "not contiguous") : "not on a ROCm device");
        return false;
    }
    return true;
}

// a tuple slot from a tensor; a failed wrap (out of memory) leaves the Python error set and the slot NULL (the tuple's destructor copes)
inline bool set_wrapped(PyObject* tuple, Py_ssize_t i, const at::Tensor& t) {
    PyObject* w = THPVariable_Wrap(t);
    if (!w) return false;
    PyTuple_SET_ITEM(tuple, i, w);
    return true;
}

PyObject* fail(int rc, const char* what) {
    PyErr_Format(g_error ? g_error : PyExc_RuntimeError, "%s failed with status %d", what, rc);
    return nullptr;
}

// lattice_track_screens(plan, x (N, 7), energy, s_in, charges (N,), survival (N,) | None, mass_eV, n_charges, device index,
//                       image_limit, survival_out | None, n_bpm, readings | None, workspace | None, workspace bytes)
//   -> (out, energy_out, s_out, (record, ...), (image | None, ...))
// record of screen k: ONE tensor of 9 M + 2 values [rows M x 7 | charges M | survival M | energy | s] of the beam AT the screen
// (M = N, or B N for a vectorised beam x of shape (B, N, 7): the beams one behind the other);
// image: (bins_y, bins_x) — (B, bins_y, bins_x) —, deposited by the particle pass when the screen allows it and N <= image_limit (else None: the caller
// forms it from the record when it is asked for).
PyObject* host_track_impl(PyObject* const* args, Py_ssize_t nargs) {
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
    if (!tensor_ok(args[1], "x", p->code, -1)) return nullptr;
    const at::Tensor& x = unpack(args[1]);
    // (N, 7): one beam; (B, N, 7): a vectorised beam of B beams under the one lattice setting — every record and image then holds B of them
    if ((x.dim() != 2 && x.dim() != 3) || x.size(-1) != 7 || x.size(-2) < 1 || x.size(0) < 1) {
        PyErr_SetString(PyExc_ValueError, "x: an (N, 7) or (B, N, 7) tensor is expected");
        return nullptr;
    }
    const int64_t N = x.size(-2), B = x.dim() == 3 ? x.size(0) : 1;
    const long long n_bpm = PyLong_AsLongLong(args[11]);
    if (PyErr_Occurred()) return nullptr;
    if (!tensor_ok(args[2], "energy", p->code, 1, &x) || !tensor_ok(args[3], "s", p->code, 1, &x) ||
        !tensor_ok(args[4], "particle_charges", p->code, N, &x) || (args[5] != Py_None && !tensor_ok(args[5], "survival_probabilities", p->code, -1, &x)) ||
        (args[10] != Py_None && !tensor_ok(args[10], "survival_out", p->code, B * N, &x)) ||
        (args[12] != Py_None && !tensor_ok(args[12], "readings", p->code, 2 * n_bpm * B, &x)))
        return nullptr;
    int64_t Bw = 1;
    if (args[5] != Py_None) {
        const int64_t nw = unpack(args[5]).numel();
        if (nw != N && nw != B * N) {
            PyErr_SetString(PyExc_ValueError, "survival_probabilities: N values shared by the beams or B x N");
            return nullptr;
        }
        Bw = nw == N ? 1 : B;
    }
    if ((n_bpm > 0) != (args[12] != Py_None) || (args[13] != Py_None && (!THPVariable_Check(args[13]) || !unpack(args[13]).is_cuda()))) {
        PyErr_SetString(PyExc_ValueError, "readings / workspace: device tensors for every active monitor of the plan (and only then)");
        return nullptr;
    }
    const at::Tensor& energy = unpack(args[2]);
    const at::Tensor& s_in = unpack(args[3]);
    const at::Tensor& charges = unpack(args[4]);
    const double mass = PyFloat_AsDouble(args[6]), nq = PyFloat_AsDouble(args[7]);
    const long long image_limit = PyLong_AsLongLong(args[9]);
    const unsigned long long ws_bytes = PyLong_AsUnsignedLongLong(args[14]);
    if (PyErr_Occurred()) return nullptr;
    if (args[13] != Py_None && static_cast<unsigned long long>(unpack(args[13]).nbytes()) < ws_bytes) {
        PyErr_SetString(PyExc_ValueError, "workspace: smaller than the byte count handed over with it");
        return nullptr;
    }
    const c10::DeviceGuard device_guard(x.device());      // (allocations and the stream below belong to the beam's device; PyTorch-ROCm
    // tensors carry the CUDA device type: the generic guard dispatches to the registered implementation)
    const void* survival = args[5] == Py_None ? nullptr : unpack(args[5]).data_ptr();
    void* survival_out = args[10] == Py_None ? nullptr : unpack(args[10]).data_ptr();
    void* readings = args[12] == Py_None ? nullptr : unpack(args[12]).data_ptr();
    void* workspace = args[13] == Py_None ? nullptr : unpack(args[13]).data_ptr();
    void* stream = stream_of(x);
    const auto opts = x.options();
    at::Tensor out = at::empty_like(x);
    at::Tensor e_out = at::empty_like(energy);
    at::Tensor s_out = at::empty_like(s_in);
    const size_t n_screens = p->screens.size();
    chx_lattice_screen scr[CHX_LATTICE_MAX_SCREENS] = {};
    at::Tensor recs[CHX_LATTICE_MAX_SCREENS], images[CHX_LATTICE_MAX_SCREENS];
    const size_t esize = x.element_size();
    const int64_t M = B * N;          // rows of a record: the B beams one behind the other
    for (size_t k = 0; k < n_screens; ++k) {
        recs[k] = at::empty({9 * M + 2}, opts);
        char* base = static_cast<char*>(recs[k].data_ptr());
        scr[k].rows = base;
        scr[k].charges = base + 7 * M * esize;
        scr[k].survival = base + 8 * M * esize;
        scr[k].energy = base + 9 * M * esize;
        scr[k].s = base + (9 * M + 1) * esize;
        if (p->screens[k].deposit && N <= image_limit) {
            images[k] = x.dim() == 3 ? at::empty({B, p->screens[k].bins_y, p->screens[k].bins_x}, opts)
                                     : at::empty({p->screens[k].bins_y, p->screens[k].bins_x}, opts);
            scr[k].image = images[k].data_ptr();
            scr[k].image_bytes = static_cast<int64_t>(B * p->screens[k].bins_x * p->screens[k].bins_y * esize);
        }
    }
    const int rc = p_track(p->table, p->n_items, p->n_elems, p->n_ptrs, energy.data_ptr(), mass, nq, p->code, p->state, p->state_bytes,
                           x.data_ptr(), out.data_ptr(), N, B, B, 1, Bw, 0, e_out.data_ptr(), s_in.data_ptr(), s_out.data_ptr(), survival,
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
    bool wrapped = true;
    for (size_t k = 0; k < n_screens; ++k) {
        wrapped = set_wrapped(rec_t, k, recs[k]) && wrapped;
        if (images[k].defined()) {
            wrapped = set_wrapped(img_t, k, images[k]) && wrapped;
        } else {
            Py_INCREF(Py_None);
            PyTuple_SET_ITEM(img_t, k, Py_None);
        }
    }
    PyObject* res = PyTuple_New(5);
    if (!res || !wrapped) {
        Py_XDECREF(res);
        Py_DECREF(rec_t);
        Py_DECREF(img_t);
        return nullptr;
    }
    PyTuple_SET_ITEM(res, 3, rec_t);
    PyTuple_SET_ITEM(res, 4, img_t);
    if (!set_wrapped(res, 0, out) || !set_wrapped(res, 1, e_out) || !set_wrapped(res, 2, s_out)) {
        Py_DECREF(res);
        return nullptr;
    }
    return res;
}

// parameter_lattice_track_screens(plan, mu (7,), cov (7, 7), energy, s_in, total_charge, mass_eV, n_charges, device index,
//                                 ((geom, shift, width, height) | None, ...) per screen, n_bpm, readings | None)
//   -> (mu_out, cov_out, energy_out, s_out, (record, ...), (image | None, ...))
// record of screen k: ONE tensor of 59 values [mu 7 | cov 49 | energy | s | total charge] of the beam AT the screen; image
// (height, width): the bivariate normal density of the recorded moments (screen.py:255-291), when geometry is given.
PyObject* host_parameter_impl(PyObject* const* args, Py_ssize_t nargs) {
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
    const long long n_bpm = PyLong_AsLongLong(args[10]);
    if (PyErr_Occurred()) return nullptr;
    if (!tensor_ok(args[1], "mu", p->code, 7)) return nullptr;
    const at::Tensor& mu = unpack(args[1]);
    if (!tensor_ok(args[2], "cov", p->code, 49, &mu) || !tensor_ok(args[3], "energy", p->code, 1, &mu) || !tensor_ok(args[4], "s", p->code, 1, &mu) ||
        !tensor_ok(args[5], "total_charge", p->code, 1, &mu) || (args[11] != Py_None && !tensor_ok(args[11], "readings", p->code, 2 * n_bpm, &mu)))
        return nullptr;
    if ((n_bpm > 0) != (args[11] != Py_None)) {
        PyErr_SetString(PyExc_ValueError, "readings: a device tensor for every active monitor of the plan (and only then)");
        return nullptr;
    }
    const at::Tensor& cov = unpack(args[2]);
    const at::Tensor& energy = unpack(args[3]);
    const at::Tensor& s_in = unpack(args[4]);
    const at::Tensor& q = unpack(args[5]);
    const double mass = PyFloat_AsDouble(args[6]), nq = PyFloat_AsDouble(args[7]);
    if (PyErr_Occurred()) return nullptr;
    const c10::DeviceGuard device_guard(mu.device());
    void* readings = args[11] == Py_None ? nullptr : unpack(args[11]).data_ptr();
    void* stream = stream_of(mu);
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
            if (width < 1 || height < 1 || !tensor_ok(geom, "screen geometry", p->code, 4, &mu) || !tensor_ok(shift, "screen misalignment", p->code, 2, &mu))
                return nullptr;
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
    bool wrapped = true;
    for (size_t k = 0; k < n_screens; ++k) {
        wrapped = set_wrapped(rec_t, k, recs[k]) && wrapped;
        if (images[k].defined()) {
            wrapped = set_wrapped(img_t, k, images[k]) && wrapped;
        } else {
            Py_INCREF(Py_None);
            PyTuple_SET_ITEM(img_t, k, Py_None);
        }
    }
    PyObject* res = PyTuple_New(6);
    if (!res || !wrapped) {
        Py_XDECREF(res);
        Py_DECREF(rec_t);
        Py_DECREF(img_t);
        return nullptr;
    }
    PyTuple_SET_ITEM(res, 4, rec_t);
    PyTuple_SET_ITEM(res, 5, img_t);
    if (!set_wrapped(res, 0, mu_out) || !set_wrapped(res, 1, cov_out) || !set_wrapped(res, 2, e_out) || !set_wrapped(res, 3, s_out)) {
        Py_DECREF(res);
        return nullptr;
    }
    return res;
}

// (no C++ exception may cross into the interpreter: an allocation failure or a failed check inside ATen becomes a Python error)
template <typename F>
PyObject* guarded(F&& body) {
    try {
        return body();
    } catch (const std::exception& e) {
        PyErr_SetString(g_error ? g_error : PyExc_RuntimeError, e.what());
        return nullptr;
    }
}
PyObject* host_track(PyObject*, PyObject* const* args, Py_ssize_t nargs) { return guarded([&] { return host_track_impl(args, nargs); }); }
PyObject* host_parameter(PyObject*, PyObject* const* args, Py_ssize_t nargs) { return guarded([&] { return host_parameter_impl(args, nargs); }); }

// ---- differentiable nodes in C++ -----------------------------------------------------------------------------------------------
// d(screen sigma_x) / d(quadrupole strength) (tests/test_differentiable.py:10-32 of the reference; BASELINE config C5) is, per step,
// eight launches of ~70 us — and was ~0.25 ms of Python around them: four `torch.autograd.Function.apply` calls forward, their
// `backward` methods called back from the engine's device thread. The two nodes below do the same work from C++ (no GIL, no
// Python frames in the backward pass):
//  RunScreenTrack   [run of linear elements with scalar settings | active Screen]: the stretch call of the forward pass (two
//                   launches: the run's map, the particle pass with the screen's record) as ONE node with outputs (outgoing rows,
//                   the screen's record, the run's composed map C); backward: dC (+ the particle-sized terms dY x^T only when a
//                   gradient arrives through the rows) -> chx_run_build_compose + chx_run_vjp_masked -> the settings' gradients
//                   (element.py:180-191, segment.py:545-574, screen.py:187-214);
//  MomentEntryMapped one beam property (mu_*, sigma_*, cov_*; particle_beam.py:1672-1943) of y = C x as a node on C: forward
//                   chx_moments_entry of y, backward chx_moment_entry_mapped_bwd from the INCOMING beam's moments — no
//                   particle-sized backward pass.
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline void chx_check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed with status ", rc); }

// meta (int64 words): [E, code, n_distinct, kinds[E], ptrs[E * CHX_MAX_PARAMS], then per distinct setting tensor: n, (element,
// slot, index | -1) x n]; mass_eV and n_charges travel as doubles
struct RunDescription {
    int64_t E = 0, n_distinct = 0;
    std::vector<int32_t> kinds;
    std::vector<const void*> ptrs;
    std::vector<size_t> slot_at;        // where distinct setting `pos` starts in meta
    std::vector<uint16_t> need;         // per element: the slots whose derivative is wanted (bit CHX_MAX_PARAMS: the energy)

    // wanted(pos): does distinct setting `pos` need a gradient?
    template <typename Wanted>
    RunDescription(const std::vector<int64_t>& meta, Wanted wanted, bool need_energy) {
        TORCH_CHECK(meta.size() >= 3 && meta[0] >= 1 && static_cast<int64_t>(meta.size()) >= 3 + meta[0] * (1 + CHX_MAX_PARAMS),
                    "malformed plan description");
        E = meta[0];
        n_distinct = meta[2];
        kinds.resize(E);
        ptrs.resize(E * CHX_MAX_PARAMS);
        need.assign(E, 0);
        slot_at.resize(n_distinct);
        for (int64_t e = 0; e < E; ++e) kinds[e] = static_cast<int32_t>(meta[3 + e]);
        for (int64_t k = 0; k < E * CHX_MAX_PARAMS; ++k) ptrs[k] = reinterpret_cast<const void*>(static_cast<uintptr_t>(meta[3 + E + k]));
        size_t at = 3 + E + E * CHX_MAX_PARAMS;
        for (int64_t pos = 0; pos < n_distinct; ++pos) {
            TORCH_CHECK(at < meta.size(), "malformed plan description");
            slot_at[pos] = at;
            const int64_t n = meta[at];
            TORCH_CHECK(n >= 0 && at + 1 + 3 * static_cast<size_t>(n) <= meta.size(), "malformed plan description");
            if (wanted(pos))
                for (int64_t i = 0; i < n; ++i) {
                    const int64_t e = meta[at + 1 + 3 * i], k = meta[at + 2 + 3 * i];
                    TORCH_CHECK(e >= 0 && e < E && k >= 0 && k < CHX_MAX_PARAMS, "malformed plan description");
                    need[e] |= static_cast<uint16_t>(1u << k);
                }
            at += 1 + 3 * n;
        }
        if (need_energy)
            for (auto& m : need) m |= static_cast<uint16_t>(1u << CHX_MAX_PARAMS);
    }

    // the gradient of distinct setting `pos` (the tensor t) out of d[E][CHX_MAX_PARAMS + 1]
    at::Tensor setting_gradient(const std::vector<int64_t>& meta, int64_t pos, const at::Tensor& t, const at::Tensor& d) const {
        const size_t a = slot_at[pos];
        const int64_t n = meta[a];
        at::Tensor g;
        if (t.dim() == 0) {
            g = d.select(0, meta[a + 1]).select(0, meta[a + 2]);
            for (int64_t i = 1; i < n; ++i) g = g + d.select(0, meta[a + 1 + 3 * i]).select(0, meta[a + 2 + 3 * i]);
        } else {
            g = at::zeros_like(t);
            for (int64_t i = 0; i < n; ++i)
                g.select(0, meta[a + 3 + 3 * i]).add_(d.select(0, meta[a + 1 + 3 * i]).select(0, meta[a + 2 + 3 * i]));
        }
        return g;
    }
};

struct RunScreenTrack : public torch::autograd::Function<RunScreenTrack> {
    static variable_list forward(AutogradContext* ctx, const at::Tensor& x, const at::Tensor& energy, const at::Tensor& s_in,
                                 const at::Tensor& charges, const at::Tensor& survival, at::TensorList settings, int64_t plan_addr,
                                 std::vector<int64_t> meta, double mass, double nq) {
        auto* p = reinterpret_cast<StretchPlan*>(static_cast<uintptr_t>(plan_addr));
        TORCH_CHECK(p->screens.size() == 1, "RunScreenTrack: a stretch [run | one active Screen]");
        const auto want = p->code == CHX_F64 ? at::kDouble : at::kFloat;
        TORCH_CHECK(x.is_cuda() && x.dim() == 2 && x.size(1) == 7 && x.size(0) >= 1 && x.is_contiguous() && x.scalar_type() == want,
                    "RunScreenTrack: x must be a contiguous (N, 7) device tensor of the plan's dtype");
        for (const at::Tensor* t : {&energy, &s_in})
            TORCH_CHECK(t->numel() == 1 && t->scalar_type() == want && t->device() == x.device(), "RunScreenTrack: energy / s: one value of x's dtype on its device");
        for (const at::Tensor* t : {&charges, &survival})
            TORCH_CHECK(t->numel() == x.size(0) && t->is_contiguous() && t->scalar_type() == want && t->device() == x.device(),
                        "RunScreenTrack: charges / survival probabilities: contiguous (N,) tensors of x's dtype on its device");
        TORCH_CHECK(meta.size() >= 3 && meta[0] >= 1 && static_cast<int64_t>(meta.size()) >= 3 + meta[0] * (1 + CHX_MAX_PARAMS), "RunScreenTrack: malformed plan description");
        const c10::DeviceGuard device_guard(x.device());
        ctx->set_materialize_grads(false);       // (an output nobody differentiates arrives undefined, not as N x 7 zeros)
        const int64_t N = x.size(0);
        const auto opts = x.options();
        at::Tensor out = at::empty_like(x), e_out = at::empty_like(energy), s_out = at::empty_like(s_in);
        // the screen's record: the rows (differentiable: a loss on the image reaches the map through them) and, in one more tensor,
        // [charges N | survival N | energy | s] (constants of this node)
        const int64_t E = meta[0];
        at::Tensor rows = at::empty({N, 7}, opts), rest = at::empty({2 * N + 2}, opts), C = at::empty({1, 7, 7}, opts);
        at::Tensor maps = at::empty({E, 7, 7}, opts);
        const size_t esize = x.element_size();
        chx_lattice_screen scr = {};
        char* base = static_cast<char*>(rest.data_ptr());
        scr.rows = rows.data_ptr();
        scr.charges = base;
        scr.survival = base + N * esize;
        scr.energy = base + 2 * N * esize;
        scr.s = base + (2 * N + 1) * esize;
        scr.map = C.data_ptr();
        scr.element_maps = maps.data_ptr();
        // the one-pass sums of the recorded beam's moments, a set per workgroup of the particle pass (chx_lattice_screen.mom_partials):
        // a beam property of the screen's beam (MomentEntryMappedNode) is then one small launch instead of a pass over the rows
        at::Tensor sums = at::empty({CHX_LATTICE_MOMENT_DOUBLES}, opts.dtype(at::kDouble));      // (zeroed by the preparation launch)
        scr.mom_partials = sums.data_ptr();
        chx_check(p_track(p->table, p->n_items, p->n_elems, p->n_ptrs, energy.data_ptr(), mass, nq, p->code, p->state, p->state_bytes,
                          x.data_ptr(), out.data_ptr(), N, 1, 1, 1, 1, 0, e_out.data_ptr(), s_in.data_ptr(), s_out.data_ptr(),
                          survival.data_ptr(), nullptr, 0, nullptr, nullptr, 0, charges.data_ptr(), &scr, 1, stream_of(x)),
                  "chx_lattice_track_screens");
        variable_list saved = {x, energy, C, maps};
        for (const at::Tensor& t : settings) saved.push_back(t);
        ctx->save_for_backward(saved);
        ctx->saved_data["meta"] = meta;
        ctx->saved_data["mass"] = mass;
        ctx->saved_data["nq"] = nq;
        ctx->mark_non_differentiable({e_out, s_out, rest, sums, maps});
        return {out, rows, C, e_out, s_out, rest, sums, maps};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const variable_list saved = ctx->get_saved_variables();
        const at::Tensor &x = saved[0], &energy = saved[1], &C = saved[2], &maps = saved[3];
        const std::vector<int64_t> meta = ctx->saved_data["meta"].toIntVector();
        const double mass = ctx->saved_data["mass"].toDouble(), nq = ctx->saved_data["nq"].toDouble();
        const int64_t E = meta[0], n_distinct = meta[2];
        const int code = static_cast<int>(meta[1]);
        const int64_t N = x.size(0);
        void* stream = stream_of(x);
        const auto opts = x.options();
        // dL/dC: what came through C itself (beam properties of the tracked rows) plus sum_n dY_n x_n^T for the rows
        at::Tensor dC = grads[2].defined() ? grads[2].to(opts.dtype()).contiguous() : at::Tensor();
        auto add_rows = [&](const at::Tensor& dY) {
            const size_t ws_bytes = p_apply_bwd_workspace_bytes(1, N);
            at::Tensor ws = at::empty({static_cast<int64_t>(ws_bytes)}, opts.dtype(at::kByte));
            at::Tensor dR = at::empty({49}, opts.dtype(at::kDouble));
            at::Tensor g = dY.contiguous();
            chx_check(p_apply_bwd(g.data_ptr(), C.data_ptr(), x.data_ptr(), nullptr, static_cast<double*>(dR.data_ptr()), 1, 1, 1, N, code,
                                  ws.data_ptr(), ws_bytes, stream),
                      "chx_apply_affine7_bwd");
            at::Tensor part = dR.to(opts.dtype()).reshape({1, 7, 7});
            dC = dC.defined() ? dC + part : part;
        };
        if (grads[0].defined()) add_rows(grads[0]);
        if (grads[1].defined()) add_rows(grads[1]);
        variable_list result(9 + n_distinct);        // x, energy, s_in, charges, survival, settings..., plan, meta, mass, nq
        if (!dC.defined()) return result;
        // the VJP of the builders the wanted settings feed, from the element maps the forward pass's preparation launch left
        const bool need_energy = ctx->needs_input_grad(1);
        const RunDescription run(meta, [&](int64_t pos) { return ctx->needs_input_grad(5 + pos); }, need_energy);
        const size_t ws_bytes = p_run_vjp_workspace_bytes(E);
        at::Tensor ws = at::empty({static_cast<int64_t>(ws_bytes)}, opts.dtype(at::kByte));
        at::Tensor d = at::empty({E, CHX_MAX_PARAMS + 1}, opts);
        chx_check(p_run_vjp_masked(run.kinds.data(), run.ptrs.data(), E, energy.data_ptr(), mass, nq, code, maps.data_ptr(), dC.data_ptr(),
                                   run.need.data(), d.data_ptr(), ws.data_ptr(), ws_bytes, stream),
                  "chx_run_vjp_masked");
        for (int64_t pos = 0; pos < n_distinct; ++pos)
            if (ctx->needs_input_grad(5 + pos)) result[5 + pos] = run.setting_gradient(meta, pos, saved[4 + pos], d);
        if (need_energy) result[1] = d.select(1, CHX_MAX_PARAMS).sum();
        return result;
    }
};

struct MomentEntryMappedNode : public torch::autograd::Function<MomentEntryMappedNode> {
    static variable_list forward(AutogradContext* ctx, const at::Tensor& C, const at::Tensor& y, const std::optional<at::Tensor>& w,
                                 const at::Tensor& mom_x, const std::optional<at::Tensor>& mom_y_in, int64_t index, bool take_sqrt,
                                 const std::optional<at::Tensor>& partials) {
        ctx->set_materialize_grads(false);
        const int64_t N = y.size(-2);
        const auto opts = y.options();
        const int code = code_of(y);
        void* stream = stream_of(y);
        at::Tensor picked = at::empty({}, opts);
        at::Tensor mom_y;
        if (mom_y_in.has_value() && mom_y_in->defined()) {
            mom_y = *mom_y_in;
            chx_check(p_moment_entry(static_cast<const double*>(mom_y.data_ptr()), 1, static_cast<int>(index), take_sqrt ? 1 : 0, code,
                                     picked.data_ptr(), stream),
                      "chx_moment_entry");
        } else if (partials.has_value() && partials->defined()) {
            // y's one-pass sums came out of the particle pass that wrote it (RunScreenTrack): re-centre, add, finalise, pick
            mom_y = at::empty({1, 29}, opts.dtype(at::kDouble));
            chx_check(p_screen_moments(static_cast<const double*>(partials->data_ptr()), p_moment_blocks(N, 1), code,
                                       static_cast<double*>(mom_y.data_ptr()), static_cast<int>(index), take_sqrt ? 1 : 0, picked.data_ptr(),
                                       stream),
                      "chx_lattice_screen_moments");
        } else {
            mom_y = at::empty({1, 29}, opts.dtype(at::kDouble));
            const size_t ws_bytes = p_moments_workspace_bytes(1, N);
            at::Tensor ws = at::empty({static_cast<int64_t>(ws_bytes)}, opts.dtype(at::kByte));
            chx_check(p_moments_entry(y.data_ptr(), w.has_value() && w->defined() ? w->data_ptr() : nullptr, 1, 1, 1, N, code,
                                      static_cast<double*>(mom_y.data_ptr()), static_cast<int>(index), take_sqrt ? 1 : 0, picked.data_ptr(),
                                      ws.data_ptr(), ws_bytes, stream),
                      "chx_moments_entry");
        }
        ctx->save_for_backward({C, mom_y, mom_x});
        ctx->saved_data["index"] = index;
        ctx->saved_data["sqrt"] = take_sqrt;
        ctx->mark_non_differentiable({mom_y});
        return {picked, mom_y};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const variable_list saved = ctx->get_saved_variables();
        const at::Tensor &C = saved[0], &mom_y = saved[1], &mom_x = saved[2];
        variable_list result(8);
        if (!grads[0].defined()) return result;
        at::Tensor g = grads[0].to(C.scalar_type()).contiguous();
        at::Tensor dR = at::empty_like(C);
        chx_check(p_moment_entry_mapped_bwd(g.data_ptr(), static_cast<const double*>(mom_y.data_ptr()),
                                            static_cast<int>(ctx->saved_data["index"].toInt()), ctx->saved_data["sqrt"].toBool() ? 1 : 0,
                                            C.data_ptr(), static_cast<const double*>(mom_x.data_ptr()), 1, 1, mom_x.size(0), code_of(C),
                                            dR.data_ptr(), 0, stream_of(C)),
                  "chx_moment_entry_mapped_bwd");
        result[0] = dR;
        return result;
    }
};

// One beam property of the screen's beam as ONE node on the run's settings: the record's rows are y = C x with C
// the run's composed map, so the property depends on the settings through C alone. MomentEntryMappedNode hangs it on C and leaves the
// way from C to the settings to RunScreenTrack's backward — two nodes, two launches. This node takes the settings themselves: forward
// = the property out of the particle pass's moment sums (or the memoised moments); backward = chx_run_vjp_entry, the builders' VJP
// whose waves form dL/dC from the property's gradient themselves: one node, one launch. C, the element maps and the moments are
// constants of the node (C is a function of the settings and the energy only: there is no other way a gradient could take).
struct RunMomentEntry : public torch::autograd::Function<RunMomentEntry> {
    static variable_list forward(AutogradContext* ctx, const at::Tensor& energy, at::TensorList settings, const at::Tensor& C,
                                 const at::Tensor& mom_x, const std::optional<at::Tensor>& partials, const std::optional<at::Tensor>& mom_y_in,
                                 const at::Tensor& maps, std::vector<int64_t> meta, double mass, double nq, int64_t index, bool take_sqrt,
                                 int64_t N) {
        ctx->set_materialize_grads(false);
        const auto opts = C.options();
        const int code = code_of(C);
        TORCH_CHECK(C.is_cuda() && C.numel() == 49 && C.is_contiguous() && (code == CHX_F64 || C.scalar_type() == at::kFloat),
                    "run_moment_entry: C as a contiguous (1, 7, 7) device tensor of float32 / float64");
        TORCH_CHECK(meta.size() >= 3 && meta[0] >= 1 && meta[1] == code, "run_moment_entry: malformed plan description");
        TORCH_CHECK(maps.is_contiguous() && maps.numel() == meta[0] * 49 && maps.scalar_type() == C.scalar_type() && maps.device() == C.device(),
                    "run_moment_entry: the run's element maps as (E, 7, 7) of C's dtype");
        TORCH_CHECK(mom_x.scalar_type() == at::kDouble && mom_x.is_contiguous() && mom_x.numel() == 29 && mom_x.device() == C.device(),
                    "run_moment_entry: the incoming beam's moments as 29 doubles");
        TORCH_CHECK(energy.numel() == 1 && energy.scalar_type() == C.scalar_type() && energy.device() == C.device(),
                    "run_moment_entry: energy as one value of C's dtype");
        TORCH_CHECK(index >= 2 && index < 29 && N >= 1, "run_moment_entry: entry ", index, " of the 29 moments");
        const c10::DeviceGuard device_guard(C.device());
        void* stream = stream_of(C);
        at::Tensor picked = at::empty({}, opts);
        at::Tensor mom_y;
        if (mom_y_in.has_value() && mom_y_in->defined()) {
            mom_y = *mom_y_in;
            TORCH_CHECK(mom_y.scalar_type() == at::kDouble && mom_y.is_contiguous() && mom_y.numel() == 29 && mom_y.device() == C.device(),
                        "run_moment_entry: the beam's moments as 29 doubles");
            chx_check(p_moment_entry(static_cast<const double*>(mom_y.data_ptr()), 1, static_cast<int>(index), take_sqrt ? 1 : 0, code,
                                     picked.data_ptr(), stream),
                      "chx_moment_entry");
        } else {
            TORCH_CHECK(partials.has_value() && partials->defined() && partials->scalar_type() == at::kDouble && partials->is_contiguous() &&
                            partials->numel() == CHX_LATTICE_MOMENT_DOUBLES && partials->device() == C.device(),
                        "run_moment_entry: the particle pass's moment sums (chx_lattice_screen.mom_partials)");
            mom_y = at::empty({1, 29}, opts.dtype(at::kDouble));
            chx_check(p_screen_moments(static_cast<const double*>(partials->data_ptr()), p_moment_blocks(N, 1), code,
                                       static_cast<double*>(mom_y.data_ptr()), static_cast<int>(index), take_sqrt ? 1 : 0, picked.data_ptr(),
                                       stream),
                      "chx_lattice_screen_moments");
        }
        variable_list saved = {energy, C, mom_y, mom_x, maps};
        for (const at::Tensor& t : settings) saved.push_back(t);
        ctx->save_for_backward(saved);
        ctx->saved_data["meta"] = meta;
        ctx->saved_data["mass"] = mass;
        ctx->saved_data["nq"] = nq;
        ctx->saved_data["index"] = index;
        ctx->saved_data["sqrt"] = take_sqrt;
        ctx->mark_non_differentiable({mom_y});
        return {picked, mom_y};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const variable_list saved = ctx->get_saved_variables();
        const at::Tensor &energy = saved[0], &C = saved[1], &mom_y = saved[2], &mom_x = saved[3], &maps = saved[4];
        const std::vector<int64_t> meta = ctx->saved_data["meta"].toIntVector();
        const int64_t n_distinct = meta[2], E = meta[0];
        variable_list result(1 + n_distinct + 11);   // energy, settings..., C, mom_x, partials, mom_y, maps, meta, mass, nq, index, sqrt, N
        if (!grads[0].defined()) return result;
        const c10::DeviceGuard device_guard(C.device());
        const auto opts = C.options();
        const bool need_energy = ctx->needs_input_grad(0);
        const RunDescription run(meta, [&](int64_t pos) { return ctx->needs_input_grad(1 + pos); }, need_energy);
        at::Tensor g = grads[0].to(C.scalar_type()).contiguous();
        const size_t ws_bytes = p_run_vjp_entry_workspace_bytes(E);
        at::Tensor ws = at::empty({static_cast<int64_t>(ws_bytes)}, opts.dtype(at::kByte));
        at::Tensor d = at::empty({E, CHX_MAX_PARAMS + 1}, opts);
        chx_check(p_run_vjp_entry(run.kinds.data(), run.ptrs.data(), E, energy.data_ptr(), ctx->saved_data["mass"].toDouble(),
                                  ctx->saved_data["nq"].toDouble(), code_of(C), maps.data_ptr(), run.need.data(), g.data_ptr(),
                                  static_cast<const double*>(mom_y.data_ptr()), static_cast<int>(ctx->saved_data["index"].toInt()),
                                  ctx->saved_data["sqrt"].toBool() ? 1 : 0, C.data_ptr(), static_cast<const double*>(mom_x.data_ptr()),
                                  d.data_ptr(), ws.data_ptr(), ws_bytes, stream_of(C)),
                  "chx_run_vjp_entry");
        for (int64_t pos = 0; pos < n_distinct; ++pos)
            if (ctx->needs_input_grad(1 + pos)) result[1 + pos] = run.setting_gradient(meta, pos, saved[5 + pos], d);
        if (need_energy) result[0] = d.select(1, CHX_MAX_PARAMS).sum();
        return result;
    }
};

// run_screen_track(plan, x, energy, s_in, charges, survival, settings tuple, meta (list of ints), mass_eV, n_charges)
//   -> (out, rows at the screen, C (1, 7, 7), charges, survival, energy, s at the screen, moment sums of the rows, element maps)
//      [the first three differentiable in the settings and the energy; the next four are views of one allocation; then mom_partials
//      and the run's (E, 7, 7) element maps: what run_moment_entry takes]
PyObject* host_run_screen_track(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 10) {
        PyErr_SetString(PyExc_TypeError, "run_screen_track takes 10 arguments");
        return nullptr;
    }
    auto* p = static_cast<StretchPlan*>(PyCapsule_GetPointer(args[0], "chx.stretch_plan"));
    if (!p) return nullptr;
    if (!PyTuple_Check(args[6]) || !PyList_Check(args[7])) {
        PyErr_SetString(PyExc_TypeError, "run_screen_track: settings as a tuple of tensors, meta as a list of ints");
        return nullptr;
    }
    std::vector<at::Tensor> settings;
    for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(args[6]); ++i) settings.push_back(unpack(PyTuple_GET_ITEM(args[6], i)));
    std::vector<int64_t> meta(PyList_GET_SIZE(args[7]));
    for (Py_ssize_t i = 0; i < PyList_GET_SIZE(args[7]); ++i) meta[i] = static_cast<int64_t>(PyLong_AsUnsignedLongLongMask(PyList_GET_ITEM(args[7], i)));
    const double mass = PyFloat_AsDouble(args[8]), nq = PyFloat_AsDouble(args[9]);
    if (PyErr_Occurred()) return nullptr;
    try {
        variable_list r = RunScreenTrack::apply(unpack(args[1]), unpack(args[2]), unpack(args[3]), unpack(args[4]), unpack(args[5]),
                                                at::TensorList(settings), static_cast<int64_t>(reinterpret_cast<uintptr_t>(p)), meta, mass, nq);
        PyObject* res = PyTuple_New(9);
        if (!res) return nullptr;
        // the constants of the record as the beam's tensors (views of one allocation, made here: ~0.5 us each against ~2 us from Python)
        const at::Tensor& rest = r[5];
        const int64_t N = r[1].size(0);
        if (!set_wrapped(res, 0, r[0]) || !set_wrapped(res, 1, r[1]) || !set_wrapped(res, 2, r[2]) || !set_wrapped(res, 7, r[6]) ||
            !set_wrapped(res, 8, r[7]) ||
            !set_wrapped(res, 3, rest.narrow(0, 0, N)) || !set_wrapped(res, 4, rest.narrow(0, N, N)) || !set_wrapped(res, 5, rest.select(0, 2 * N)) ||
            !set_wrapped(res, 6, rest.select(0, 2 * N + 1))) {
            Py_DECREF(res);
            return nullptr;
        }
        return res;
    } catch (const std::exception& e) {
        PyErr_SetString(g_error ? g_error : PyExc_RuntimeError, e.what());
        return nullptr;
    }
}

// moment_entry_mapped(C, y, w | None, mom_x, mom_y | None, index, take_sqrt) -> (entry, mom_y)
PyObject* host_moment_entry_mapped(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 8) {
        PyErr_SetString(PyExc_TypeError, "moment_entry_mapped takes 8 arguments");
        return nullptr;
    }
    const long long index = PyLong_AsLongLong(args[5]);
    const int take_sqrt = PyObject_IsTrue(args[6]);
    if (PyErr_Occurred()) return nullptr;
    try {
        std::optional<at::Tensor> w, mom_y, partials;
        if (args[2] != Py_None) w = unpack(args[2]);
        if (args[4] != Py_None) mom_y = unpack(args[4]);
        if (args[7] != Py_None) partials = unpack(args[7]);
        variable_list r = MomentEntryMappedNode::apply(unpack(args[0]), unpack(args[1]), w, unpack(args[3]), mom_y, static_cast<int64_t>(index),
                                                       take_sqrt != 0, partials);
        PyObject* res = PyTuple_New(2);
        if (!res) return nullptr;
        if (!set_wrapped(res, 0, r[0]) || !set_wrapped(res, 1, r[1])) {
            Py_DECREF(res);
            return nullptr;
        }
        return res;
    } catch (const std::exception& e) {
        PyErr_SetString(g_error ? g_error : PyExc_RuntimeError, e.what());
        return nullptr;
    }
}

// run_moment_entry(energy, settings tuple, C, mom_x, partial sums | None, mom_y | None, element maps, meta, mass_eV, n_charges, index,
//                  take_sqrt, N) -> (entry, mom_y)
PyObject* host_run_moment_entry(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 13) {
        PyErr_SetString(PyExc_TypeError, "run_moment_entry takes 13 arguments");
        return nullptr;
    }
    if (!PyTuple_Check(args[1]) || !PyList_Check(args[7])) {
        PyErr_SetString(PyExc_TypeError, "run_moment_entry: settings as a tuple of tensors, meta as a list of ints");
        return nullptr;
    }
    for (const int k : {0, 2, 3, 4, 5, 6})
        if (!(THPVariable_Check(args[k]) || ((k == 4 || k == 5) && args[k] == Py_None))) {
            PyErr_Format(PyExc_TypeError, "run_moment_entry: argument %d: a tensor is expected", k);
            return nullptr;
        }
    for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(args[1]); ++i)
        if (!THPVariable_Check(PyTuple_GET_ITEM(args[1], i))) {
            PyErr_SetString(PyExc_TypeError, "run_moment_entry: settings as a tuple of tensors");
            return nullptr;
        }
    return guarded([&]() -> PyObject* {
        std::vector<at::Tensor> settings;
        for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(args[1]); ++i) settings.push_back(unpack(PyTuple_GET_ITEM(args[1], i)));
        std::vector<int64_t> meta(PyList_GET_SIZE(args[7]));
        for (Py_ssize_t i = 0; i < PyList_GET_SIZE(args[7]); ++i) meta[i] = static_cast<int64_t>(PyLong_AsUnsignedLongLongMask(PyList_GET_ITEM(args[7], i)));
        const double mass = PyFloat_AsDouble(args[8]), nq = PyFloat_AsDouble(args[9]);
        const long long index = PyLong_AsLongLong(args[10]);
        const int take_sqrt = PyObject_IsTrue(args[11]);
        const long long N = PyLong_AsLongLong(args[12]);
        if (PyErr_Occurred()) return nullptr;
        TORCH_CHECK(static_cast<int64_t>(settings.size()) == (meta.size() >= 3 ? meta[2] : -1), "run_moment_entry: one tensor per distinct setting");
        std::optional<at::Tensor> partials, mom_y;
        if (args[4] != Py_None) partials = unpack(args[4]);
        if (args[5] != Py_None) mom_y = unpack(args[5]);
        variable_list r = RunMomentEntry::apply(unpack(args[0]), at::TensorList(settings), unpack(args[2]), unpack(args[3]), partials, mom_y,
                                                unpack(args[6]), meta, mass, nq, static_cast<int64_t>(index), take_sqrt != 0,
                                                static_cast<int64_t>(N));
        PyObject* res = PyTuple_New(2);
        if (!res) return nullptr;
        if (!set_wrapped(res, 0, r[0]) || !set_wrapped(res, 1, r[1])) {
            Py_DECREF(res);
            return nullptr;
        }
        return res;
    });
}

PyMethodDef methods[] = {
    {"any_requires_grad", any_requires_grad, METH_O, "any_requires_grad(tuple_of_tensors) -> bool (non-tensor items count as False)"},
    {"bind", host_bind, METH_VARARGS, "bind({libchx symbol: address}, error class)"},
    {"stretch_plan", host_plan, METH_VARARGS, "stretch_plan(table addr, n_items, n_elems, n_ptrs, state addr, state bytes, dtype code, screens) -> capsule"},
    {"lattice_track_screens", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(host_track)), METH_FASTCALL,
     "lattice_track_screens(plan, x, energy, s_in, charges, survival | None, mass_eV, n_charges, device index, image_limit, survival_out | None, "
     "n_bpm, readings | None, workspace | None, workspace bytes) -> (out, energy_out, s_out, records, images)"},
    {"parameter_lattice_track_screens", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(host_parameter)), METH_FASTCALL,
     "parameter_lattice_track_screens(plan, mu, cov, energy, s_in, total_charge, mass_eV, n_charges, device index, geometries, n_bpm, "
     "readings | None) -> (mu_out, cov_out, energy_out, s_out, records, images)"},
    {"run_screen_track", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(host_run_screen_track)), METH_FASTCALL,
     "run_screen_track(plan, x, energy, s_in, charges, survival, settings, meta, mass_eV, n_charges) -> (out, rows, C, rest): the stretch "
     "[run | active Screen] as one differentiable node"},
    {"moment_entry_mapped", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(host_moment_entry_mapped)), METH_FASTCALL,
     "moment_entry_mapped(C, y, w | None, mom_x, mom_y | None, index, take_sqrt, partial sums | None) -> (entry, mom_y): one beam property of y = C x as a node on C"},
    {"run_moment_entry", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(host_run_moment_entry)), METH_FASTCALL,
     "run_moment_entry(energy, settings, C, mom_x, partial sums | None, mom_y | None, element maps, meta, mass_eV, n_charges, index, take_sqrt, N) "
     "-> (entry, mom_y): one property of the screen's beam as ONE node on the run's settings"},
    {nullptr, nullptr, 0, nullptr}};

struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_chxtorch", "torch-side host step of cheetah_amd (see chx_torch_host.cpp)", -1, methods};

}  // namespace

PyMODINIT_FUNC PyInit__chxtorch(void) { return PyModule_Create(&moddef); }
