// cheetah_amd._chxtorch — the one question the host layer asks torch hundreds of times per track: does any of these tensors
// require grad? `torch._C._any_requires_grad(*tensors)` goes through torch's generic Python argument parser (~23 ns per tensor,
// 7 us for the 300 setting tensors of the 100-element FODO); reading the flag from the THPVariable directly is ~2 ns per tensor.
// Read-only on torch objects; nothing here touches libchx or device memory. Built by csrc/Makefile against the torch headers of
// the running interpreter; when it is missing (a torch upgrade without a rebuild) segment.py falls back to torch._C.
#include <Python.h>

#include <torch/csrc/autograd/python_variable.h>

static PyObject* any_requires_grad(PyObject*, PyObject* seq) {
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

static PyMethodDef methods[] = {{"any_requires_grad", any_requires_grad, METH_O,
                                 "any_requires_grad(tuple_of_tensors) -> bool (non-tensor items count as False)"},
                                {nullptr, nullptr, 0, nullptr}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_chxtorch", "requires_grad scan over a tuple of tensors", -1, methods};
PyMODINIT_FUNC PyInit__chxtorch(void) { return PyModule_Create(&moddef); }
