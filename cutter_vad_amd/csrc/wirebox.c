/*
 * _wirebox: the serving pool's inbox for int16 wire frames, as a small CPython extension (host-side plumbing; no arithmetic).
 *
 * The reference server hands every websocket message to its client's VADWrapper inside the receive loop
 * (websocket_service/server/vad_websocket_server.py:326-382).  Here a frame is only QUEUED when it arrives and all clients' frames
 * reach the engine together at the next tick; what is left on the per-frame path is the cost of the call itself, which in pure
 * Python is ~0.4 us (method dispatch, a lock, a tuple, a list append): 3.5 ms per tick at 8 192 sessions.  This module makes that
 * call a C function - append (slot, bytes object) to an array - and hands the tick ONE array of slots and ONE array of pointers
 * into the bytes objects, which vad_tick_push_gather (include/vad_engine.h) copies straight into the tick's staging rows.
 *
 *   box = Inbox(max_bytes)                 frames longer than max_bytes (or odd-sized, or empty) are not taken
 *   push = box.pusher(slot, gate_on[, rate, nbytes])   a callable bound to one session: push(data) -> True (queued) / False (not
 *                                          taken: the caller goes the general way); push.invalidate() makes every later call return
 *                                          False.  rate != 0: a session whose chunks arrive at another rate and are resampled in the
 *                                          tick (vad_tick_push_rate): only chunks of exactly `nbytes` bytes are taken
 *   len(box)                               frames waiting
 *   box.flush(fn_address, engine_address[, rate_fn_address])   fn = vad_tick_push_gather, rate_fn = vad_tick_push_rate_gather; one
 *                                          call per (frame length, gate, rate) with the GIL released;
 *                                          -> [(slot, status), ...] for the frames the engine refused
 *   box.drain()                            -> [(nbytes, gate, rate, [(slot, data), ...]), ...] and empties the inbox (engines
 *                                          without a C entry point: the test doubles)
 * All methods run under the GIL; a frame's bytes object is kept alive until its flush.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef int (*push_gather_fn)(void *e, const int64_t *slots, int64_t n, const void *const *frames, int32_t nsamples, int frame_fmt,
                              int gate_on, int32_t *status);

typedef int (*push_rate_gather_fn)(void *e, const int64_t *slots, int64_t n, const void *const *frames, int32_t nsamples, int frame_fmt,
                                   int gate_on, int32_t sr_in, int32_t *status);

typedef struct {
    int32_t nbytes;
    int gate;
    int32_t rate;            /* 0: frames at the engine's rate */
    int64_t *slots;
    PyObject **data;
    Py_ssize_t n, cap;
} Box;

typedef struct {
    PyObject_HEAD
    Box *boxes;
    Py_ssize_t nboxes;
    Py_ssize_t total;
    int32_t max_bytes;
    uint64_t epoch;          /* bumped when a flush / drain starts */
} Inbox;

typedef struct {
    PyObject_HEAD
    Inbox *inbox;
    int64_t slot;
    int gate;
    int32_t rate;            /* 0, or the session's input rate ... */
    int32_t want_bytes;      /* ... and the one chunk length it may send */
    int valid;
    Py_ssize_t box;          /* index of the box this session last used (its frames have one length almost always) */
    uint64_t epoch;          /* the inbox epoch of this session's last frame ... */
    int32_t nbytes;          /* ... and its length: see pusher_vectorcall */
    vectorcallfunc vectorcall;
} Pusher;

static PyTypeObject InboxType, PusherType;

static void box_release(Box *b) {
    for (Py_ssize_t i = 0; i < b->n; ++i) Py_DECREF(b->data[i]);
    b->n = 0;
}

static Box *inbox_box(Inbox *self, int32_t nbytes, int gate, int32_t rate, Py_ssize_t *hint) {
    if (*hint >= 0 && *hint < self->nboxes && self->boxes[*hint].nbytes == nbytes && self->boxes[*hint].gate == gate &&
        self->boxes[*hint].rate == rate)
        return &self->boxes[*hint];
    for (Py_ssize_t k = 0; k < self->nboxes; ++k)
        if (self->boxes[k].nbytes == nbytes && self->boxes[k].gate == gate && self->boxes[k].rate == rate) {
            *hint = k;
            return &self->boxes[k];
        }
    Box *nb = (Box *)PyMem_Realloc(self->boxes, (size_t)(self->nboxes + 1) * sizeof(Box));
    if (!nb) return NULL;
    self->boxes = nb;
    Box *b = &nb[self->nboxes];
    memset(b, 0, sizeof *b);
    b->nbytes = nbytes;
    b->gate = gate;
    b->rate = rate;
    *hint = self->nboxes++;
    return b;
}

static int box_append(Box *b, int64_t slot, PyObject *data) {
    if (b->n == b->cap) {
        const Py_ssize_t cap = b->cap ? 2 * b->cap : 256;
        int64_t *s = (int64_t *)PyMem_Realloc(b->slots, (size_t)cap * sizeof(int64_t));
        if (!s) return -1;
        b->slots = s;
        PyObject **d = (PyObject **)PyMem_Realloc(b->data, (size_t)cap * sizeof(PyObject *));
        if (!d) return -1;
        b->data = d;
        b->cap = cap;
    }
    b->slots[b->n] = slot;
    Py_INCREF(data);
    b->data[b->n] = data;
    b->n += 1;
    return 0;
}

/* ---- Pusher ------------------------------------------------------------------------------------------------------------- */
static PyObject *pusher_vectorcall(PyObject *callable, PyObject *const *args, size_t nargsf, PyObject *kwnames) {
    Pusher *self = (Pusher *)callable;
    if (PyVectorcall_NARGS(nargsf) != 1 || (kwnames && PyTuple_GET_SIZE(kwnames))) {
        PyErr_SetString(PyExc_TypeError, "push(data) takes exactly one positional argument");
        return NULL;
    }
    PyObject *data = args[0];
    if (!self->valid || !PyBytes_CheckExact(data)) Py_RETURN_FALSE;
    const Py_ssize_t nb = PyBytes_GET_SIZE(data);
    if (nb < 2 || (nb & 1)) Py_RETURN_FALSE;
    if (self->rate ? nb != self->want_bytes : nb > self->inbox->max_bytes) Py_RETURN_FALSE;
    /* frames are grouped by length for the engine, and the groups go one after the other: two frames of ONE session with different
     * lengths (the last chunk of a file) must not wait in the same flush, or the later one could overtake.  The second one is not
     * taken; the general path flushes first. */
    if (self->epoch == self->inbox->epoch && self->nbytes != (int32_t)nb) Py_RETURN_FALSE;
    Box *b = inbox_box(self->inbox, (int32_t)nb, self->gate, self->rate, &self->box);
    if (!b || box_append(b, self->slot, data) < 0) return PyErr_NoMemory();
    self->inbox->total += 1;
    self->epoch = self->inbox->epoch;
    self->nbytes = (int32_t)nb;
    Py_RETURN_TRUE;
}

static PyObject *pusher_invalidate(Pusher *self, PyObject *Py_UNUSED(ignored)) {
    self->valid = 0;
    Py_RETURN_NONE;
}

static PyObject *pusher_get_valid(Pusher *self, void *Py_UNUSED(c)) { return PyBool_FromLong(self->valid); }

static void pusher_dealloc(Pusher *self) {
    Py_XDECREF((PyObject *)self->inbox);
    Py_TYPE(self)->tp_free((PyObject *)self);
}

static PyMethodDef pusher_methods[] = {
    {"invalidate", (PyCFunction)pusher_invalidate, METH_NOARGS, "every later call returns False (the session was closed, is moving, or was reconfigured)"},
    {NULL, NULL, 0, NULL}};
static PyGetSetDef pusher_getset[] = {{"valid", (getter)pusher_get_valid, NULL, "still bound to a live session", NULL}, {NULL, NULL, NULL, NULL, NULL}};

/* ---- Inbox -------------------------------------------------------------------------------------------------------------- */
static int inbox_init(Inbox *self, PyObject *args, PyObject *kwds) {
    static char *kw[] = {"max_bytes", NULL};
    int max_bytes = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kwds, "i", kw, &max_bytes)) return -1;
    if (max_bytes < 2) {
        PyErr_SetString(PyExc_ValueError, "max_bytes must be at least 2");
        return -1;
    }
    self->max_bytes = max_bytes;
    return 0;
}

static void inbox_dealloc(Inbox *self) {
    for (Py_ssize_t k = 0; k < self->nboxes; ++k) {
        box_release(&self->boxes[k]);
        PyMem_Free(self->boxes[k].slots);
        PyMem_Free(self->boxes[k].data);
    }
    PyMem_Free(self->boxes);
    Py_TYPE(self)->tp_free((PyObject *)self);
}

static PyObject *inbox_pusher(Inbox *self, PyObject *args) {
    long long slot;
    int gate, rate = 0, want = 0;
    if (!PyArg_ParseTuple(args, "Lp|ii", &slot, &gate, &rate, &want)) return NULL;
    if (rate && (want < 2 || (want & 1))) {
        PyErr_SetString(PyExc_ValueError, "pusher(slot, gate_on, rate, nbytes): a rate needs its chunk length in bytes (even, >= 2)");
        return NULL;
    }
    Pusher *p = PyObject_New(Pusher, &PusherType);
    if (!p) return NULL;
    Py_INCREF((PyObject *)self);
    p->inbox = self;
    p->slot = (int64_t)slot;
    p->gate = gate ? 1 : 0;
    p->rate = rate;
    p->want_bytes = want;
    p->valid = 1;
    p->box = -1;
    p->epoch = (uint64_t)-1;
    p->nbytes = 0;
    p->vectorcall = pusher_vectorcall;
    return (PyObject *)p;
}

static Py_ssize_t inbox_len(Inbox *self) { return self->total; }

static PyObject *inbox_drain(Inbox *self, PyObject *Py_UNUSED(ignored)) {
    PyObject *out = PyList_New(0);
    if (!out) return NULL;
    self->epoch += 1;
    for (Py_ssize_t k = 0; k < self->nboxes; ++k) {
        Box *b = &self->boxes[k];
        if (!b->n) continue;
        PyObject *items = PyList_New(b->n);
        if (!items) goto fail;
        for (Py_ssize_t i = 0; i < b->n; ++i) {
            PyObject *t = Py_BuildValue("(LO)", (long long)b->slots[i], b->data[i]);
            if (!t) {
                Py_DECREF(items);
                goto fail;
            }
            PyList_SET_ITEM(items, i, t);
        }
        PyObject *row = Py_BuildValue("(iOiN)", (int)b->nbytes, b->gate ? Py_True : Py_False, (int)b->rate, items);
        if (!row || PyList_Append(out, row) < 0) {
            Py_XDECREF(row);
            goto fail;
        }
        Py_DECREF(row);
        self->total -= b->n;
        box_release(b);
    }
    return out;
fail:
    Py_DECREF(out);
    return NULL;
}

static PyObject *inbox_flush(Inbox *self, PyObject *args) {
    unsigned long long fn_addr, eng_addr, rate_fn_addr = 0;
    if (!PyArg_ParseTuple(args, "KK|K", &fn_addr, &eng_addr, &rate_fn_addr)) return NULL;
    if (!fn_addr || !eng_addr) {
        PyErr_SetString(PyExc_ValueError, "flush(fn_address, engine_address): both must be non-zero");
        return NULL;
    }
    push_gather_fn fn = (push_gather_fn)(uintptr_t)fn_addr;
    push_rate_gather_fn rate_fn = (push_rate_gather_fn)(uintptr_t)rate_fn_addr;
    for (Py_ssize_t k = 0; k < self->nboxes; ++k)
        if (self->boxes[k].n && self->boxes[k].rate && !rate_fn) {
            PyErr_SetString(PyExc_ValueError, "flush: chunks at another rate are waiting and no rate entry point was given");
            return NULL;
        }
    PyObject *fails = PyList_New(0);
    if (!fails) return NULL;
    /* EVERY box's arrays are taken out before the GIL is released for the first time: a frame that arrives while one group is in
     * the engine starts a fresh array and waits for the NEXT flush, whatever its box.  (Walking the live boxes instead let such a
     * frame - it passes the epoch guard of pusher_vectorcall, its session's last frame carries the previous epoch - leave in this
     * flush from a box not yet visited, ahead of the session's earlier frame in a box visited later.) */
    const Py_ssize_t nsnap = self->nboxes;
    Box *snap = (Box *)PyMem_Calloc((size_t)(nsnap ? nsnap : 1), sizeof(Box));
    if (!snap) {
        Py_DECREF(fails);
        return PyErr_NoMemory();
    }
    self->epoch += 1;
    for (Py_ssize_t k = 0; k < nsnap; ++k) {
        Box *b = &self->boxes[k];
        snap[k] = *b;
        b->slots = NULL; b->data = NULL; b->n = 0; b->cap = 0;
        self->total -= snap[k].n;
    }
    int ok = 1;
    for (Py_ssize_t k = 0; k < nsnap; ++k) {
        int64_t *slots = snap[k].slots;
        PyObject **data = snap[k].data;
        const Py_ssize_t n = snap[k].n;
        if (n && ok) {
            const int32_t nbytes = snap[k].nbytes;
            const int gate = snap[k].gate;
            const int32_t rate = snap[k].rate;
            const void **ptrs = (const void **)PyMem_Malloc((size_t)n * sizeof(void *));
            int32_t *status = (int32_t *)PyMem_Calloc((size_t)n, sizeof(int32_t));
            ok = ptrs && status;
            if (ok) {
                for (Py_ssize_t i = 0; i < n; ++i) ptrs[i] = PyBytes_AS_STRING(data[i]);
                Py_BEGIN_ALLOW_THREADS
                if (rate) (void)rate_fn((void *)(uintptr_t)eng_addr, slots, (int64_t)n, ptrs, nbytes / 2, /* VAD_FMT_I16_32767 */ 1, gate, rate, status);
                else (void)fn((void *)(uintptr_t)eng_addr, slots, (int64_t)n, ptrs, nbytes / 2, /* VAD_FMT_I16_32767 */ 1, gate, status);
                Py_END_ALLOW_THREADS
                for (Py_ssize_t i = 0; i < n && ok; ++i)
                    if (status[i] != 0) {
                        PyObject *t = Py_BuildValue("(Li)", (long long)slots[i], (int)status[i]);
                        if (!t || PyList_Append(fails, t) < 0) ok = 0;
                        Py_XDECREF(t);
                    }
            }
            PyMem_Free(ptrs);
            PyMem_Free(status);
        }
        for (Py_ssize_t i = 0; i < n; ++i) Py_DECREF(data[i]);      /* also the groups behind a failure: their frames are dropped */
        /* hand the (empty) arrays back if nothing arrived meanwhile, so that a steady stream of ticks does not reallocate
         * (self->boxes may have moved while the GIL was away: index, never a kept pointer) */
        Box *b = &self->boxes[k];
        if (b->slots == NULL && b->data == NULL) {
            b->slots = slots;
            b->data = data;
            b->cap = snap[k].cap;
        } else {
            PyMem_Free(slots);
            PyMem_Free(data);
        }
    }
    PyMem_Free(snap);
    if (!ok) {
        Py_DECREF(fails);
        return PyErr_NoMemory();
    }
    return fails;
}

static PyMethodDef inbox_methods[] = {
    {"pusher", (PyCFunction)inbox_pusher, METH_VARARGS, "pusher(slot, gate_on[, rate, nbytes]) -> callable push(data) -> bool"},
    {"flush", (PyCFunction)inbox_flush, METH_VARARGS, "flush(fn_address, engine_address[, rate_fn_address]) -> [(slot, status), ...] of refused frames"},
    {"drain", (PyCFunction)inbox_drain, METH_NOARGS, "drain() -> [(nbytes, gate, rate, [(slot, data), ...]), ...]; empties the inbox"},
    {NULL, NULL, 0, NULL}};
static PySequenceMethods inbox_as_sequence = {.sq_length = (lenfunc)inbox_len};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_wirebox", "inbox for int16 wire frames (see csrc/wirebox.c)", -1, NULL, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__wirebox(void) {
    InboxType = (PyTypeObject){PyVarObject_HEAD_INIT(NULL, 0)};
    InboxType.tp_name = "_wirebox.Inbox";
    InboxType.tp_basicsize = sizeof(Inbox);
    InboxType.tp_flags = Py_TPFLAGS_DEFAULT;
    InboxType.tp_new = PyType_GenericNew;
    InboxType.tp_init = (initproc)inbox_init;
    InboxType.tp_dealloc = (destructor)inbox_dealloc;
    InboxType.tp_methods = inbox_methods;
    InboxType.tp_as_sequence = &inbox_as_sequence;
    PusherType = (PyTypeObject){PyVarObject_HEAD_INIT(NULL, 0)};
    PusherType.tp_name = "_wirebox.Pusher";
    PusherType.tp_basicsize = sizeof(Pusher);
    PusherType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL;
    PusherType.tp_dealloc = (destructor)pusher_dealloc;
    PusherType.tp_methods = pusher_methods;
    PusherType.tp_getset = pusher_getset;
    PusherType.tp_call = PyVectorcall_Call;
    PusherType.tp_vectorcall_offset = offsetof(Pusher, vectorcall);
    if (PyType_Ready(&InboxType) < 0 || PyType_Ready(&PusherType) < 0) return NULL;
    PyObject *m = PyModule_Create(&moduledef);
    if (!m) return NULL;
    Py_INCREF(&InboxType);
    Py_INCREF(&PusherType);
    if (PyModule_AddObject(m, "Inbox", (PyObject *)&InboxType) < 0 || PyModule_AddObject(m, "Pusher", (PyObject *)&PusherType) < 0) {
        Py_DECREF(m);
        return NULL;
    }
    return m;
}
