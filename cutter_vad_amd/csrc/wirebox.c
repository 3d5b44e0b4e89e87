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
 *   push = box.pusher(slot, gate_on[, rate, nbytes[, fallback]])   a callable bound to one session: push(data) -> True (queued) /
 *                                          False (not taken: the caller goes the general way); push.invalidate() makes every later
 *                                          call return False.  rate != 0: a session whose chunks arrive at another rate and are
 *                                          resampled in the tick (vad_tick_push_rate): only chunks of exactly `nbytes` bytes are
 *                                          taken.  fallback: a callable - then a frame that is not taken is handed to
 *                                          fallback(data) and ITS result returned, so that the pusher can BE the session's
 *                                          submit method (one C call per wire frame, no interpreter frame in between)
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
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "vad_engine.h"          /* types only (vad_tick_result, vad_tick_work): the engine is reached through function addresses */

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
    PyObject *fallback;      /* NULL, or where a frame that is not taken goes (its result is returned) */
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
    Py_ssize_t nb = 0;
    int take = self->valid && PyBytes_CheckExact(data);
    if (take) {
        nb = PyBytes_GET_SIZE(data);
        take = nb >= 2 && !(nb & 1) && (self->rate ? nb == self->want_bytes : nb <= self->inbox->max_bytes);
    }
    /* frames are grouped by length for the engine, and the groups go one after the other: two frames of ONE session with different
     * lengths (the last chunk of a file) must not wait in the same flush, or the later one could overtake.  The second one is not
     * taken; the general path flushes first. */
    if (take && self->epoch == self->inbox->epoch && self->nbytes != (int32_t)nb) take = 0;
    if (!take) {
        if (self->fallback) return PyObject_CallOneArg(self->fallback, data);
        Py_RETURN_FALSE;
    }
    Box *b = inbox_box(self->inbox, (int32_t)nb, self->gate, self->rate, &self->box);
    if (!b || box_append(b, self->slot, data) < 0) return PyErr_NoMemory();
    self->inbox->total += 1;
    self->epoch = self->inbox->epoch;
    self->nbytes = (int32_t)nb;
    Py_RETURN_TRUE;
}

static PyObject *pusher_invalidate(Pusher *self, PyObject *Py_UNUSED(ignored)) {
    self->valid = 0;                             /* (the fallback stays: an invalidated pusher hands everything to it) */
    Py_RETURN_NONE;
}

static int pusher_traverse(Pusher *self, visitproc visit, void *arg) {
    Py_VISIT((PyObject *)self->inbox);
    Py_VISIT(self->fallback);
    return 0;
}

static int pusher_clear(Pusher *self) {
    Py_CLEAR(self->fallback);                     /* the session <-> pusher <-> bound-method cycle is broken here */
    return 0;
}

static PyObject *pusher_get_valid(Pusher *self, void *Py_UNUSED(c)) { return PyBool_FromLong(self->valid); }

static void pusher_dealloc(Pusher *self) {
    PyObject_GC_UnTrack(self);
    Py_CLEAR(self->fallback);
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
    PyObject *fallback = NULL;
    if (!PyArg_ParseTuple(args, "Lp|iiO", &slot, &gate, &rate, &want, &fallback)) return NULL;
    if (fallback == Py_None) fallback = NULL;
    if (fallback && !PyCallable_Check(fallback)) {
        PyErr_SetString(PyExc_TypeError, "pusher(..., fallback): fallback must be callable");
        return NULL;
    }
    if (rate && (want < 2 || (want & 1))) {
        PyErr_SetString(PyExc_ValueError, "pusher(slot, gate_on, rate, nbytes): a rate needs its chunk length in bytes (even, >= 2)");
        return NULL;
    }
    Pusher *p = PyObject_GC_New(Pusher, &PusherType);
    if (!p) return NULL;
    Py_XINCREF(fallback);
    p->fallback = fallback;
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
    PyObject_GC_Track((PyObject *)p);
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

/* ---- module functions: two more per-event costs of the tick's fan-out taken out of the interpreter ---------------------------- */

/* call_each(callbacks, slots, arg) -> [(slot, exception), ...]
 * callbacks: list indexed by slot (None = nobody listens); slots: a contiguous int64 buffer (a numpy array).  Calls
 * callbacks[slot](arg) for every slot in order - the voice_continue notifications of a tick, one per talking session and frame
 * (the reference server sends each of them: vad_websocket_server.py:420-430) - and collects what raised instead of stopping. */
static PyObject *wirebox_call_each(PyObject *Py_UNUSED(mod), PyObject *args) {
    PyObject *cbs, *slots_obj, *arg;
    if (!PyArg_ParseTuple(args, "O!OO", &PyList_Type, &cbs, &slots_obj, &arg)) return NULL;
    Py_buffer view;
    if (PyObject_GetBuffer(slots_obj, &view, PyBUF_CONTIG_RO | PyBUF_FORMAT) < 0) return NULL;
    if (view.itemsize != 8 || view.ndim > 1) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_TypeError, "call_each: slots must be a contiguous one-dimensional int64 buffer");
        return NULL;
    }
    PyObject *fails = PyList_New(0);
    if (!fails) {
        PyBuffer_Release(&view);
        return NULL;
    }
    const int64_t *sl = (const int64_t *)view.buf;
    const Py_ssize_t n = view.len / 8;
    for (Py_ssize_t k = 0; k < n; ++k) {
        const int64_t i = sl[k];
        if (i < 0 || i >= PyList_GET_SIZE(cbs)) continue;           /* (the list may shrink under a callback: checked every time) */
        PyObject *cb = PyList_GET_ITEM(cbs, (Py_ssize_t)i);
        if (cb == Py_None) continue;
        Py_INCREF(cb);
        PyObject *r = PyObject_CallOneArg(cb, arg);
        Py_DECREF(cb);
        if (r) {
            Py_DECREF(r);
            continue;
        }
        PyObject *et, *ev, *tb;
        PyErr_Fetch(&et, &ev, &tb);
        PyErr_NormalizeException(&et, &ev, &tb);
        PyObject *t = Py_BuildValue("(LO)", (long long)i, ev ? ev : Py_None);
        Py_XDECREF(et); Py_XDECREF(ev); Py_XDECREF(tb);
        if (!t || PyList_Append(fails, t) < 0) {
            Py_XDECREF(t);
            Py_DECREF(fails);
            PyBuffer_Release(&view);
            return NULL;
        }
        Py_DECREF(t);
    }
    PyBuffer_Release(&view);
    return fails;
}

/* take_wav16(fn_address, engine_address, slot, sample_rate, nsamples) -> bytes
 * fn = vad_tick_take_segment_wav16 (include/vad_engine.h): the finished segment as the voice_end payload, written by the engine
 * straight into the bytes object (GIL released while it converts); nsamples from vad_tick_work.work_samples. */
typedef int (*take_wav16_fn)(void *e, int64_t slot, int32_t sample_rate, void *out, int64_t cap, int64_t *nbytes);

static PyObject *wirebox_take_wav16(PyObject *Py_UNUSED(mod), PyObject *args) {
    unsigned long long fn_addr, eng_addr;
    long long slot, nsamples;
    int rate;
    if (!PyArg_ParseTuple(args, "KKLiL", &fn_addr, &eng_addr, &slot, &rate, &nsamples)) return NULL;
    if (!fn_addr || !eng_addr || nsamples < 0) {
        PyErr_SetString(PyExc_ValueError, "take_wav16(fn_address, engine_address, slot, sample_rate, nsamples)");
        return NULL;
    }
    const Py_ssize_t nb = 44 + 2 * (Py_ssize_t)nsamples;
    PyObject *out = PyBytes_FromStringAndSize(NULL, nb);
    if (!out) return NULL;
    int64_t got = 0;
    int rc;
    char *buf = PyBytes_AS_STRING(out);
    Py_BEGIN_ALLOW_THREADS
    rc = ((take_wav16_fn)(uintptr_t)fn_addr)((void *)(uintptr_t)eng_addr, (int64_t)slot, (int32_t)rate, buf, (int64_t)nb, &got);
    Py_END_ALLOW_THREADS
    if (rc != 0 || got != (int64_t)nb) {
        Py_DECREF(out);
        PyErr_Format(PyExc_RuntimeError, "vad_tick_take_segment_wav16 failed (status %d, %lld of %lld bytes)", rc, (long long)got, (long long)nb);
        return NULL;
    }
    return out;
}

/* take_wav16_many(fn_address, engine_address, slots, rates, nsamples) -> [bytes, ...]
 * The same for ALL the segments that ended in a tick, with the GIL released ONCE: slots / nsamples contiguous int64 buffers, rates a
 * contiguous int32 buffer, one entry per segment.  A tick's voice_end payloads then cost one trip through the interpreter lock
 * instead of one per segment - with a ticker thread per GPU that lock is what the threads queue for. */
static PyObject *wirebox_take_wav16_many(PyObject *Py_UNUSED(mod), PyObject *args) {
    unsigned long long fn_addr, eng_addr;
    PyObject *so, *ro, *no;
    if (!PyArg_ParseTuple(args, "KKOOO", &fn_addr, &eng_addr, &so, &ro, &no)) return NULL;
    Py_buffer sv, rv, nv;
    if (PyObject_GetBuffer(so, &sv, PyBUF_CONTIG_RO) < 0) return NULL;
    if (PyObject_GetBuffer(ro, &rv, PyBUF_CONTIG_RO) < 0) { PyBuffer_Release(&sv); return NULL; }
    if (PyObject_GetBuffer(no, &nv, PyBUF_CONTIG_RO) < 0) { PyBuffer_Release(&sv); PyBuffer_Release(&rv); return NULL; }
    PyObject *out = NULL;
    char **bufs = NULL;
    const Py_ssize_t n = sv.len / 8;
    if (!fn_addr || !eng_addr || sv.itemsize != 8 || nv.itemsize != 8 || rv.itemsize != 4 || nv.len / 8 != n || rv.len / 4 != n) {
        PyErr_SetString(PyExc_ValueError, "take_wav16_many(fn_address, engine_address, slots int64[n], rates int32[n], nsamples int64[n])");
        goto done;
    }
    const int64_t *slots = (const int64_t *)sv.buf, *ns = (const int64_t *)nv.buf;
    const int32_t *rates = (const int32_t *)rv.buf;
    out = PyList_New(n);
    bufs = (char **)PyMem_Malloc((size_t)(n ? n : 1) * sizeof(char *));
    if (!out || !bufs) { Py_CLEAR(out); PyErr_NoMemory(); goto done; }
    for (Py_ssize_t k = 0; k < n; ++k) {
        PyObject *b = ns[k] >= 0 ? PyBytes_FromStringAndSize(NULL, 44 + 2 * (Py_ssize_t)ns[k]) : NULL;
        if (!b) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "take_wav16_many: negative sample count"); Py_CLEAR(out); goto done; }
        PyList_SET_ITEM(out, k, b);
        bufs[k] = PyBytes_AS_STRING(b);
    }
    {
        int bad = 0;
        Py_ssize_t at = -1;
        Py_BEGIN_ALLOW_THREADS
        for (Py_ssize_t k = 0; k < n; ++k) {
            int64_t got = 0;
            const int64_t nb = 44 + 2 * ns[k];
            const int rc = ((take_wav16_fn)(uintptr_t)fn_addr)((void *)(uintptr_t)eng_addr, slots[k], rates[k], bufs[k], nb, &got);
            if ((rc != 0 || got != nb) && !bad) { bad = rc ? rc : -1; at = k; }
        }
        Py_END_ALLOW_THREADS
        if (bad) {
            Py_CLEAR(out);
            PyErr_Format(PyExc_RuntimeError, "vad_tick_take_segment_wav16 failed for entry %zd (status %d)", at, bad);
        }
    }
done:
    PyMem_Free(bufs);
    PyBuffer_Release(&sv); PyBuffer_Release(&rv); PyBuffer_Release(&nv);
    return out;
}

/* ---- tick_shards: every engine's tick of one round, side by side, behind ONE release of the interpreter lock ---------------------
 * A serving process with a pool per GPU (ShardedStreamPool) used to run one Python ticker thread per pool.  Each tick gives the
 * lock up and takes it back several times (inbox flush, vad_tick_run, the segments' payloads); with eight of them the threads
 * spend the round queueing for the lock (rehearsed at 65 536 sessions: 0.4 x real time, profiles/r04_server_65536_fake.jsonl).
 * Here ONE thread conducts the round:
 *   1. under the lock: every shard's inbox hands over what has arrived (as Inbox.flush does);
 *   2. lock released: one C thread per shard pushes its frames into its engine (vad_tick_push_gather / _rate_gather) and runs the
 *      tick with its bookkeeping (vad_tick_run_work);
 *   3. under the lock: refusals are collected, and a bytes object is made for every segment that ended (size from
 *      vad_tick_work.work_samples) whose session takes 16-bit mono WAV payloads;
 *   4. lock released: one C thread per shard writes those payloads (vad_tick_take_segment_wav16);
 *   5. under the lock: per shard (status of the tick, refused frames, payloads) go back to the caller, which fans the events out.
 * tick_shards(jobs) -> [(rc, [(slot, status), ...], [(work entry, bytes), ...]), ...]
 *   job = (inbox | None, push_fn, engine, rate_fn, run_fn, denoise_thresh, address of vad_tick_result, address of vad_tick_work,
 *          wav_fn | 0, wav_rates | None)
 *   wav_rates: contiguous int32 buffer indexed by slot - the session's WAV sample rate if its voice_end payload is 16-bit mono,
 *   else 0 (the caller builds that one itself).  The two structs are the caller's (ctypes), filled in as by vad_tick_run_work. */
typedef int (*run_work_fn)(void *e, float thr, vad_tick_result *out, vad_tick_work *work);

typedef struct ShardJob {
    Inbox *inbox;
    Box *snap;
    Py_ssize_t nsnap;
    const void ***ptrs;          /* per snapshot box: the frames' addresses */
    int32_t **status;            /* per snapshot box: the engine's answer per frame */
    push_gather_fn fn;
    push_rate_gather_fn rate_fn;
    void *eng;
    run_work_fn run;
    float thr;
    vad_tick_result *res;
    vad_tick_work *work;
    int rc;
    take_wav16_fn wav;
    Py_buffer rates;             /* int32 per slot; rates.buf == NULL: no payloads built here */
    Py_ssize_t n_wav;
    int64_t *wav_slot, *wav_ns;
    int32_t *wav_sr;
    char **wav_buf;
    Py_ssize_t *wav_entry;
    int wav_bad;
} ShardJob;

static void shard_push_and_run(ShardJob *j) {
    for (Py_ssize_t k = 0; k < j->nsnap; ++k) {
        const Box *b = &j->snap[k];
        if (!b->n || !j->ptrs[k]) continue;
        if (b->rate) (void)j->rate_fn(j->eng, b->slots, (int64_t)b->n, j->ptrs[k], b->nbytes / 2, /* VAD_FMT_I16_32767 */ 1, b->gate, b->rate, j->status[k]);
        else (void)j->fn(j->eng, b->slots, (int64_t)b->n, j->ptrs[k], b->nbytes / 2, 1, b->gate, j->status[k]);
    }
    j->rc = j->run(j->eng, j->thr, j->res, j->work);
}

static void shard_write_wavs(ShardJob *j) {
    struct timespec a_, b_;
    const int timing = getenv("VAD_WIREBOX_TIMING") != NULL;
    if (timing) clock_gettime(CLOCK_MONOTONIC, &a_);
    for (Py_ssize_t k = 0; k < j->n_wav; ++k) {
        int64_t got = 0;
        const int64_t nb = 44 + 2 * j->wav_ns[k];
        const int rc = j->wav(j->eng, j->wav_slot[k], j->wav_sr[k], j->wav_buf[k], nb, &got);
        if ((rc != 0 || got != nb) && !j->wav_bad) j->wav_bad = rc ? rc : -1;
    }
    if (timing) {
        clock_gettime(CLOCK_MONOTONIC, &b_);
        fprintf(stderr, "  wavs of one shard: %zd payloads in %.2f ms\n", j->n_wav, (b_.tv_sec - a_.tv_sec) * 1e3 + (b_.tv_nsec - a_.tv_nsec) * 1e-6);
    }
}

typedef struct { void (*f)(ShardJob *); ShardJob *job; } ShardCall;
static void *shard_thread(void *a) {
    ShardCall *c = (ShardCall *)a;
    c->f(c->job);
    return NULL;
}

/* f(job) for every job, side by side: the first on this thread, job k on worker k - a small crew of threads that is started
 * once and sleeps between rounds (a thread created per job and round cost ~25 us each and started on a cold core every time).
 * The crew serves one caller at a time; a second conductor in the process, or a thread that cannot be started, falls back to
 * threads of its own / to running the job here. */
typedef struct CrewWorker {
    pthread_t th;
    int started;
    void (*f)(ShardJob *);
    ShardJob *job;
    uint64_t posted, taken;      /* rounds handed over / picked up */
} CrewWorker;
static CrewWorker g_crew[64];
static pthread_mutex_t g_crew_m = PTHREAD_MUTEX_INITIALIZER;       /* guards every field of the workers */
static pthread_cond_t g_crew_go = PTHREAD_COND_INITIALIZER, g_crew_done = PTHREAD_COND_INITIALIZER;
static pthread_mutex_t g_crew_busy = PTHREAD_MUTEX_INITIALIZER;    /* one conductor at a time */
static int g_crew_pending = 0;

/* A forked child has none of the crew's threads (fork copies the calling thread only) but would inherit their bookkeeping, post
 * rounds to workers that do not exist and wait for them for ever: the child starts with an empty crew and fresh locks. */
static void crew_atfork_child(void) {
    memset(g_crew, 0, sizeof g_crew);
    g_crew_pending = 0;
    pthread_mutex_init(&g_crew_m, NULL);
    pthread_mutex_init(&g_crew_busy, NULL);
    pthread_cond_init(&g_crew_go, NULL);
    pthread_cond_init(&g_crew_done, NULL);
}

static void *crew_thread(void *a) {
    CrewWorker *w = (CrewWorker *)a;
    pthread_mutex_lock(&g_crew_m);
    for (;;) {
        while (w->taken == w->posted) pthread_cond_wait(&g_crew_go, &g_crew_m);
        w->taken = w->posted;
        void (*f)(ShardJob *) = w->f;
        ShardJob *job = w->job;
        pthread_mutex_unlock(&g_crew_m);
        f(job);
        pthread_mutex_lock(&g_crew_m);
        if (--g_crew_pending == 0) pthread_cond_signal(&g_crew_done);
    }
    return NULL;
}

static void shards_parallel_own_threads(void (*f)(ShardJob *), ShardJob *jobs, Py_ssize_t n) {
    pthread_t th[64];
    ShardCall call[64];
    int started[64];
    for (Py_ssize_t k = 1; k < n; ++k) {
        call[k].f = f;
        call[k].job = &jobs[k];
        started[k] = pthread_create(&th[k], NULL, shard_thread, &call[k]) == 0;
    }
    if (n > 0) f(&jobs[0]);
    for (Py_ssize_t k = 1; k < n; ++k) {
        if (started[k]) pthread_join(th[k], NULL);
        else f(&jobs[k]);
    }
}

static void shards_parallel(void (*f)(ShardJob *), ShardJob *jobs, Py_ssize_t n) {
    if (n <= 1) {
        if (n == 1) f(&jobs[0]);
        return;
    }
    if (pthread_mutex_trylock(&g_crew_busy) != 0) {
        shards_parallel_own_threads(f, jobs, n);
        return;
    }
    int inline_from = (int)n;                  /* jobs from here on have no worker: run on this thread */
    pthread_mutex_lock(&g_crew_m);
    for (Py_ssize_t k = 1; k < n; ++k) {
        CrewWorker *w = &g_crew[k];
        if (!w->started) {
            pthread_attr_t at;
            pthread_attr_init(&at);
            pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
            w->started = pthread_create(&w->th, &at, crew_thread, w) == 0;
            pthread_attr_destroy(&at);
            if (!w->started) { inline_from = (int)k; break; }
        }
        w->f = f;
        w->job = &jobs[k];
        w->posted += 1;
        g_crew_pending += 1;
    }
    pthread_cond_broadcast(&g_crew_go);
    pthread_mutex_unlock(&g_crew_m);
    f(&jobs[0]);
    for (Py_ssize_t k = inline_from; k < n; ++k) f(&jobs[k]);
    pthread_mutex_lock(&g_crew_m);
    while (g_crew_pending > 0) pthread_cond_wait(&g_crew_done, &g_crew_m);
    pthread_mutex_unlock(&g_crew_m);
    pthread_mutex_unlock(&g_crew_busy);
}

static void shard_job_free(ShardJob *j) {
    if (j->snap) {
        for (Py_ssize_t k = 0; k < j->nsnap; ++k) {
            /* frames not yet released (error paths) */
            if (j->snap[k].data) for (Py_ssize_t i = 0; i < j->snap[k].n; ++i) Py_XDECREF(j->snap[k].data[i]);
            PyMem_Free(j->snap[k].slots);
            PyMem_Free(j->snap[k].data);
            if (j->ptrs) PyMem_Free((void *)j->ptrs[k]);
            if (j->status) PyMem_Free(j->status[k]);
        }
    }
    PyMem_Free(j->snap);
    PyMem_Free((void *)j->ptrs);
    PyMem_Free(j->status);
    PyMem_Free(j->wav_slot); PyMem_Free(j->wav_ns); PyMem_Free(j->wav_sr); PyMem_Free(j->wav_buf); PyMem_Free(j->wav_entry);
    if (j->rates.buf) PyBuffer_Release(&j->rates);
    Py_XDECREF((PyObject *)j->inbox);
}

static PyObject *wirebox_tick_shards(PyObject *Py_UNUSED(mod), PyObject *args) {
    PyObject *jobs_obj;
    if (!PyArg_ParseTuple(args, "O!", &PyList_Type, &jobs_obj)) return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(jobs_obj);
    if (n < 1 || n > 64) {
        PyErr_SetString(PyExc_ValueError, "tick_shards: 1..64 shards");
        return NULL;
    }
    ShardJob *jobs = (ShardJob *)PyMem_Calloc((size_t)n, sizeof(ShardJob));
    if (!jobs) return PyErr_NoMemory();
    PyObject *out = NULL;
    int ok = 1;
    /* ---- 1. arguments; every inbox hands over its boxes (the same detachment as Inbox.flush: all of them before the lock goes) */
    for (Py_ssize_t s = 0; s < n && ok; ++s) {
        ShardJob *j = &jobs[s];
        PyObject *inbox, *rates;
        unsigned long long fn, eng, rate_fn, run, res, work, wav;
        double thr;
        if (!PyArg_ParseTuple(PyList_GET_ITEM(jobs_obj, s), "OKKKKdKKKO", &inbox, &fn, &eng, &rate_fn, &run, &thr, &res, &work, &wav, &rates)) { ok = 0; break; }
        if (!eng || !run || !res || !work || (inbox != Py_None && (!PyObject_TypeCheck(inbox, &InboxType) || !fn))) {
            PyErr_SetString(PyExc_ValueError, "tick_shards: job = (inbox | None, push_fn, engine, rate_fn, run_fn, thresh, result, work, wav_fn, wav_rates)");
            ok = 0;
            break;
        }
        j->fn = (push_gather_fn)(uintptr_t)fn; j->rate_fn = (push_rate_gather_fn)(uintptr_t)rate_fn; j->eng = (void *)(uintptr_t)eng;
        j->run = (run_work_fn)(uintptr_t)run; j->thr = (float)thr;
        j->res = (vad_tick_result *)(uintptr_t)res; j->work = (vad_tick_work *)(uintptr_t)work;
        j->wav = (take_wav16_fn)(uintptr_t)wav;
        if (wav && rates != Py_None) {
            if (PyObject_GetBuffer(rates, &j->rates, PyBUF_CONTIG_RO) < 0) { j->rates.buf = NULL; ok = 0; break; }
            if (j->rates.itemsize != 4) { PyErr_SetString(PyExc_TypeError, "tick_shards: wav_rates must be an int32 buffer"); ok = 0; break; }
        }
        if (inbox == Py_None) continue;
        Inbox *ib = (Inbox *)inbox;
        for (Py_ssize_t k = 0; k < ib->nboxes; ++k)
            if (ib->boxes[k].n && ib->boxes[k].rate && !rate_fn) {
                PyErr_SetString(PyExc_ValueError, "tick_shards: chunks at another rate are waiting and no rate entry point was given");
                ok = 0;
            }
        if (!ok) break;
        Py_INCREF(inbox);
        j->inbox = ib;
        j->nsnap = ib->nboxes;
        j->snap = (Box *)PyMem_Calloc((size_t)(j->nsnap ? j->nsnap : 1), sizeof(Box));
        j->ptrs = (const void ***)PyMem_Calloc((size_t)(j->nsnap ? j->nsnap : 1), sizeof(void **));
        j->status = (int32_t **)PyMem_Calloc((size_t)(j->nsnap ? j->nsnap : 1), sizeof(int32_t *));
        if (!j->snap || !j->ptrs || !j->status) { PyErr_NoMemory(); ok = 0; break; }
        ib->epoch += 1;
        for (Py_ssize_t k = 0; k < j->nsnap; ++k) {
            Box *b = &ib->boxes[k];
            j->snap[k] = *b;
            b->slots = NULL; b->data = NULL; b->n = 0; b->cap = 0;
            ib->total -= j->snap[k].n;
        }
        for (Py_ssize_t k = 0; k < j->nsnap && ok; ++k) {
            const Py_ssize_t cnt = j->snap[k].n;
            if (!cnt) continue;
            j->ptrs[k] = (const void **)PyMem_Malloc((size_t)cnt * sizeof(void *));
            j->status[k] = (int32_t *)PyMem_Calloc((size_t)cnt, sizeof(int32_t));
            if (!j->ptrs[k] || !j->status[k]) { PyErr_NoMemory(); ok = 0; break; }
            for (Py_ssize_t i = 0; i < cnt; ++i) j->ptrs[k][i] = PyBytes_AS_STRING(j->snap[k].data[i]);
        }
    }
    if (!ok) goto done;
    /* ---- 2. the engines, side by side */
    struct timespec t0_, t1_, t2_, t3_, t4_;
    clock_gettime(CLOCK_MONOTONIC, &t0_);
    Py_BEGIN_ALLOW_THREADS
    shards_parallel(shard_push_and_run, jobs, n);
    Py_END_ALLOW_THREADS
    clock_gettime(CLOCK_MONOTONIC, &t1_);
    /* ---- 3. refusals; the frames are released; the payload buffers of the segments that ended */
    out = PyList_New(n);
    if (!out) { ok = 0; goto done; }
    for (Py_ssize_t s = 0; s < n && ok; ++s) {
        ShardJob *j = &jobs[s];
        PyObject *fails = PyList_New(0), *wavs = PyList_New(0);
        PyObject *row = (fails && wavs) ? Py_BuildValue("(iNN)", j->rc, fails, wavs) : NULL;
        if (!row) { Py_XDECREF(fails); Py_XDECREF(wavs); ok = 0; break; }
        PyList_SET_ITEM(out, s, row);
        for (Py_ssize_t k = 0; k < j->nsnap; ++k) {
            Box *sb = &j->snap[k];
            for (Py_ssize_t i = 0; i < sb->n && ok; ++i)
                if (j->status[k] && j->status[k][i] != 0) {
                    PyObject *t = Py_BuildValue("(Li)", (long long)sb->slots[i], (int)j->status[k][i]);
                    if (!t || PyList_Append(fails, t) < 0) ok = 0;
                    Py_XDECREF(t);
                }
            for (Py_ssize_t i = 0; i < sb->n; ++i) Py_DECREF(sb->data[i]);
            /* the (empty) arrays go back if nothing arrived meanwhile (index, never a kept pointer: the boxes may have moved) */
            Box *b = &j->inbox->boxes[k];
            if (b->slots == NULL && b->data == NULL) {
                b->slots = sb->slots; b->data = sb->data; b->cap = sb->cap;
            } else {
                PyMem_Free(sb->slots); PyMem_Free(sb->data);
            }
            sb->slots = NULL; sb->data = NULL; sb->n = 0;
        }
        if (!ok || j->rc != 0 || !j->wav || !j->rates.buf) continue;
        const vad_tick_work *w = j->work;
        Py_ssize_t ends = 0;
        for (int64_t q = 0; q < w->n_work; ++q) ends += (w->work_kind[q] & VAD_WORK_END) != 0;
        if (!ends) continue;
        j->wav_slot = (int64_t *)PyMem_Malloc((size_t)ends * sizeof(int64_t));
        j->wav_ns = (int64_t *)PyMem_Malloc((size_t)ends * sizeof(int64_t));
        j->wav_sr = (int32_t *)PyMem_Malloc((size_t)ends * sizeof(int32_t));
        j->wav_buf = (char **)PyMem_Malloc((size_t)ends * sizeof(char *));
        j->wav_entry = (Py_ssize_t *)PyMem_Malloc((size_t)ends * sizeof(Py_ssize_t));
        if (!j->wav_slot || !j->wav_ns || !j->wav_sr || !j->wav_buf || !j->wav_entry) { PyErr_NoMemory(); ok = 0; break; }
        const int32_t *rate_of = (const int32_t *)j->rates.buf;
        const Py_ssize_t nrates = j->rates.len / 4;
        for (int64_t q = 0; q < w->n_work && ok; ++q) {
            if (!(w->work_kind[q] & VAD_WORK_END)) continue;
            const int64_t slot = j->res->slots[w->work_index[q]];
            if (slot < 0 || slot >= nrates || rate_of[slot] <= 0 || w->work_samples[q] < 0) continue;
            PyObject *b = PyBytes_FromStringAndSize(NULL, 44 + 2 * (Py_ssize_t)w->work_samples[q]);
            PyObject *t = b ? Py_BuildValue("(nN)", (Py_ssize_t)q, b) : NULL;
            if (!t || PyList_Append(wavs, t) < 0) { Py_XDECREF(t); ok = 0; break; }
            Py_DECREF(t);
            const Py_ssize_t k = j->n_wav++;
            j->wav_slot[k] = slot; j->wav_ns[k] = w->work_samples[q]; j->wav_sr[k] = rate_of[slot];
            j->wav_buf[k] = PyBytes_AS_STRING(b); j->wav_entry[k] = (Py_ssize_t)q;
            if (getenv("VAD_WIREBOX_TOUCH")) memset(j->wav_buf[k], 0, (size_t)(44 + 2 * w->work_samples[q]));
        }
    }
    if (!ok) goto done;
    /* ---- 4. the payloads, side by side */
    clock_gettime(CLOCK_MONOTONIC, &t2_);
    Py_BEGIN_ALLOW_THREADS
    shards_parallel(shard_write_wavs, jobs, n);
    Py_END_ALLOW_THREADS
    clock_gettime(CLOCK_MONOTONIC, &t3_);
    if (getenv("VAD_WIREBOX_TIMING")) {
        clock_gettime(CLOCK_MONOTONIC, &t4_);
#define MS_(a, b) (((b).tv_sec - (a).tv_sec) * 1e3 + ((b).tv_nsec - (a).tv_nsec) * 1e-6)
        fprintf(stderr, "tick_shards: push+run %.2f ms | refusals + buffers %.2f ms | wavs %.2f ms\n", MS_(t0_, t1_), MS_(t1_, t2_), MS_(t2_, t3_));
#undef MS_
    }
    for (Py_ssize_t s = 0; s < n; ++s)
        if (jobs[s].wav_bad) {
            PyErr_Format(PyExc_RuntimeError, "vad_tick_take_segment_wav16 failed on shard %zd (status %d)", s, jobs[s].wav_bad);
            ok = 0;
            break;
        }
done:
    for (Py_ssize_t s = 0; s < n; ++s) shard_job_free(&jobs[s]);
    PyMem_Free(jobs);
    if (!ok) {
        Py_XDECREF(out);
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, "tick_shards failed");
        return NULL;
    }
    return out;
}

/* keep_heap(trim_threshold_bytes, top_pad_bytes) -> bool
 * A serving process with a pool per GPU builds tens of megabytes of voice_end payloads per tick and drops them a moment later.
 * glibc hands such memory back to the kernel as soon as it is freed (M_TRIM_THRESHOLD = 128 KiB) and the next tick's payload
 * buffers are fresh pages again: first touched by eight threads at once, the page faults queue inside the kernel - measured on a
 * 2 x 64-core host: 10.4 - 11.6 ms per round for 32 MB of payloads, 140 us per segment instead of 26.  With the two thresholds
 * raised the allocator keeps what a round frees and the next round writes warm memory.  Process-wide, so it is the serving
 * pool's constructor that asks for it (ShardedStreamPool), not the import of this module. */
#include <malloc.h>
static PyObject *wirebox_keep_heap(PyObject *Py_UNUSED(mod), PyObject *args) {
    long long trim, pad;
    if (!PyArg_ParseTuple(args, "LL", &trim, &pad)) return NULL;
    if (trim < 0 || pad < 0 || trim > (1LL << 31) - 1 || pad > (1LL << 31) - 1) {
        PyErr_SetString(PyExc_ValueError, "keep_heap(trim_threshold_bytes, top_pad_bytes): 0 .. 2^31 - 1");
        return NULL;
    }
#if defined(M_TRIM_THRESHOLD) && defined(M_TOP_PAD) && defined(M_MMAP_THRESHOLD)
    /* payloads of long utterances exceed the mmap threshold (128 KiB): those would be mapped and unmapped per segment */
    const int ok = mallopt(M_TRIM_THRESHOLD, (int)trim) && mallopt(M_TOP_PAD, (int)pad) && mallopt(M_MMAP_THRESHOLD, 64 << 20);
    return PyBool_FromLong(ok);
#else
    Py_RETURN_FALSE;
#endif
}

static PyMethodDef module_methods[] = {
    {"keep_heap", wirebox_keep_heap, METH_VARARGS, "keep_heap(trim_threshold_bytes, top_pad_bytes) -> bool (glibc mallopt; process-wide)"},
    {"tick_shards", wirebox_tick_shards, METH_VARARGS, "tick_shards(jobs) -> [(rc, [(slot, status), ...], [(work entry, bytes), ...]), ...]"},
    {"take_wav16_many", wirebox_take_wav16_many, METH_VARARGS, "take_wav16_many(fn_address, engine_address, slots, rates, nsamples) -> [bytes, ...]"},
    {"call_each", wirebox_call_each, METH_VARARGS, "call_each(callbacks, slots, arg) -> [(slot, exception), ...]"},
    {"take_wav16", wirebox_take_wav16, METH_VARARGS, "take_wav16(fn_address, engine_address, slot, sample_rate, nsamples) -> bytes"},
    {NULL, NULL, 0, NULL}};

static PyMethodDef inbox_methods[] = {
    {"pusher", (PyCFunction)inbox_pusher, METH_VARARGS, "pusher(slot, gate_on[, rate, nbytes[, fallback]]) -> callable push(data) -> bool (or fallback(data)'s result)"},
    {"flush", (PyCFunction)inbox_flush, METH_VARARGS, "flush(fn_address, engine_address[, rate_fn_address]) -> [(slot, status), ...] of refused frames"},
    {"drain", (PyCFunction)inbox_drain, METH_NOARGS, "drain() -> [(nbytes, gate, rate, [(slot, data), ...]), ...]; empties the inbox"},
    {NULL, NULL, 0, NULL}};
static PySequenceMethods inbox_as_sequence = {.sq_length = (lenfunc)inbox_len};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_wirebox", "inbox for int16 wire frames (see csrc/wirebox.c)", -1, module_methods, NULL, NULL, NULL, NULL};

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
    PusherType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL | Py_TPFLAGS_HAVE_GC;
    PusherType.tp_dealloc = (destructor)pusher_dealloc;
    PusherType.tp_traverse = (traverseproc)pusher_traverse;
    PusherType.tp_clear = (inquiry)pusher_clear;
    PusherType.tp_methods = pusher_methods;
    PusherType.tp_getset = pusher_getset;
    PusherType.tp_call = PyVectorcall_Call;
    PusherType.tp_vectorcall_offset = offsetof(Pusher, vectorcall);
    if (PyType_Ready(&InboxType) < 0 || PyType_Ready(&PusherType) < 0) return NULL;
    PyObject *m = PyModule_Create(&moduledef);
    if (!m) return NULL;
    pthread_atfork(NULL, NULL, crew_atfork_child);
    Py_INCREF(&InboxType);
    Py_INCREF(&PusherType);
    if (PyModule_AddObject(m, "Inbox", (PyObject *)&InboxType) < 0 || PyModule_AddObject(m, "Pusher", (PyObject *)&PusherType) < 0) {
        Py_DECREF(m);
        return NULL;
    }
    return m;
}
