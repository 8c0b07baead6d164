/* Native replay of inlier tables into the track store's Python containers.
 *
 * The track database keeps the reference's schema (models/TrackDatabase.py:11-31, 121-137):
 * tracks = {trackId: [(kp, (xl, xr, yl), frameId), ...]}, frame_ids = {frameId: set(trackId)},
 * last_insertions = {0: {kp: trackId}, 1: {...}} keyed by frame parity.  Feeding it row by row
 * (models/TrackDatabase.py:34-53, 171-184) is sequential by construction - track ids are handed
 * out in arrival order - and, for a whole sequence, millions of small Python objects.  This
 * module does exactly what TrackDatabase.update_tracks_from_tables does for a run of consecutive
 * frame pairs, with the containers built through the C API.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *loc_triple(const float *l)
{
    PyObject *t = PyTuple_New(3);
    if (!t)
        return NULL;
    for (int i = 0; i < 3; ++i) {
        PyObject *f = PyFloat_FromDouble((double)l[i]);
        if (!f) {
            Py_DECREF(t);
            return NULL;
        }
        PyTuple_SET_ITEM(t, i, f);
    }
    return t;
}

/* (kp, (a, b, c), frame) - steals nothing, returns a new reference */
static PyObject *track_point(PyObject *kp, const float *l, PyObject *frame)
{
    PyObject *xyz = loc_triple(l);
    if (!xyz)
        return NULL;
    PyObject *t = PyTuple_New(3);
    if (!t) {
        Py_DECREF(xyz);
        return NULL;
    }
    Py_INCREF(kp);
    Py_INCREF(frame);
    PyTuple_SET_ITEM(t, 0, kp);
    PyTuple_SET_ITEM(t, 1, xyz);
    PyTuple_SET_ITEM(t, 2, frame);
    return t;
}

/* replay_pairs(tracks, frame_ids, last_insertions, first_frame, rows, loc, n_inl,
 *              next_free, num_tracks, max_len, min_len, num_frames)
 *   -> (next_free, num_tracks, max_len, min_len, num_frames)
 * rows: int32 [T,4] (prev_left, prev_right, cur_left, cur_right), loc: float32 [T,6], n_inl: int64 [P]
 * rows per pair, in frame order; min_len < 0 stands for "no track yet"; num_frames follows the
 * frame table only on pairs that have rows (the reference updates it per appended point). */
static PyObject *replay_pairs(PyObject *Py_UNUSED(self), PyObject *args)
{
    PyObject *tracks, *frame_ids, *last_ins;
    long first_frame, next_free, num_tracks, max_len, min_len, num_frames;
    Py_buffer rows_b, loc_b, n_b;
    if (!PyArg_ParseTuple(args, "O!O!O!ly*y*y*lllll", &PyDict_Type, &tracks, &PyDict_Type, &frame_ids, &PyDict_Type,
                          &last_ins, &first_frame, &rows_b, &loc_b, &n_b, &next_free, &num_tracks, &max_len,
                          &min_len, &num_frames))
        return NULL;
    PyObject *result = NULL;
    const int32_t *rows = (const int32_t *)rows_b.buf;
    const float *loc = (const float *)loc_b.buf;
    const int64_t *n_inl = (const int64_t *)n_b.buf;
    const Py_ssize_t P = n_b.len / (Py_ssize_t)sizeof(int64_t);
    const Py_ssize_t T = rows_b.len / (Py_ssize_t)(4 * sizeof(int32_t));
    Py_ssize_t total = 0;
    for (Py_ssize_t p = 0; p < P; ++p) {
        if (n_inl[p] < 0) {
            PyErr_SetString(PyExc_ValueError, "replay_pairs: negative row count");
            goto done;
        }
        total += (Py_ssize_t)n_inl[p];
    }
    if (total > T || loc_b.len / (Py_ssize_t)(6 * sizeof(float)) < total) {
        PyErr_SetString(PyExc_ValueError, "replay_pairs: tables shorter than the row counts say");
        goto done;
    }

    Py_ssize_t off = 0;
    for (Py_ssize_t p = 0; p < P; ++p) {
        const long frame = first_frame + (long)p;
        PyObject *f_cur = PyLong_FromLong(frame), *f_nxt = PyLong_FromLong(frame + 1);
        PyObject *par_prev = PyLong_FromLong(frame % 2 < 0 ? frame % 2 + 2 : frame % 2);
        PyObject *par_cur = PyLong_FromLong((frame + 1) % 2 < 0 ? (frame + 1) % 2 + 2 : (frame + 1) % 2);
        PyObject *in_cur = PySet_New(NULL), *last_cur = PyDict_New();
        int ok = f_cur && f_nxt && par_prev && par_cur && in_cur && last_cur;
        PyObject *in_prev = NULL, *last_prev = NULL;
        if (ok) {
            /* prepare_to_next_pair */
            in_prev = PyDict_GetItemWithError(frame_ids, f_cur);            /* borrowed */
            if (!in_prev && !PyErr_Occurred()) {
                PyObject *s = PySet_New(NULL);
                ok = s && PyDict_SetItem(frame_ids, f_cur, s) == 0;
                in_prev = s;                                                 /* kept alive by the dict */
                Py_XDECREF(s);
            } else if (!in_prev) {
                ok = 0;
            }
            ok = ok && PyDict_SetItem(frame_ids, f_nxt, in_cur) == 0 && PyDict_SetItem(last_ins, par_cur, last_cur) == 0;
            if (ok) {
                last_prev = PyDict_GetItemWithError(last_ins, par_prev);     /* borrowed */
                if (!last_prev) {
                    if (!PyErr_Occurred())
                        PyErr_SetString(PyExc_KeyError, "replay_pairs: last_insertions lacks a parity slot");
                    ok = 0;
                }
            }
        }
        const Py_ssize_t m = (Py_ssize_t)n_inl[p];
        for (Py_ssize_t i = 0; ok && i < m; ++i) {
            const int32_t *r = rows + 4 * (off + i);
            const float *l = loc + 6 * (off + i);
            PyObject *kp_prev = PyLong_FromLong(r[0]), *kp_cur = PyLong_FromLong(r[2]);
            PyObject *tid = NULL, *point = NULL;
            ok = kp_prev && kp_cur;
            if (ok) {
                tid = PyDict_GetItemWithError(last_prev, kp_prev);           /* borrowed */
                if (tid) {
                    Py_INCREF(tid);
                } else if (PyErr_Occurred()) {
                    ok = 0;
                } else {
                    tid = PyLong_FromLong(next_free++);
                    ok = tid != NULL;
                }
            }
            PyObject *track = NULL;
            if (ok) {
                track = PyDict_GetItemWithError(tracks, tid);                /* borrowed */
                if (!track && PyErr_Occurred())
                    ok = 0;
            }
            if (ok && !track) {
                PyObject *lst = PyList_New(0);
                PyObject *first = lst ? track_point(kp_prev, l, f_cur) : NULL;
                ok = lst && first && PyList_Append(lst, first) == 0 && PyDict_SetItem(tracks, tid, lst) == 0 &&
                     PySet_Add(in_prev, tid) == 0;
                track = lst;                                                 /* kept alive by the dict */
                Py_XDECREF(first);
                Py_XDECREF(lst);
                if (ok)
                    ++num_tracks;
            }
            if (ok) {
                point = track_point(kp_cur, l + 3, f_nxt);
                ok = point && PyList_Append(track, point) == 0 && PySet_Add(in_cur, tid) == 0 &&
                     PyDict_SetItem(last_cur, kp_cur, tid) == 0;
            }
            if (ok) {
                const long n = (long)PyList_GET_SIZE(track);
                if (n > max_len)
                    max_len = n;
                if (min_len < 0 || n < min_len)
                    min_len = n;
            }
            Py_XDECREF(point);
            Py_XDECREF(tid);
            Py_XDECREF(kp_prev);
            Py_XDECREF(kp_cur);
        }
        Py_XDECREF(f_cur);
        Py_XDECREF(f_nxt);
        Py_XDECREF(par_prev);
        Py_XDECREF(par_cur);
        Py_XDECREF(in_cur);
        Py_XDECREF(last_cur);
        if (!ok)
            goto done;
        if (m > 0)
            num_frames = (long)PyDict_Size(frame_ids);
        off += m;
    }
    result = Py_BuildValue("lllll", next_free, num_tracks, max_len, min_len, num_frames);
done:
    PyBuffer_Release(&rows_b);
    PyBuffer_Release(&loc_b);
    PyBuffer_Release(&n_b);
    return result;
}

static PyMethodDef methods[] = {
    {"replay_pairs", replay_pairs, METH_VARARGS,
     "Feed the inlier tables of consecutive frame pairs into the track store's containers."},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_trackdb_native", NULL, -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__trackdb_native(void) { return PyModule_Create(&moduledef); }
