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
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

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
    PyObject **kp_obj = NULL, **slot_tid = NULL, **slot_list = NULL;
    Py_ssize_t *slot_pair = NULL;
    int32_t maxkp_done = -1;
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

    /* Keypoint indices are small: their int objects are made once and shared by every tuple and
     * dict key that needs them, and "which track did keypoint k of the previous pair's later frame
     * join" is answered from two arrays (one per frame parity) instead of the Python dicts.  Only
     * the dicts that outlive the call are filled: last_insertions keeps just the last two pairs. */
    int32_t maxkp = -1;
    for (Py_ssize_t i = 0; i < total; ++i) {
        const int32_t a = rows[4 * i], c = rows[4 * i + 2];
        if (a < 0 || c < 0) {
            PyErr_SetString(PyExc_ValueError, "replay_pairs: negative keypoint index");
            goto done;
        }
        if (a > maxkp)
            maxkp = a;
        if (c > maxkp)
            maxkp = c;
    }
    const size_t nkp = (size_t)(maxkp + 1);
    maxkp_done = maxkp;
    kp_obj = (PyObject **)calloc(nkp ? nkp : 1, sizeof(PyObject *));
    slot_tid = (PyObject **)calloc(2 * (nkp ? nkp : 1), sizeof(PyObject *));
    slot_list = (PyObject **)calloc(2 * (nkp ? nkp : 1), sizeof(PyObject *));
    slot_pair = (Py_ssize_t *)malloc(2 * (nkp ? nkp : 1) * sizeof(Py_ssize_t));
    if (!kp_obj || !slot_tid || !slot_list || !slot_pair) {
        PyErr_NoMemory();
        goto done;
    }
    for (size_t i = 0; i < 2 * nkp; ++i)
        slot_pair[i] = -1;

    Py_ssize_t off = 0;
    for (Py_ssize_t p = 0; p < P; ++p) {
        const long frame = first_frame + (long)p;
        const int keep_dict = p >= P - 2;      /* this pair's keypoint -> track dict survives the call */
        PyObject *f_cur = PyLong_FromLong(frame), *f_nxt = PyLong_FromLong(frame + 1);
        PyObject *par_prev = PyLong_FromLong(frame % 2 < 0 ? frame % 2 + 2 : frame % 2);
        PyObject *par_cur = PyLong_FromLong((frame + 1) % 2 < 0 ? (frame + 1) % 2 + 2 : (frame + 1) % 2);
        PyObject *in_cur = PySet_New(NULL), *last_cur = keep_dict ? PyDict_New() : NULL;
        int ok = f_cur && f_nxt && par_prev && par_cur && in_cur && (last_cur || !keep_dict);
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
            ok = ok && PyDict_SetItem(frame_ids, f_nxt, in_cur) == 0;
            if (ok && p == 0) {
                last_prev = PyDict_GetItemWithError(last_ins, par_prev);     /* borrowed */
                if (!last_prev) {
                    if (!PyErr_Occurred())
                        PyErr_SetString(PyExc_KeyError, "replay_pairs: last_insertions lacks a parity slot");
                    ok = 0;
                }
            }
            if (ok && keep_dict)
                ok = PyDict_SetItem(last_ins, par_cur, last_cur) == 0;
        }
        const size_t cur_slot = (size_t)(p & 1) * nkp, prev_slot = (size_t)((p & 1) ^ 1) * nkp;
        const Py_ssize_t m = (Py_ssize_t)n_inl[p];
        for (Py_ssize_t i = 0; ok && i < m; ++i) {
            const int32_t *r = rows + 4 * (off + i);
            const float *l = loc + 6 * (off + i);
            const int32_t a = r[0], c = r[2];
            if (!kp_obj[a])
                kp_obj[a] = PyLong_FromLong(a);
            if (!kp_obj[c])
                kp_obj[c] = PyLong_FromLong(c);
            PyObject *kp_prev = kp_obj[a], *kp_cur = kp_obj[c];              /* owned by kp_obj[] */
            PyObject *tid = NULL, *track = NULL;                            /* borrowed (the track dict owns both) */
            ok = kp_prev && kp_cur;
            if (ok && p > 0) {
                if (slot_pair[prev_slot + (size_t)a] == p - 1) {
                    tid = slot_tid[prev_slot + (size_t)a];
                    track = slot_list[prev_slot + (size_t)a];
                }
            } else if (ok) {
                tid = PyDict_GetItemWithError(last_prev, kp_prev);
                if (!tid && PyErr_Occurred())
                    ok = 0;
                if (ok && tid) {
                    track = PyDict_GetItemWithError(tracks, tid);
                    if (!track && PyErr_Occurred())
                        ok = 0;
                }
            }
            if (ok && !track) {
                PyObject *new_tid = NULL;
                if (!tid) {
                    new_tid = PyLong_FromLong(next_free++);
                    tid = new_tid;
                }
                PyObject *lst = PyList_New(0);
                PyObject *first = (lst && tid) ? track_point(kp_prev, l, f_cur) : NULL;
                ok = lst && first && PyList_Append(lst, first) == 0 && PyDict_SetItem(tracks, tid, lst) == 0 &&
                     PySet_Add(in_prev, tid) == 0;
                track = lst;                                                 /* kept alive by the dict */
                Py_XDECREF(first);
                Py_XDECREF(lst);
                Py_XDECREF(new_tid);                                         /* likewise, as its key */
                if (ok)
                    ++num_tracks;
            }
            if (ok) {
                PyObject *point = track_point(kp_cur, l + 3, f_nxt);
                ok = point && PyList_Append(track, point) == 0 && PySet_Add(in_cur, tid) == 0 &&
                     (!keep_dict || PyDict_SetItem(last_cur, kp_cur, tid) == 0);
                Py_XDECREF(point);
            }
            if (ok) {
                slot_pair[cur_slot + (size_t)c] = p;
                slot_tid[cur_slot + (size_t)c] = tid;
                slot_list[cur_slot + (size_t)c] = track;
                const long n = (long)PyList_GET_SIZE(track);
                if (n > max_len)
                    max_len = n;
                if (min_len < 0 || n < min_len)
                    min_len = n;
            }
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
    if (kp_obj)
        for (int32_t i = 0; i <= maxkp_done; ++i)
            Py_XDECREF(kp_obj[i]);
    free(kp_obj);
    free(slot_tid);
    free(slot_list);
    free(slot_pair);
    PyBuffer_Release(&rows_b);
    PyBuffer_Release(&loc_b);
    PyBuffer_Release(&n_b);
    return result;
}

/* build_tracks(pt_kp int32 [N], pt_frame int32 [N], pt_loc float32 [N,3], track_start int32 [n+1], n)
 *   -> {track id: [(kp, (xl, xr, yl), frame), ...]} for tracks 0 .. n-1
 * The dict-of-lists form of a track-point table that is laid out track after track (CSR). */
static PyObject *build_tracks(PyObject *Py_UNUSED(self), PyObject *args)
{
    Py_buffer kp_b, fr_b, loc_b, st_b;
    long n_tracks;
    if (!PyArg_ParseTuple(args, "y*y*y*y*l", &kp_b, &fr_b, &loc_b, &st_b, &n_tracks))
        return NULL;
    PyObject *tracks = NULL, *result = NULL;
    PyObject **small = NULL;           /* shared int objects for keypoint indices and frame ids */
    long n_small = 0;
    const int32_t *kp = (const int32_t *)kp_b.buf, *fr = (const int32_t *)fr_b.buf, *st = (const int32_t *)st_b.buf;
    const float *loc = (const float *)loc_b.buf;
    const Py_ssize_t N = kp_b.len / (Py_ssize_t)sizeof(int32_t);
    if (n_tracks < 0 || st_b.len / (Py_ssize_t)sizeof(int32_t) < n_tracks + 1 ||
        fr_b.len / (Py_ssize_t)sizeof(int32_t) < N || loc_b.len / (Py_ssize_t)(3 * sizeof(float)) < N) {
        PyErr_SetString(PyExc_ValueError, "build_tracks: tables shorter than the track table says");
        goto done;
    }
    if (n_tracks > 0 && (st[0] != 0 || st[n_tracks] > N)) {
        PyErr_SetString(PyExc_ValueError, "build_tracks: track_start does not fit the point table");
        goto done;
    }
    for (Py_ssize_t i = 0; i < (n_tracks ? st[n_tracks] : 0); ++i) {
        if (kp[i] < 0 || fr[i] < 0) {
            PyErr_SetString(PyExc_ValueError, "build_tracks: negative keypoint index or frame id");
            goto done;
        }
        if (kp[i] >= n_small)
            n_small = (long)kp[i] + 1;
        if (fr[i] >= n_small)
            n_small = (long)fr[i] + 1;
    }
    small = (PyObject **)calloc(n_small ? (size_t)n_small : 1, sizeof(PyObject *));
    tracks = PyDict_New();
    if (!small || !tracks) {
        PyErr_NoMemory();
        goto done;
    }
    for (long t = 0; t < n_tracks; ++t) {
        const Py_ssize_t a = st[t], b = st[t + 1];
        if (b < a) {
            PyErr_SetString(PyExc_ValueError, "build_tracks: track_start must not decrease");
            goto done;
        }
        PyObject *lst = PyList_New(b - a);
        PyObject *tid = PyLong_FromLong(t);
        int ok = lst && tid;
        for (Py_ssize_t i = a; ok && i < b; ++i) {
            if (!small[kp[i]])
                small[kp[i]] = PyLong_FromLong(kp[i]);
            if (!small[fr[i]])
                small[fr[i]] = PyLong_FromLong(fr[i]);
            PyObject *pt = (small[kp[i]] && small[fr[i]]) ? track_point(small[kp[i]], loc + 3 * i, small[fr[i]]) : NULL;
            if (!pt)
                ok = 0;
            else
                PyList_SET_ITEM(lst, i - a, pt);
        }
        ok = ok && PyDict_SetItem(tracks, tid, lst) == 0;
        Py_XDECREF(lst);
        Py_XDECREF(tid);
        if (!ok)
            goto done;
    }
    result = tracks;
    tracks = NULL;
done:
    Py_XDECREF(tracks);
    if (small)
        for (long i = 0; i < n_small; ++i)
            Py_XDECREF(small[i]);
    free(small);
    PyBuffer_Release(&kp_b);
    PyBuffer_Release(&fr_b);
    PyBuffer_Release(&loc_b);
    PyBuffer_Release(&st_b);
    return result;
}

/* track_list(pt_kp, pt_frame, pt_loc, start, stop) -> [(kp, (xl, xr, yl), frame), ...] of point slots
 * [start, stop): one track of the table, as the list the track store keeps for it. */
static PyObject *track_list(PyObject *Py_UNUSED(self), PyObject *args)
{
    Py_buffer kp_b, fr_b, loc_b;
    Py_ssize_t a, b;
    if (!PyArg_ParseTuple(args, "y*y*y*nn", &kp_b, &fr_b, &loc_b, &a, &b))
        return NULL;
    PyObject *lst = NULL;
    const int32_t *kp = (const int32_t *)kp_b.buf, *fr = (const int32_t *)fr_b.buf;
    const float *loc = (const float *)loc_b.buf;
    const Py_ssize_t N = kp_b.len / (Py_ssize_t)sizeof(int32_t);
    if (a < 0 || b < a || b > N || fr_b.len / (Py_ssize_t)sizeof(int32_t) < N ||
        loc_b.len / (Py_ssize_t)(3 * sizeof(float)) < N) {
        PyErr_SetString(PyExc_ValueError, "track_list: slot range outside the point table");
        goto done;
    }
    lst = PyList_New(b - a);
    for (Py_ssize_t i = a; lst && i < b; ++i) {
        PyObject *k = PyLong_FromLong(kp[i]), *f = PyLong_FromLong(fr[i]);
        PyObject *pt = (k && f) ? track_point(k, loc + 3 * i, f) : NULL;
        Py_XDECREF(k);
        Py_XDECREF(f);
        if (!pt) {
            Py_CLEAR(lst);
            break;
        }
        PyList_SET_ITEM(lst, i - a, pt);
    }
done:
    PyBuffer_Release(&kp_b);
    PyBuffer_Release(&fr_b);
    PyBuffer_Release(&loc_b);
    return lst;
}

/* chain_poses(rel float64 [P,12], first float64 [12], ex float64 [P+1,12] out, cam float64 [P+1,3] out)
 * Global extrinsics and camera centres from relative poses (utils/utils.py:283-288):
 * R' = R R_prev, t' = R t_prev + t, centre = -R'^T t'.
 * The reference takes these products with numpy's `@`, i.e. with whatever its BLAS does; the sums
 * below are taken in the order numpy 2.2 / OpenBLAS 0.3.29 (x86-64 kernels with FMA) takes them for
 * these shapes -- matrix product: fma(a2,b2, fma(a1,b1, a0*b0)); matrix (row stride 4) times vector:
 * fma(a2,b2, fma(a0,b0, a1*b1)); negated transpose times vector: the first form again.  The caller
 * checks the result against numpy's own on a sample before trusting it (sharding.chain_poses). */
static PyObject *chain_poses(PyObject *Py_UNUSED(self), PyObject *args)
{
    Py_buffer rel_b, first_b, ex_b, cam_b;
    if (!PyArg_ParseTuple(args, "y*y*w*w*", &rel_b, &first_b, &ex_b, &cam_b))
        return NULL;
    PyObject *result = NULL;
    const Py_ssize_t P = rel_b.len / (Py_ssize_t)(12 * sizeof(double));
    if (first_b.len < (Py_ssize_t)(12 * sizeof(double)) || ex_b.len < (P + 1) * (Py_ssize_t)(12 * sizeof(double)) ||
        cam_b.len < (P + 1) * (Py_ssize_t)(3 * sizeof(double))) {
        PyErr_SetString(PyExc_ValueError, "chain_poses: output buffers too small");
        goto done;
    }
    {
        const double *rel = (const double *)rel_b.buf, *first = (const double *)first_b.buf;
        double *ex = (double *)ex_b.buf, *cam = (double *)cam_b.buf;
        for (int i = 0; i < 12; ++i)
            ex[i] = first[i];
        cam[0] = cam[1] = cam[2] = 0.0;
        for (Py_ssize_t p = 0; p < P; ++p) {
            const double *r = rel + 12 * p, *e = ex + 12 * p;
            double *o = ex + 12 * (p + 1), *c = cam + 3 * (p + 1);
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j)
                    o[4 * i + j] = fma(r[4 * i + 2], e[8 + j], fma(r[4 * i + 1], e[4 + j], r[4 * i] * e[j]));
                o[4 * i + 3] = fma(r[4 * i + 2], e[11], fma(r[4 * i], e[3], r[4 * i + 1] * e[7])) + r[4 * i + 3];
            }
            for (int j = 0; j < 3; ++j)
                c[j] = fma(-o[8 + j], o[11], fma(-o[4 + j], o[7], (-o[j]) * o[3]));
        }
    }
    result = Py_None;
    Py_INCREF(result);
done:
    PyBuffer_Release(&rel_b);
    PyBuffer_Release(&first_b);
    PyBuffer_Release(&ex_b);
    PyBuffer_Release(&cam_b);
    return result;
}

static PyMethodDef methods[] = {
    {"build_tracks", build_tracks, METH_VARARGS, "Track-point table (CSR by track) -> {track id: [(kp, (xl, xr, yl), frame), ...]}."},
    {"track_list", track_list, METH_VARARGS, "One track of a track-point table as a list of (kp, (xl, xr, yl), frame)."},
    {"chain_poses", chain_poses, METH_VARARGS, "Relative poses -> global extrinsics and camera centres."},
    {"replay_pairs", replay_pairs, METH_VARARGS,
     "Feed the inlier tables of consecutive frame pairs into the track store's containers."},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_trackdb_native", NULL, -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__trackdb_native(void) { return PyModule_Create(&moduledef); }
