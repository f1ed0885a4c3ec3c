/*
 * genome_walk.c -- CPython extension `_genome_walk`: reads a list of NEAT genome OBJECTS into plain arrays.
 *
 * Host-side plumbing of the drop-in boundary (not a compute path): generate_illusion.py hands get_fitnesses_neat a list
 * of neat-python genome objects (/root/reference/generate_illusion.py:502-514: `for genome_id, genome in population`,
 * create_cppn(genome, config, ...) at :384-389 walks genome.connections / genome.nodes).  The engine wants the arrays of
 * eigen_flatten_genomes (include/eigen_engine.h).  Doing that walk in Python cost 10 ms per 256 genomes (28 K attribute
 * reads + list -> array conversions), 40 % of a generation at the small configurations; here it is one C call that
 * writes straight into caller-allocated numpy buffers.
 *
 * Specification = genome.py: GenomeBatch._marshal_python (same arrays, element for element; tests/test_host_logic.py).
 *   walk(genomes, act_ids, conn_off, conn_in, conn_out, conn_w, conn_en, node_off, node_key, node_act, node_agg, node_bias, node_resp)
 *     genomes  : sequence of objects with .connections {(i, o): gene(.weight, .enabled)} and
 *                .nodes {key: gene(.bias, .response, .activation, .aggregation)}
 *     act_ids  : dict activation-name -> id (unknown names are written as 255)
 *     the rest : writable buffers (int32 / float64 / uint8) sized by the caller from len(g.connections), len(g.nodes)
 * The (in, out) pair of a connection is its gene's `.key` attribute, as _marshal_python and PyTorch-NEAT's create_cppn
 * (`i, o = cg.key`) read it -- not the dict key it is stored under.  Keys must fit int32 (OverflowError otherwise).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *s_connections, *s_nodes, *s_weight, *s_enabled, *s_bias, *s_response, *s_activation, *s_aggregation, *s_key;

/* instance attribute: straight from the instance __dict__ when there is one AND the type defines nothing under that name
 * (plain objects, SimpleNamespace, neat-python genes); the generic protocol otherwise, so that properties, data descriptors
 * and slots on the class win exactly as they do for `getattr` in _marshal_python.  Returns a NEW reference or NULL with an
 * exception set. */
static PyObject* attr(PyObject* o, PyObject* name)
{
    PyObject** dp = _PyType_Lookup(Py_TYPE(o), name) == NULL ? _PyObject_GetDictPtr(o) : NULL;
    if (dp && *dp) {
        PyObject* v = PyDict_GetItemWithError(*dp, name);
        if (v) { Py_INCREF(v); return v; }
        if (PyErr_Occurred()) return NULL;
    }
    return PyObject_GetAttr(o, name);
}

/* Python int -> int32 with a range check (a silent cast would alias distinct node keys) */
static int as_i32(PyObject* o, int32_t* out)
{
    const long v = PyLong_AsLong(o);
    if (v == -1 && PyErr_Occurred()) return -1;
    if (v < INT32_MIN || v > INT32_MAX) { PyErr_Format(PyExc_OverflowError, "gene key %ld does not fit int32", v); return -1; }
    *out = (int32_t)v;
    return 0;
}

static int as_double(PyObject* o, PyObject* name, double* out)
{
    PyObject* v = attr(o, name);
    if (!v) return -1;
    const double d = PyFloat_AsDouble(v);
    Py_DECREF(v);
    if (d == -1.0 && PyErr_Occurred()) return -1;
    *out = d;
    return 0;
}

typedef struct { Py_buffer b; int ok; } Buf;

static int get_buf(PyObject* o, Buf* b, Py_ssize_t itemsize, const char* what)
{
    b->ok = 0;
    if (PyObject_GetBuffer(o, &b->b, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return -1;
    b->ok = 1;
    if (b->b.itemsize != itemsize) { PyErr_Format(PyExc_TypeError, "%s: item size %zd, expected %zd", what, b->b.itemsize, itemsize); return -1; }
    return 0;
}

static PyObject* walk(PyObject* self, PyObject* args)
{
    PyObject *genomes, *act_ids, *o[11];
    if (!PyArg_ParseTuple(args, "OO!OOOOOOOOOOO", &genomes, &PyDict_Type, &act_ids, &o[0], &o[1], &o[2], &o[3], &o[4], &o[5], &o[6], &o[7],
                          &o[8], &o[9], &o[10]))
        return NULL;
    static const Py_ssize_t isz[11] = {4, 4, 4, 8, 1, 4, 4, 1, 1, 8, 8};
    static const char* nm[11] = {"conn_off", "conn_in", "conn_out", "conn_w", "conn_en", "node_off", "node_key", "node_act", "node_agg", "node_bias", "node_resp"};
    Buf b[11];
    PyObject* seq = NULL;
    PyObject* ret = NULL;
    int nb = 0;
    for (; nb < 11; ++nb)
        if (get_buf(o[nb], &b[nb], isz[nb], nm[nb]) != 0) { if (b[nb].ok) ++nb; goto done; }
    seq = PySequence_Fast(genomes, "genomes must be a sequence");
    if (!seq) goto done;
    {
        const Py_ssize_t G = PySequence_Fast_GET_SIZE(seq);
        int32_t* conn_off = (int32_t*)b[0].b.buf; int32_t* cin = (int32_t*)b[1].b.buf; int32_t* cout = (int32_t*)b[2].b.buf;
        double* cw = (double*)b[3].b.buf; uint8_t* cen = (uint8_t*)b[4].b.buf;
        int32_t* node_off = (int32_t*)b[5].b.buf; int32_t* nkey = (int32_t*)b[6].b.buf; uint8_t* nact = (uint8_t*)b[7].b.buf;
        uint8_t* nagg = (uint8_t*)b[8].b.buf; double* nbias = (double*)b[9].b.buf; double* nresp = (double*)b[10].b.buf;
        const Py_ssize_t cap_c = b[1].b.len / 4, cap_n = b[6].b.len / 4;
        if (b[0].b.len / 4 < G + 1 || b[5].b.len / 4 < G + 1 || b[2].b.len / 4 < cap_c || b[3].b.len / 8 < cap_c || b[4].b.len < cap_c ||
            b[7].b.len < cap_n || b[8].b.len < cap_n || b[9].b.len / 8 < cap_n || b[10].b.len / 8 < cap_n) {
            PyErr_SetString(PyExc_ValueError, "output buffers are inconsistent in size");
            goto done;
        }
        Py_ssize_t nc = 0, nn = 0;
        conn_off[0] = 0; node_off[0] = 0;
        for (Py_ssize_t g = 0; g < G; ++g) {
            PyObject* genome = PySequence_Fast_GET_ITEM(seq, g);
            PyObject* conns = attr(genome, s_connections);
            if (!conns) goto done;
            if (!PyDict_Check(conns)) { Py_DECREF(conns); PyErr_SetString(PyExc_TypeError, "genome.connections must be a dict"); goto done; }
            Py_ssize_t pos = 0; PyObject *k, *v;
            while (PyDict_Next(conns, &pos, &k, &v)) {
                if (nc >= cap_c) { Py_DECREF(conns); PyErr_SetString(PyExc_ValueError, "connection buffers too small"); goto done; }
                PyObject* ck = attr(v, s_key);  /* the gene's own key: `i, o = cg.key` */
                if (!ck) { Py_DECREF(conns); goto done; }
                if (!PyTuple_Check(ck) || PyTuple_GET_SIZE(ck) != 2) { Py_DECREF(ck); Py_DECREF(conns); PyErr_SetString(PyExc_TypeError, "connection gene .key must be an (in, out) tuple"); goto done; }
                const int kerr = as_i32(PyTuple_GET_ITEM(ck, 0), &cin[nc]) || as_i32(PyTuple_GET_ITEM(ck, 1), &cout[nc]);
                Py_DECREF(ck);
                if (kerr) { Py_DECREF(conns); goto done; }
                if (as_double(v, s_weight, &cw[nc]) != 0) { Py_DECREF(conns); goto done; }
                PyObject* en = attr(v, s_enabled);
                if (!en) { Py_DECREF(conns); goto done; }
                const int t = PyObject_IsTrue(en);
                Py_DECREF(en);
                if (t < 0) { Py_DECREF(conns); goto done; }
                cen[nc] = (uint8_t)t;
                ++nc;
            }
            Py_DECREF(conns);
            conn_off[g + 1] = (int32_t)nc;
            PyObject* nodes = attr(genome, s_nodes);
            if (!nodes) goto done;
            if (!PyDict_Check(nodes)) { Py_DECREF(nodes); PyErr_SetString(PyExc_TypeError, "genome.nodes must be a dict"); goto done; }
            pos = 0;
            while (PyDict_Next(nodes, &pos, &k, &v)) {
                if (nn >= cap_n) { Py_DECREF(nodes); PyErr_SetString(PyExc_ValueError, "node buffers too small"); goto done; }
                if (as_i32(k, &nkey[nn]) != 0) { Py_DECREF(nodes); goto done; }
                PyObject* a = attr(v, s_activation);
                if (!a) { Py_DECREF(nodes); goto done; }
                PyObject* id = PyDict_GetItemWithError(act_ids, a);  /* borrowed */
                Py_DECREF(a);
                if (!id && PyErr_Occurred()) { Py_DECREF(nodes); goto done; }
                nact[nn] = id ? (uint8_t)PyLong_AsLong(id) : 255;
                PyObject* ag = attr(v, s_aggregation);
                if (!ag) { Py_DECREF(nodes); goto done; }
                nagg[nn] = (PyUnicode_Check(ag) && PyUnicode_CompareWithASCIIString(ag, "sum") == 0) ? 1 : 0;
                Py_DECREF(ag);
                if (as_double(v, s_bias, &nbias[nn]) != 0 || as_double(v, s_response, &nresp[nn]) != 0) { Py_DECREF(nodes); goto done; }
                ++nn;
            }
            Py_DECREF(nodes);
            node_off[g + 1] = (int32_t)nn;
        }
        ret = Py_BuildValue("(nn)", nc, nn);
    }
done:
    Py_XDECREF(seq);
    for (int i = 0; i < nb; ++i)
        if (b[i].ok) PyBuffer_Release(&b[i].b);
    return ret;
}

static PyMethodDef methods[] = {{"walk", walk, METH_VARARGS, "read genome objects into caller-allocated arrays; returns (n_connections, n_nodes)"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_genome_walk", "genome object walker (see genome_walk.c)", -1, methods};

PyMODINIT_FUNC PyInit__genome_walk(void)
{
    s_connections = PyUnicode_InternFromString("connections"); s_nodes = PyUnicode_InternFromString("nodes");
    s_weight = PyUnicode_InternFromString("weight"); s_enabled = PyUnicode_InternFromString("enabled");
    s_bias = PyUnicode_InternFromString("bias"); s_response = PyUnicode_InternFromString("response");
    s_activation = PyUnicode_InternFromString("activation"); s_aggregation = PyUnicode_InternFromString("aggregation");
    s_key = PyUnicode_InternFromString("key");
    return PyModule_Create(&moddef);
}
