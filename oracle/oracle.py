"""ctypes/numpy front-end of the CPU oracle (oracle/kpop_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under kpop_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkpop_oracle.so")

DNA_DS, DNA_SS, PROTEIN = 0, 1, 2
EUCLIDEAN, COSINE, MINKOWSKI = 0, 1, 2


def build(force=False):
    src = os.path.join(_HERE, "kpop_oracle.c")
    hdr = os.path.join(_HERE, "kpop_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libkpop_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


u8p, u32p, u64p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_double))


def _declare(L):
    L.kpo_mix64.restype = C.c_uint64
    L.kpo_mix64.argtypes = [C.c_uint64]
    L.kpo_splitmix_at.restype = C.c_uint64
    L.kpo_splitmix_at.argtypes = [C.c_uint64, C.c_uint64]
    L.kpo_synth_reads.restype = None
    L.kpo_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, u8p]
    L.kpo_synth_twister_coeff.restype = C.c_double
    L.kpo_synth_twister_coeff.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    L.kpo_synth_inertia.restype = None
    L.kpo_synth_inertia.argtypes = [C.c_uint32, f64p]
    L.kpo_enumerate_kmers.restype = C.c_uint64
    L.kpo_enumerate_kmers.argtypes = [C.c_int, C.c_int, u64p]
    L.kpo_synth_twister.restype = None
    L.kpo_synth_twister.argtypes = [C.c_uint64, C.c_uint32, u64p, C.c_uint64, f64p]
    L.kpo_base_code.restype = C.c_int
    L.kpo_base_code.argtypes = [C.c_uint8]
    L.kpo_to_hex.restype = None
    L.kpo_to_hex.argtypes = [C.c_uint64, C.c_int, C.c_char_p]
    L.kpo_to_hex_protein.restype = None
    L.kpo_to_hex_protein.argtypes = [C.c_uint64, C.c_int, C.c_char_p]
    L.kpo_count_read.restype = C.c_int64
    L.kpo_count_read.argtypes = [u8p, C.c_uint64, C.c_int, C.c_int, u64p, u32p, C.c_uint64]
    L.kpo_count_reads.restype = C.c_int
    L.kpo_count_reads.argtypes = [u8p, u64p, C.c_uint32, C.c_int, C.c_int, C.c_int, u64p, u32p, u64p,
                                  C.c_uint64]
    L.kpo_twist.restype = C.c_int
    L.kpo_twist.argtypes = [f64p, C.c_uint64, C.c_uint32, u64p, u64p, f64p, u64p, C.c_uint32, C.c_int,
                            f64p]
    L.kpo_metric_flat.restype = None
    L.kpo_metric_flat.argtypes = [C.c_uint32, f64p]
    L.kpo_metric_powers.restype = None
    L.kpo_metric_powers.argtypes = [f64p, C.c_uint32, C.c_double, C.c_double, C.c_double, f64p]
    L.kpo_norm.restype = C.c_double
    L.kpo_norm.argtypes = [C.c_int, C.c_double, f64p, f64p, C.c_uint32]
    L.kpo_normalizations.restype = None
    L.kpo_normalizations.argtypes = [C.c_int, C.c_double, f64p, f64p, C.c_uint32, C.c_uint32, f64p]
    L.kpo_distance_rowwise.restype = None
    L.kpo_distance_rowwise.argtypes = [f64p, C.c_uint32, f64p, C.c_uint32, C.c_uint32, f64p, C.c_int,
                                       C.c_double, C.c_int, f64p]
    L.kpo_summarize_row.restype = C.c_uint32
    L.kpo_summarize_row.argtypes = [f64p, C.c_uint32, C.c_uint32, f64p, u32p, f64p, f64p]
    L.kpo_distance_summary.restype = C.c_int
    L.kpo_distance_summary.argtypes = [f64p, C.c_uint32, f64p, C.c_uint32, C.c_uint32, f64p, C.c_int,
                                       C.c_double, C.c_int, C.c_uint32, f64p, u64p, u32p, f64p, f64p,
                                       C.c_uint64]
    L.kpo_counter_stats.restype = None
    L.kpo_counter_stats.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint64, C.c_double, C.c_double, f64p, f64p]
    L.kpo_counter_transform_one.restype = C.c_double
    L.kpo_counter_transform_one.argtypes = [C.c_int, C.c_double, C.c_double, f64p, C.c_int32]
    L.kpo_counter_combine.restype = C.c_double
    L.kpo_counter_combine.argtypes = [C.POINTER(C.c_void_p), C.c_uint64, u32p, C.c_uint32, f64p, C.c_int, C.c_void_p]
    L.kpo_embeddings.restype = None
    L.kpo_embeddings.argtypes = [f64p, C.c_uint32, C.c_uint32, f64p, C.c_int, C.c_double, C.c_int, f64p]
    L.kpo_pipeline.restype = C.c_double
    L.kpo_pipeline.argtypes = [u8p, u64p, C.c_uint32, C.c_int, C.c_int, f64p, C.c_uint64, C.c_uint32,
                               u64p, f64p, C.c_uint32, f64p, C.c_int, C.c_double, C.c_int, C.c_int,
                               C.c_int, f64p, f64p]


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ------------------------------------------------------------------ synth
def synth_reads(seed, n_reads, read_len):
    out = np.empty(n_reads * read_len, dtype=np.uint8)
    lib().kpo_synth_reads(seed, n_reads, read_len, _p(out, C.c_uint8))
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return out, offsets


def enumerate_kmers(k, content=DNA_DS):
    n = lib().kpo_enumerate_kmers(k, content, None)
    out = np.empty(n, dtype=np.uint64)
    lib().kpo_enumerate_kmers(k, content, _p(out, C.c_uint64))
    return out


def synth_twister(seed, n_dims, col_hash):
    col_hash = _c(col_hash, np.uint64)
    T = np.empty((n_dims, len(col_hash)), dtype=np.float64)
    lib().kpo_synth_twister(seed, n_dims, _p(col_hash, C.c_uint64), len(col_hash), _p(T, C.c_double))
    return T


def synth_inertia(n_dims):
    w = np.empty(n_dims, dtype=np.float64)
    lib().kpo_synth_inertia(n_dims, _p(w, C.c_double))
    return w


def to_hex(h, k, content=DNA_DS):
    buf = C.create_string_buffer(20)
    (lib().kpo_to_hex_protein if content == PROTEIN else lib().kpo_to_hex)(int(h), k, buf)
    return buf.value.decode()


# ------------------------------------------------------------------ count
def count_reads(bases, offsets, k, content=DNA_DS, per_read=True):
    bases = _c(bases, np.uint8)
    offsets = _c(offsets, np.uint64)
    n = len(offsets) - 1
    cap = int(offsets[-1] - offsets[0]) + 1
    oh = np.empty(cap, dtype=np.uint64)
    oc = np.empty(cap, dtype=np.uint32)
    oo = np.zeros(n + 1 if per_read else 2, dtype=np.uint64)
    bb = bases if len(bases) else np.zeros(1, np.uint8)
    rc = lib().kpo_count_reads(_p(bb, C.c_uint8), _p(offsets, C.c_uint64), n, k, content,
                               1 if per_read else 0, _p(oh, C.c_uint64), _p(oc, C.c_uint32),
                               _p(oo, C.c_uint64), cap)
    if rc != 0:
        raise RuntimeError("kpo_count_reads failed: %d" % rc)
    tot = int(oo[-1])
    return oh[:tot].copy(), oc[:tot].copy(), oo


# ------------------------------------------------------------------ twist
def twist(T_dims_major, col_hash, hash_, value, offsets, normalize=True):
    T = _c(T_dims_major, np.float64)
    n_dims, n_cols = T.shape
    col_hash = _c(col_hash, np.uint64)
    hash_ = _c(hash_, np.uint64)
    value = _c(value, np.float64)
    offsets = _c(offsets, np.uint64)
    n = len(offsets) - 1
    out = np.zeros((n, n_dims), dtype=np.float64)
    hh = hash_ if len(hash_) else np.zeros(1, np.uint64)
    vv = value if len(value) else np.zeros(1, np.float64)
    lib().kpo_twist(_p(T, C.c_double), n_cols, n_dims, _p(col_hash, C.c_uint64), _p(hh, C.c_uint64),
                    _p(vv, C.c_double), _p(offsets, C.c_uint64), n, 1 if normalize else 0,
                    _p(out, C.c_double))
    return out


# ------------------------------------------------------------------ metric
def metric_flat(n):
    out = np.empty(n, dtype=np.float64)
    lib().kpo_metric_flat(n, _p(out, C.c_double))
    return out


def metric_powers(inertia, power_int=1., threshold=1., power_ext=2.):
    inertia = _c(inertia, np.float64)
    out = np.empty(len(inertia), dtype=np.float64)
    lib().kpo_metric_powers(_p(inertia, C.c_double), len(inertia), power_int, threshold, power_ext,
                            _p(out, C.c_double))
    return out


# ------------------------------------------------------------------ distance
def normalizations(m, metric, kind=EUCLIDEAN, p=2.):
    m = _c(m, np.float64)
    metric = _c(metric, np.float64)
    out = np.empty(m.shape[0], dtype=np.float64)
    lib().kpo_normalizations(kind, p, _p(metric, C.c_double), _p(m, C.c_double), m.shape[0], m.shape[1],
                             _p(out, C.c_double))
    return out


def distance_rowwise(m1, m2, metric, kind=EUCLIDEAN, p=2., normalize=True):
    m1 = _c(m1, np.float64)
    m2 = _c(m2, np.float64)
    metric = _c(metric, np.float64)
    r1, d = m1.shape
    r2 = m2.shape[0]
    out = np.empty((r2, r1), dtype=np.float64)
    lib().kpo_distance_rowwise(_p(m1, C.c_double), r1, _p(m2, C.c_double), r2, d, _p(metric, C.c_double),
                               kind, p, 1 if normalize else 0, _p(out, C.c_double))
    return out


def embeddings(m, metric, kind=EUCLIDEAN, p=2., normalize=True):
    m = _c(m, np.float64)
    metric = _c(metric, np.float64)
    out = np.zeros_like(m)
    lib().kpo_embeddings(_p(m, C.c_double), m.shape[0], m.shape[1], _p(metric, C.c_double), kind, p, 1 if normalize else 0,
                         _p(out, C.c_double))
    return out


def summarize_row(row, req_len):
    row = _c(row, np.float64)
    n = len(row)
    stats = np.empty(4, dtype=np.float64)
    idx = np.empty(max(n, 1), dtype=np.uint32)
    dist = np.empty(max(n, 1), dtype=np.float64)
    z = np.empty(max(n, 1), dtype=np.float64)
    e = lib().kpo_summarize_row(_p(row, C.c_double), n, req_len, _p(stats, C.c_double),
                                _p(idx, C.c_uint32), _p(dist, C.c_double), _p(z, C.c_double))
    return stats, idx[:e].copy(), dist[:e].copy(), z[:e].copy()


def distance_summary(m1, m2, metric, kind=EUCLIDEAN, p=2., normalize=True, keep_at_most=2):
    m1 = _c(m1, np.float64)
    m2 = _c(m2, np.float64)
    metric = _c(metric, np.float64)
    r1, d = m1.shape
    r2 = m2.shape[0]
    cap = max(r1 * r2, 1)
    stats = np.empty((r2, 4), dtype=np.float64)
    offs = np.zeros(r2 + 1, dtype=np.uint64)
    idx = np.empty(cap, dtype=np.uint32)
    dist = np.empty(cap, dtype=np.float64)
    z = np.empty(cap, dtype=np.float64)
    rc = lib().kpo_distance_summary(_p(m1, C.c_double), r1, _p(m2, C.c_double), r2, d,
                                    _p(metric, C.c_double), kind, p, 1 if normalize else 0, keep_at_most,
                                    _p(stats, C.c_double), _p(offs, C.c_uint64), _p(idx, C.c_uint32),
                                    _p(dist, C.c_double), _p(z, C.c_double), cap)
    if rc != 0:
        raise RuntimeError("kpo_distance_summary failed")
    t = int(offs[-1])
    return stats, offs, idx[:t].copy(), dist[:t].copy(), z[:t].copy()


# ----------------------------------------------------- k-mer database (lib/KMerDB.ml)
def _columns(columns):
    cols = [_c(v, np.int32) for v in columns]
    n_rows = cols[0].size if cols else 0
    ptrs = (C.c_void_p * max(len(cols), 1))(*[v.ctypes.data for v in cols])
    return cols, ptrs, n_rows


def counter_stats(columns, threshold=1., power=1.):
    cols, ptrs, n_rows = _columns(columns)
    cs = np.zeros((max(len(cols), 1), 4))
    rs = np.zeros((max(n_rows, 1), 4))
    lib().kpo_counter_stats(ptrs, len(cols), n_rows, threshold, power, _p(cs, C.c_double), _p(rs, C.c_double))
    return cs[:len(cols)], rs[:n_rows]


def counter_transform(columns, col_stats, which=1, threshold=1., power=1., kmer_major=True):
    cols = [_c(v, np.int32) for v in columns]
    n_rows = cols[0].size if cols else 0
    col_stats = _c(col_stats, np.float64)
    out = np.zeros((len(cols), n_rows))
    L = lib()
    for c, v in enumerate(cols):
        cs = _p(col_stats[c], C.c_double)
        for r in range(n_rows):
            out[c, r] = L.kpo_counter_transform_one(which, threshold, power, cs, int(v[r]))
    return np.ascontiguousarray(out.T) if kmer_major else out


def counter_combine(columns, sel, col_sum, criterion=0):
    cols, ptrs, n_rows = _columns(columns)
    sel = _c(sel, np.uint32)
    col_sum = _c(col_sum, np.float64)
    out = np.zeros(max(n_rows, 1), dtype=np.int32)
    z = np.zeros(1)
    norm = lib().kpo_counter_combine(ptrs, n_rows, _p(sel if sel.size else np.zeros(1, np.uint32), C.c_uint32), sel.size,
                                     _p(col_sum if col_sum.size else z, C.c_double), criterion, out.ctypes.data)
    return out[:n_rows], norm


def c_g15(x):
    """C printf("%.15g") -- what OCaml's Printf produces.  Unlike Python it prints the sign of a NaN, and the
    NaN an x86-64 host gets from 0/0 has its sign bit set: "-nan"."""
    import math
    import struct
    x = float(x)
    if math.isnan(x):
        return "-nan" if struct.pack(">d", x)[0] & 0x80 else "nan"
    return "%.15g" % x


def format_summary_line(name, stats, names, idx, dist, z):
    """lib/Matrix.ml:684-690: all %.15g, names as stored."""
    s = "%s\t%s\t%s\t%s\t%s" % (name, c_g15(stats[0]), c_g15(stats[1]), c_g15(stats[2]), c_g15(stats[3]))
    for i, d, zz in zip(idx, dist, z):
        s += "\t%s\t%s\t%s" % (names[int(i)], c_g15(d), c_g15(zz))
    return s + "\n"


# ------------------------------------------------------------------ pipeline
def pipeline(bases, offsets, k, T_dims_major, col_hash, classes, metric, content=DNA_DS, kind=EUCLIDEAN,
             p=2., normalize_counts=True, normalize_distance=True, threads=1):
    bases = _c(bases, np.uint8)
    offsets = _c(offsets, np.uint64)
    T = _c(T_dims_major, np.float64)
    col_hash = _c(col_hash, np.uint64)
    classes = _c(classes, np.float64)
    metric = _c(metric, np.float64)
    n = len(offsets) - 1
    n_dims, n_cols = T.shape
    tw = np.zeros((n, n_dims), dtype=np.float64)
    di = np.zeros((n, classes.shape[0]), dtype=np.float64)
    secs = lib().kpo_pipeline(_p(bases, C.c_uint8), _p(offsets, C.c_uint64), n, k, content,
                              _p(T, C.c_double), n_cols, n_dims, _p(col_hash, C.c_uint64),
                              _p(classes, C.c_double), classes.shape[0], _p(metric, C.c_double), kind, p,
                              1 if normalize_counts else 0, 1 if normalize_distance else 0, threads,
                              _p(tw, C.c_double), _p(di, C.c_double))
    return tw, di, secs
