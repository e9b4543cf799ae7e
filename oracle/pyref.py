"""Second, independent restatement of the hot path in pure Python (small cases only).

TEST INFRASTRUCTURE ONLY.  It deliberately shares no code with
oracle/kpop_oracle.c: it works on strings and dicts the way the OCaml does
(Hashtbl / IntMap / Multimap), so the two restatements check each other.
Citations are into /root/reference.
"""
import math

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def kmer_hash(s):
    """Big-endian 2-bit packing, A0 C1 G2 T3 (declared encoding)."""
    h = 0
    for ch in s:
        h = h * 4 + _CODE[ch]
    return h


def count_read(seq, k, double_stranded=True):
    """bin/KPopCount.ml:38 KIH.iterc: dict of hash -> count; windows holding a
    non-ACGT symbol contribute nothing."""
    seq = seq.upper()
    res = {}
    for i in range(len(seq) - k + 1):
        w = seq[i:i + k]
        if any(ch not in _CODE for ch in w):
            continue
        h = kmer_hash(w)
        if double_stranded:
            rc = "".join(_COMP[ch] for ch in reversed(w))
            h = min(h, kmer_hash(rc))
        res[h] = res.get(h, 0) + 1
    return res


def to_hex(h, k):
    return "%0*x" % ((k + 1) // 2, h)


def spectrum_text(label, table, k):
    """bin/KPopCount.ml:45-46: header '\\t<label>' then 'hex\\tcount' lines."""
    out = ["\t%s\n" % label]
    for h in sorted(table):
        out.append("%s\t%d\n" % (to_hex(h, k), table[h]))
    return "".join(out)


def parse_spectra(text):
    """lib/Twister.ml:97-118: list of (label, [(name, value_string)])."""
    spectra, cur = [], None
    for n, line in enumerate(text.split("\n")):
        if line == "" and n == len(text.split("\n")) - 1:
            break
        f = line.split("\t")
        if len(f) != 2:
            raise ValueError("Wrong_number_of_columns(%d,%d,2)" % (n + 1, len(f)))
        if n == 0 and f[0] != "":
            raise ValueError("Header_expected")
        if f[0] == "":
            cur = (strip_quotes(f[1]), [])
            spectra.append(cur)
        else:
            cur[1].append((f[0], f[1]))
    return spectra


def strip_quotes(s):
    if len(s) >= 2 and s[0] == '"' and s[-1] == '"':
        s = s[1:-1]
    if '"' in s:
        raise ValueError("Quotes_in_name")
    return s


def twist(twister_rows, col_names, lines, normalize=True):
    """lib/Twister.ml:146-188 for one spectrum. twister_rows[d][c]; lines = [(name, value)]."""
    name_to_idx = {}
    for i, nm in enumerate(col_names):
        name_to_idx[nm] = i  # Hashtbl.add: later binding shadows
    s_v, acc = {}, 0.0
    for name, v in reversed(lines):  # rev_lines
        idx = name_to_idx.get(name)
        if idx is None:
            continue
        v = float(v)
        acc = acc + v
        s_v[idx] = (s_v[idx] + v) if idx in s_v else v
    if normalize and acc != 0.0:
        s_v = {i: el / acc for i, el in s_v.items()}
    order = sorted(s_v)
    res = []
    for row in twister_rows:
        a = 0.0
        for j in order:
            a = a + row[j] * s_v[j]
        res.append(a)
    return res


def metric_powers(inertia, pi=1.0, thr=1.0, pe=2.0):
    v = [abs(x) ** pi for x in inertia]
    total = sum(v)
    run, kept = 0.0, []
    for x in v:
        kept.append(0.0 if run >= thr * total else x)
        run += x
    v = [abs(x) ** pe for x in kept]
    s = sum(v)
    return [x / s for x in v] if s != 0.0 else v


def _scale(kind, p, x):
    if kind == "euclidean":
        return math.sqrt(x)
    if kind == "cosine":
        return x / 2.0
    return x ** (1.0 / p)


def norm(kind, p, metric, v):
    acc = 0.0
    for i, el in enumerate(v):
        if kind == "minkowski":
            acc = acc + (abs(el) ** p) * metric[i]
        else:
            acc = acc + (el * el * metric[i])
    n = _scale(kind, p, acc)
    return 1.0 if n == 0.0 else n  # lib/Matrix.ml:67


def distance(kind, p, metric, a, na, b, nb):
    acc = 0.0
    for i in range(len(a)):
        diff = a[i] / na - b[i] / nb
        if kind == "minkowski":
            acc = acc + ((abs(diff) ** p) * metric[i])
        else:
            acc = acc + (diff * diff * metric[i])
    return _scale(kind, p, acc)


def distance_rowwise(m1, m2, metric, kind="euclidean", p=2.0, normalize=True):
    n1 = [norm(kind, p, metric, r) if normalize else 1.0 for r in m1]
    n2 = [norm(kind, p, metric, r) if normalize else 1.0 for r in m2]
    return [[distance(kind, p, metric, m1[i], n1[i], m2[j], n2[j]) for i in range(len(m1))]
            for j in range(len(m2))]


def summarize_row(row, req_len):
    """lib/Matrix.ml:632-690."""
    n = len(row)
    distr = {}
    for c, d in enumerate(row):
        distr.setdefault(d, []).append(c)
    eff_len, median_pos, median, acc = 0, n // 2, 0.0, 0.0
    for d in sorted(distr):
        sl = len(distr[d])
        acc = acc + (float(sl) * d)
        if median_pos >= 0 and median_pos - sl < 0:
            median = d
        median_pos -= sl
        if eff_len < req_len:
            eff_len += sl
    mean = acc / float(n) if n > 0 else 0.0
    acc, dd = 0.0, {}
    for d in row:
        x = d - mean
        acc = acc + (x * x)
        y = abs(d - median)
        dd[y] = dd.get(y, 0) + 1
    median_pos, mad = n // 2, 0.0
    for y in sorted(dd):
        if median_pos >= 0 and median_pos - dd[y] < 0:
            mad = y
        median_pos -= dd[y]
    sd = math.sqrt(acc / (float(n) - 1.0)) if n > 1 else 0.0
    flat = [(d, c) for d in sorted(distr) for c in sorted(distr[d])][:eff_len]
    neigh = []
    for d, c in flat:
        try:
            z = (d - mean) / sd
        except ZeroDivisionError:
            z = float("nan") if d == mean else math.copysign(float("inf"), d - mean)
        neigh.append((c, d, z))
    return (mean, sd, median, mad), neigh
