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


_AA = "ACDEFGHIKLMNPQRSTVWY"  # residue codes 0..19 of the declared protein hash (kpop_amd/csrc/kmer.h)


def count_read_protein(seq, k):
    """KMers.ProteinHash (bin/KPopCount.ml:246-248) under the declared encoding: 5 bits per residue, big-endian; a
    window holding anything but the 20 standard residues contributes nothing."""
    seq = seq.upper()
    res = {}
    for i in range(len(seq) - k + 1):
        w = seq[i:i + k]
        if any(ch not in _AA for ch in w):
            continue
        h = 0
        for ch in w:
            h = h * 32 + _AA.index(ch)
        res[h] = res.get(h, 0) + 1
    return res


def to_hex_protein(h, k):
    return "%0*x" % ((5 * k + 3) // 4, h)


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


# ---------------------------------------------------------------- k-mer database (lib/KMerDB.ml)
def counter_vector_stats(v, threshold=1.0, power=1.0):
    """stats_table_of_core_db.compute_one, lib/KMerDB.ml:174-215 -> (non_zero, max, sum, sum_log)"""
    s = 0.0
    for c in v:
        s += float(c) ** power
    if threshold < 1.0:
        threshold = threshold * s
    non_zero, mx, sm, sl = 0, 0, 0.0, 0.0
    for c in v:
        f = float(c)
        if f >= threshold:
            non_zero += 1
            mx = max(mx, c)
            sm += f ** power
            sl += (math.log(f) if f > 0 else -math.inf) * power
    return (float(non_zero), float(mx), sm, sl)


def _omax(a, b):
    """Stdlib.max on floats: `if a >= b then a else b`"""
    return a if a >= b else b


def _div(a, b):
    """IEEE division (Python raises on a zero divisor)"""
    if b != 0.0:
        return a / b
    if a != a or a == 0.0:
        return math.nan
    return math.copysign(math.inf, a) * math.copysign(1.0, b)


def _floor(x):
    return float(math.floor(x)) if math.isfinite(x) else x


def counter_transform_one(which, threshold, power, cs, counts):
    """Transformation.compute, lib/KMerDB.ml:96-144; which in binary/power/clr/pseudocounts"""
    non_zero, cmax, csum, csum_log = (float(x) for x in cs)
    counts = float(counts)
    if threshold < 1.0:
        threshold = threshold * csum
    if which == "binary":
        return 1.0 if counts >= threshold else 0.0
    if which == "power":
        return (counts ** power if counts >= threshold else 0.0)
    if which == "clr":
        v = counts if counts >= threshold else 0.0
        v = _omax(v, 0.1)
        return math.log(v) * power - _div(csum_log, non_zero)
    if power == 0.0:
        q = _div(counts + 1.0, threshold)
        v = cmax * (math.log(q) if q > 0 and math.isfinite(q) else (math.inf if q > 0 else -math.inf if q == 0 else math.nan))
    else:
        red = _omax(0.0, threshold - 1.0)
        c_p = red ** power
        if power < 1.0:
            v = (counts ** power - c_p) * cmax ** (1.0 - power) / power
        else:
            v = _div(counts ** power - c_p, threshold ** power - c_p)
    return _omax(0.0, _div(_floor(v), csum))


def counter_combine(columns, sel, col_sum, criterion="mean"):
    """add_combined_selected, lib/KMerDB.ml:628-736 -> (list of int32, norm)"""
    max_norm = 0.0
    for s in sel:
        max_norm = max(max_norm, col_sum[s])
    n_rows = len(columns[0]) if len(columns) else 0
    out, norm = [], 0.0
    for i in range(n_rows):
        vals = []
        for s in sel:
            if col_sum[s] > 0.0:
                vals.append(float(columns[s][i]) * max_norm / col_sum[s])
        if criterion == "mean":
            res = 0.0
            for v in vals:
                res += v
        else:
            vals.sort()
            res = (vals[len(vals) // 2] if vals else 0.0) * float(len(sel))
        norm += res
        t = int(res)  # Int32.of_float: truncate, keep the low 32 bits
        t &= 0xFFFFFFFF
        out.append(t - (1 << 32) if t >= (1 << 31) else t)
    return out, norm


# ---------------------------------------------------------------------------------------------------------------
# phylogenetic splits from embeddings: Matrix.get_splits, lib/Matrix.ml:524-612
# ---------------------------------------------------------------------------------------------------------------
def splits_gaps(emb, max_splits):
    """SplitsAlgorithm.Gaps (:527-600).  emb: rows of floats.  -> [(gap, sorted member rows)] in the reference's order: gaps by
    decreasing size, then dimension, then position.  Equal coordinates keep ascending row order (OCaml's Array.sort leaves the
    order of equal elements open; declared, see kpop_amd/csrc/splits.hip)."""
    n = len(emb)
    d = len(emb[0]) if n else 0
    perms, gaps = [], []
    for i in range(d):
        col = sorted(((emb[r][i] + 0.0, r) for r in range(n)), key=lambda t: (t[0], t[1]))  # +0.0: -0. and 0. compare equal
        perms.append([r for _, r in col])
        gaps.extend((col[j + 1][0] - col[j][0], i, j) for j in range(n - 1))  # :571
    gaps.sort(key=lambda t: (-t[0], t[1], t[2]))  # :590-601
    return [(g, sorted(perms[i][:j + 1])) for g, i, j in gaps[:max_splits]]


class _SplitMix:
    """the declared stand-in for OCaml's Random (kpop_amd/host/splits.h)"""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def boolean(self):
        return bool(self.next() & 1)

    def integer(self, n):
        return self.next() % n

    def unit(self):
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)


def _bipartition(emb, d, elements, rng):
    """SplitsAlgorithm.Bipartition.make (:361-521), acceptance_probability_at_zero 0.2, difference_magnification_factor 10"""
    inverse_acceptance, negative_scale = (1. - 0.2) / 0.2, -10.
    one, two = set(), set()
    c1, c2 = [0.] * d, [0.] * d
    for i in elements:  # IntSet.iter: ascending
        v = emb[i]
        if rng.boolean():
            two.add(i)
            c2 = [a + b for a, b in zip(c2, v)]
        else:
            one.add(i)
            c1 = [a + b for a, b in zip(c1, v)]

    def objective_of():
        n1, n2 = float(len(one)), float(len(two))
        res = 0.
        if n1 > 0. and n2 > 0.:
            for s1, s2 in zip(c1, c2):
                a = s1 / n1 if n1 > 1. else s1
                b = s2 / n2 if n2 > 1. else s2
                res = res + (max(a, b) - min(a, b))
        return res / math.sqrt(1. + abs(n1 - n2))
    objective = objective_of()
    best = (objective, set(one), set(two))
    terminator, rejected = max(len(elements), 40), 0
    while rejected < terminator:
        old_objective, o1, o2 = objective, c1, c2
        selected = elements[rng.integer(len(elements))]
        v = emb[selected]
        from_one = selected in one
        if from_one:
            one.remove(selected)
            two.add(selected)
            c1 = [a - b for a, b in zip(o1, v)]
            c2 = [a + b for a, b in zip(o2, v)]
        else:
            two.remove(selected)
            one.add(selected)
            c2 = [a - b for a, b in zip(o2, v)]
            c1 = [a + b for a, b in zip(o1, v)]
        objective = objective_of()
        delta = objective - old_objective
        score = 1. / (1. + inverse_acceptance * math.exp(negative_scale * delta))
        if rng.unit() <= score:
            rejected = 0
            if objective > best[0]:
                best = (objective, set(one), set(two))
        else:
            rejected += 1
            if from_one:
                two.remove(selected)
                one.add(selected)
            else:
                one.remove(selected)
                two.add(selected)
            c1, c2, objective = o1, o2, old_objective
    return best


def splits_centroids(emb, seed=0x4B506F70):
    """SplitsAlgorithm.Centroids (:601-612) -> [(weight, sorted member rows)] in the order the splits are added"""
    d = len(emb[0]) if emb else 0
    rng = _SplitMix(seed)
    res = []

    def refine(s):
        if len(s) > 1:
            objective, one, two = _bipartition(emb, d, sorted(s), rng)
            res.append((objective, sorted(one)))
            refine(one)
            refine(two)
        else:
            res.append((0., sorted(s)))
    refine(set(range(len(emb))))
    return res


def splits_text(names, splits, precision=10):
    """'.PhyloSplits.txt' as declared in kpop_amd/host/splits.h"""
    out = '""' + "".join('\t"%s"' % n for n in names) + "\n"
    for w, members in splits:
        out += "%.*g" % (precision, w) + "".join('\t"%s"' % names[m] for m in members) + "\n"
    return out
