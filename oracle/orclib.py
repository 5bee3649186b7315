"""ctypes binding of the CPU oracle (oracle/hgx_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "liborc.so")


class OrcLocus(C.Structure):
    _fields_ = [("n_alleles", C.c_int32), ("n_vars", C.c_int32), ("var_pos", C.c_void_p),
                ("var_right", C.c_void_p), ("var_linked", C.c_void_p), ("link_off", C.c_void_p),
                ("link_allele", C.c_void_p)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_score_pairs.restype = C.c_int
        lib.orc_dedup.restype = C.c_int64
        lib.orc_single_abundance.restype = C.c_int

    def make_locus(self, tables):
        """tables: dict with n_alleles, var_pos, var_right, var_linked, link_off, link_allele (numpy)."""
        keep = {k: np.ascontiguousarray(tables[k], dtype=dt) for k, dt in
                (("var_pos", np.int32), ("var_right", np.int32), ("var_linked", np.uint8),
                 ("link_off", np.int32), ("link_allele", np.int32))}
        L = OrcLocus(int(tables["n_alleles"]), len(keep["var_pos"]), _p(keep["var_pos"]), _p(keep["var_right"]),
                     _p(keep["var_linked"]), _p(keep["link_off"]), _p(keep["link_allele"]))
        L._keep = keep
        return L

    def score_pairs(self, L, exon_keys, gene_keys, pair_off, level, left, right, id_off, ids,
                    want_exon=True, want_gene=True):
        A = L.n_alleles
        w64 = (A + 63) // 64
        n_pairs = len(pair_off) - 1
        exon_keys = np.ascontiguousarray(exon_keys, dtype=np.uint8)
        gene_keys = np.ascontiguousarray(gene_keys, dtype=np.uint8)
        pair_off = np.ascontiguousarray(pair_off, dtype=np.int32)
        level = np.ascontiguousarray(level, dtype=np.uint8)
        left = np.ascontiguousarray(left, dtype=np.int32)
        right = np.ascontiguousarray(right, dtype=np.int32)
        id_off = np.ascontiguousarray(id_off, dtype=np.int32)
        ids = np.ascontiguousarray(ids if len(ids) else [0], dtype=np.int32)
        eb = np.zeros((n_pairs, w64), dtype=np.uint64)
        gb = np.zeros((n_pairs, w64), dtype=np.uint64)
        gc = np.zeros(A, dtype=np.int64)
        fp = np.zeros(A, dtype=np.int32)
        rc = self.lib.orc_score_pairs(C.byref(L), _p(exon_keys), _p(gene_keys), C.c_int32(n_pairs), _p(pair_off),
                                      _p(level), _p(left), _p(right), _p(id_off), _p(ids), C.c_int32(w64),
                                      _p(eb) if want_exon else None, _p(gb) if want_gene else None, _p(gc), _p(fp))
        assert rc == 0
        return eb, gb, gc, fp

    def dedup(self, rows, weight=None, and_mask=None):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        n, w64 = rows.shape
        ub = np.zeros((max(n, 1), w64), dtype=np.uint64)
        uc = np.zeros(max(n, 1), dtype=np.int64)
        fr = np.zeros(max(n, 1), dtype=np.int64)
        wt = None if weight is None else np.ascontiguousarray(weight, dtype=np.int64)
        am = None if and_mask is None else np.ascontiguousarray(and_mask, dtype=np.uint64)
        k = self.lib.orc_dedup(_p(rows), _p(wt) if wt is not None else None, C.c_int64(n), C.c_int32(w64),
                               _p(am) if am is not None else None, _p(ub), _p(uc), _p(fr))
        assert k >= 0
        return ub[:k].copy(), uc[:k].copy(), fr[:k].copy()

    def single_abundance(self, n_alleles, classes, counts, remove_low, lengths=None):
        """classes: list of allele-index lists in key order.  Returns ([allele...], [prob...], n_iter) or raises KeyError."""
        off = np.zeros(len(classes) + 1, dtype=np.int32)
        for i, c in enumerate(classes):
            off[i + 1] = off[i] + len(c)
        flat = np.ascontiguousarray([a for c in classes for a in c] or [0], dtype=np.int32)
        return self.single_abundance_flat(n_alleles, off, flat, counts, remove_low, lengths)

    def single_abundance_flat(self, n_alleles, off, flat, counts, remove_low, lengths=None):
        """The same with the classes as (offsets [C + 1], alleles) arrays."""
        off = np.ascontiguousarray(off, dtype=np.int32)
        flat = np.ascontiguousarray(flat if len(flat) else [0], dtype=np.int32)
        n_classes = len(off) - 1
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        ln = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.int32)
        oa = np.zeros(max(n_alleles, 1), dtype=np.int32)
        op = np.zeros(max(n_alleles, 1), dtype=np.float64)
        it = C.c_int32(0)
        k = self.lib.orc_single_abundance(C.c_int32(n_alleles), C.c_int32(n_classes), _p(off), _p(flat), _p(cnt),
                                          C.c_int32(1 if remove_low else 0), _p(ln) if ln is not None else None,
                                          _p(oa), _p(op), C.byref(it))
        if k == -4:
            raise KeyError("reference KeyError (quirk Q6)")
        assert k >= 0
        return oa[:k].copy(), op[:k].copy(), it.value


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def load():
    src = os.path.join(ROOT, "oracle", "hgx_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        build()
    return Oracle(C.CDLL(LIB))
