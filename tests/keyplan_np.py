"""numpy statement of the compacted keys (include/ibu_hip.h: ibu_key_plan_init / ibu_records_compact / ibu_records_expand),
written from the header's description alone: the checker of the GPU parity tests and the stand-in of the CPU (gloo)
rehearsal of the multi-GPU sort.  Test infrastructure — nothing in the product imports it."""
import numpy as np

REC = np.dtype([("barcode", "<u8"), ("umi", "<u8"), ("index", "<u8")])
FIELDS = ("barcode", "umi", "index")


class Plan:
    """sel[j] = (field, byte) of element byte j: the varying bytes of the key, least significant first (index bytes first,
    barcode bytes last); base[f] = field f with its varying bytes cleared."""

    def __init__(self, or_words, and_words):
        self.sel, self.base, self.index_bytes = [], [int(a) for a in and_words], 0
        for f in (2, 1, 0):
            varying = int(or_words[f]) ^ int(and_words[f])
            for b in range(8):
                if (varying >> (8 * b)) & 0xFF:
                    self.sel.append((f, b))
                    self.base[f] &= ~(0xFF << (8 * b)) & (2**64 - 1)
            if f == 2:
                self.index_bytes = len(self.sel)
        self.k = len(self.sel)


def census_words(recs):
    """(OR[3], AND[3]) of a record array; the identities (0, ~0) for an empty one."""
    if len(recs) == 0:
        return [0, 0, 0], [2**64 - 1] * 3
    return ([int(np.bitwise_or.reduce(recs[f])) for f in FIELDS], [int(np.bitwise_and.reduce(recs[f])) for f in FIELDS])


def compact(plan, recs):
    """records -> uint8 [n, 12] elements."""
    assert plan.k <= 12
    raw = np.frombuffer(recs.tobytes(), dtype=np.uint8).reshape(-1, 24)
    out = np.zeros((len(recs), 12), dtype=np.uint8)
    for j, (f, b) in enumerate(plan.sel):
        out[:, j] = raw[:, 8 * f + b]
    return out


def expand(plan, elems):
    """uint8 [n, 12] elements -> records."""
    n = len(elems)
    raw = np.zeros((n, 24), dtype=np.uint8)
    for f in range(3):
        raw[:, 8 * f:8 * f + 8] = np.frombuffer(np.uint64(plan.base[f]).tobytes(), dtype=np.uint8)
    for j, (f, b) in enumerate(plan.sel):
        raw[:, 8 * f + b] = elems[:, j]
    return np.frombuffer(raw.tobytes(), dtype=REC)
