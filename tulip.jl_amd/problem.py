"""LP container, free-format MPS reader and the standard form the interior-point method works on
(the front end that produces the `A` handed to KKT.setup).

  * `read_free_mps`: what /root/reference/src/Interfaces/tulip_julia_api.jl:18-39 obtains from
    QPSReader.readqps(mpsformat=:free): objective = first N row, RHS of the objective row = -obj0,
    RANGES per row type, default bounds [0, +inf), UP with a negative value and no LO => lower -inf.
  * `standard_form`: /root/reference/src/IPM/ipmdata.jl:64-173: one slack per inequality / range / free
    row appended after the structural columns, `c` negated for maximisation.
"""
import math

import numpy as np
import scipy.sparse as sp

INF = float("inf")


# ------------------------------------------------------------------------------------------------
# LP container + free MPS reader
# ------------------------------------------------------------------------------------------------
class LP:
    def __init__(self, A, obj, obj0, lcon, ucon, lvar, uvar, objsense_min=True, name=""):
        self.A = sp.csc_matrix(A)
        self.obj = np.asarray(obj, float); self.obj0 = float(obj0)
        self.lcon = np.asarray(lcon, float); self.ucon = np.asarray(ucon, float)
        self.lvar = np.asarray(lvar, float); self.uvar = np.asarray(uvar, float)
        self.objsense_min = objsense_min
        self.name = name


def read_free_mps(path):
    rows, row_type, cols = {}, [], {}
    entries = []                      # (row idx, col idx, value) ; objective in `objc`
    objc, rhs, ranges = {}, {}, {}
    lo, up = {}, {}
    obj_row, obj0, sense_min, name = None, 0.0, True, ""
    section = None
    col_order = []
    with open(path) as fh:
        for raw in fh:
            line = raw.rstrip("\n")
            if not line.strip() or line.lstrip().startswith("*"):
                continue
            if not line[0].isspace():
                tok = line.split()
                section = tok[0].upper()
                if section == "NAME" and len(tok) > 1:
                    name = tok[1]
                if section == "OBJSENSE" and len(tok) > 1:
                    sense_min = not tok[1].upper().startswith("MAX")
                if section == "ENDATA":
                    break
                continue
            tok = line.split()
            if section == "OBJSENSE":
                sense_min = not tok[0].upper().startswith("MAX")
            elif section == "ROWS":
                t, r = tok[0].upper(), tok[1]
                if t == "N":
                    if obj_row is None:
                        obj_row = r
                else:
                    rows[r] = len(row_type); row_type.append(t)
            elif section == "COLUMNS":
                if len(tok) >= 3 and tok[1].upper() == "'MARKER'":
                    continue
                c = tok[0]
                if c not in cols:
                    cols[c] = len(col_order); col_order.append(c)
                for r, v in zip(tok[1::2], tok[2::2]):
                    if r == obj_row:
                        objc[cols[c]] = objc.get(cols[c], 0.0) + float(v)
                    elif r in rows:
                        entries.append((rows[r], cols[c], float(v)))
            elif section == "RHS":
                pairs = tok[1:] if len(tok) % 2 == 1 else tok
                for r, v in zip(pairs[0::2], pairs[1::2]):
                    if r == obj_row:
                        obj0 = -float(v)
                    elif r in rows:
                        rhs[rows[r]] = float(v)
            elif section == "RANGES":
                pairs = tok[1:] if len(tok) % 2 == 1 else tok
                for r, v in zip(pairs[0::2], pairs[1::2]):
                    if r in rows:
                        ranges[rows[r]] = float(v)
            elif section == "BOUNDS":
                t = tok[0].upper()
                if t in ("FR", "MI", "PL", "BV"):
                    c = tok[2] if len(tok) >= 3 else tok[1]
                    v = None
                else:
                    c, v = (tok[2], float(tok[3])) if len(tok) >= 4 else (tok[1], float(tok[2]))
                j = cols[c]
                if t == "UP":
                    up[j] = v
                    if v < 0 and j not in lo:
                        lo[j] = -INF
                elif t == "LO":
                    lo[j] = v
                elif t == "FX":
                    lo[j] = up[j] = v
                elif t == "FR":
                    lo[j], up[j] = -INF, INF
                elif t == "MI":
                    lo[j] = -INF
                elif t == "PL":
                    up[j] = INF
                elif t == "BV":
                    lo[j], up[j] = 0.0, 1.0
    m, n = len(row_type), len(col_order)
    lcon, ucon = np.empty(m), np.empty(m)
    for i, t in enumerate(row_type):
        b = rhs.get(i, 0.0)
        if t == "E":
            lcon[i] = ucon[i] = b
            if i in ranges:
                r = ranges[i]
                lcon[i], ucon[i] = (b, b + abs(r)) if r >= 0 else (b - abs(r), b)
        elif t == "L":
            lcon[i], ucon[i] = -INF, b
            if i in ranges:
                lcon[i] = b - abs(ranges[i])
        elif t == "G":
            lcon[i], ucon[i] = b, INF
            if i in ranges:
                ucon[i] = b + abs(ranges[i])
    lvar = np.array([lo.get(j, 0.0) for j in range(n)])
    uvar = np.array([up.get(j, INF) for j in range(n)])
    obj = np.array([objc.get(j, 0.0) for j in range(n)])
    if entries:
        ri, ci, vv = zip(*entries)
        A = sp.csc_matrix((vv, (ri, ci)), shape=(m, n))
    else:
        A = sp.csc_matrix((m, n))
    return LP(A, obj, obj0, lcon, ucon, lvar, uvar, sense_min, name)


# ------------------------------------------------------------------------------------------------
# standard form: ipmdata.jl:64-173
# ------------------------------------------------------------------------------------------------
class IPMData:
    pass


def standard_form(lp):
    m, n = lp.A.shape
    b = np.zeros(m)
    sind, sval, lslack, uslack = [], [], [], []
    for i, (lb, ub) in enumerate(zip(lp.lcon, lp.ucon)):
        if lb == ub:
            b[i] = lb
        elif lb == -INF and ub == INF:
            sind.append(i); sval.append(1.0); lslack.append(-INF); uslack.append(INF); b[i] = 0.0
        elif lb == -INF and math.isfinite(ub):
            sind.append(i); sval.append(1.0); lslack.append(0.0); uslack.append(INF); b[i] = ub
        elif math.isfinite(lb) and ub == INF:
            sind.append(i); sval.append(-1.0); lslack.append(0.0); uslack.append(INF); b[i] = lb
        elif math.isfinite(lb) and math.isfinite(ub):
            sind.append(i); sval.append(1.0); lslack.append(0.0); uslack.append(ub - lb); b[i] = ub
        else:
            raise ValueError(f"Invalid bounds for row {i}: [{lb}, {ub}]")
    ns = len(sind)
    slack = sp.csc_matrix((sval, (sind, np.arange(ns))), shape=(m, ns))
    d = IPMData()
    d.A = sp.hstack([lp.A, slack], format="csc") if ns else lp.A.copy()
    d.A.sort_indices()
    d.b = b
    d.objsense = lp.objsense_min
    d.c = np.concatenate([lp.obj, np.zeros(ns)]); d.c0 = lp.obj0
    if not lp.objsense_min:
        d.c = -d.c; d.c0 = -d.c0
    d.l = np.concatenate([lp.lvar, lslack]); d.u = np.concatenate([lp.uvar, uslack])
    d.lflag = np.isfinite(d.l); d.uflag = np.isfinite(d.u)
    d.lz = np.where(d.lflag, d.l, 0.0); d.uz = np.where(d.uflag, d.u, 0.0)     # l .* lflag, u .* uflag
    d.nrow, d.ncol, d.nvar = m, n + ns, n
    return d
