#!/usr/bin/env python3
"""tools/exp/analyse_fuzz_record_r03ar.py — what the oracle's wrong FIRST answer in profiles/r03ar_fuzz_mismatch_record.npz corresponds to (round 3's one one-off:
UYVY 101x36 -> I420 59x44, bicubic; the library's two answers and the oracle's second are equal, the oracle's first differs in ONE byte: V plane, row 2, column 29).
Re-creates the oracle's V-plane pipeline in numpy (stage 1: the two source rows' V bytes averaged; stage 2: 4-tap horizontal pass with the oracle's own taps,
2-tap vertical pass), checks that it reproduces the RIGHT answer, then searches for the single intermediate perturbation that reproduces the wrong one exactly."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'gstreamer-metal_amd'))
import numpy as np
import oracle_lib as ol
d = np.load(os.path.join(ROOT, 'profiles', 'r03ar_fuzz_mismatch_record.npz'))
raw, want2, want1 = d['raw'], d['want2'], d['want']
o = ol.load(); L = o.lib if hasattr(o, 'lib') else ol.load().lib
w, h, ow, oh = 101, 36, 59, 44
cw, ch, ocw, och = 51, 18, 30, 22
# stage 1 V plane (UYVY: U Y V Y -> vo = 2)
r = raw.reshape(h, 204).astype(int)
V = np.zeros((ch, cw), int)
for j in range(ch):
    r0, r1 = r[2 * j], r[min(2 * j + 1, h - 1)]
    V[j] = (r0[2::4][:cw] + r1[2::4][:cw] + 1) >> 1
n = 4
idx = (C.c_int * (n * ocw))(); tp = (C.c_int * (n * ocw))()
L.gst114_linear_ntaps.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
print('ntaps', L.gst114_linear_ntaps(cw, ocw, idx, tp, n * ocw))
idx = np.array(idx).reshape(ocw, n); tp = np.array(tp).reshape(ocw, n)
print('col 29 idx', idx[29], 'taps', tp[29])
def run(tp_):
    tmp = np.zeros((ch, ocw), int)
    for x in range(ocw):
        acc = sum(V[:, idx[x, l]] * tp_[x, l] for l in range(n))
        tmp[:, x] = np.clip((acc + 32) >> 6, 0, 255)
    out = np.zeros((och, ocw), int)
    i0 = C.c_int(); i1 = C.c_int(); wt = C.c_int()
    for y in range(och):
        L.gst114_vtaps(ch, och, y, C.byref(i0), C.byref(i1), C.byref(wt))
        s1, s2 = tmp[i0.value], tmp[i1.value]
        out[y] = (s1 + (((s2 - s1) * wt.value + 128) >> 8)) & 0xff
    return tmp, out
ys, cs = 60, 32
voff = ys * oh + cs * och
def plane(buf): return buf[voff:voff + cs * och].reshape(och, cs)[:, :ocw].astype(int)
tmp, out = run(tp)
print('emulation == second (right) answer:', np.array_equal(out, plane(want2)), ' first answer differs at', np.argwhere(plane(want1) != plane(want2)))
# which single perturbations reproduce the FIRST answer exactly?
hits = []
for l in range(n):
    for l2 in range(n):
        if l == l2: continue
        t2 = tp.copy(); t2[29, l] += 1; t2[29, l2] -= 1
        if np.array_equal(run(t2)[1], plane(want1)): hits.append(('taps', l, l2))
for rr in range(ch):
    for cc in range(46, 51):
        for dv in (-2, -1, 1, 2):
            V[rr, cc] += dv
            if np.array_equal(run(tp)[1], plane(want1)): hits.append(('V', rr, cc, dv))
            V[rr, cc] -= dv
print('perturbations that reproduce the first answer exactly:', hits)
