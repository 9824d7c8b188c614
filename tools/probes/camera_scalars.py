"""Order of operations of the float32 scalars of CameraModel (np.linalg.norm of a 3-vector, the 3x3 @ 3x1 matmul): host probe."""
import itertools, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import default_rng
f32, f64 = np.float32, np.float64
rng = default_rng(0)

def fma32(a, b, c):   # f32 fma through float64 (exact product; the sum rounds twice only in 2^-29 of the cases)
    return f32(f64(a) * f64(b) + f64(c))

# 1. norm
cands = {
 'f32_seq_nofma': lambda v: f32(np.sqrt(f32(f32(f32(v[0]*v[0]) + f32(v[1]*v[1])) + f32(v[2]*v[2])))),
 'f32_fma_chain': lambda v: f32(np.sqrt(fma32(v[2], v[2], fma32(v[1], v[1], f32(v[0]*v[0]))))),
 'f64_acc_of_f32_products': lambda v: f32(np.sqrt(f32(f64(f32(v[0]*v[0])) + f64(f32(v[1]*v[1])) + f64(f32(v[2]*v[2]))))),
 'f64_exact': lambda v: f32(np.sqrt(f32(f64(v[0])*f64(v[0]) + f64(v[1])*f64(v[1]) + f64(v[2])*f64(v[2])))),
}
bad = {k: 0 for k in cands}
for _ in range(20000):
    v = rng.normal(0, 1, 3).astype(f32)
    want = np.linalg.norm(v)
    for k, fn in cands.items():
        bad[k] += int(fn(v) != want)
print('norm', bad, type(want))
# 2. 3x3 @ 3x1
def orders():
    out = {}
    for perm in itertools.permutations(range(3)):
        a, b, c = perm
        out[f'nofma{perm}'] = lambda R, s, a=a, b=b, c=c: f32(f32(f32(R[a]*s[a]) + f32(R[b]*s[b])) + f32(R[c]*s[c]))
        out[f'fma{perm}'] = lambda R, s, a=a, b=b, c=c: fma32(R[c], s[c], fma32(R[b], s[b], f32(R[a]*s[a])))
        out[f'fma_pair{perm}'] = lambda R, s, a=a, b=b, c=c: f32(fma32(R[a], s[a], f32(R[b]*s[b])) + f32(R[c]*s[c]))
    return out
cands = orders()
bad = {k: 0 for k in cands}
for _ in range(5000):
    R = rng.normal(0, 1, (3, 3)).astype(f32)
    s = rng.normal(0, 1000, (3, 1)).astype(f32)
    want = np.matmul(R, s)
    for i in range(3):
        for k, fn in cands.items():
            bad[k] += int(fn(R[i], s[:, 0]) != want[i, 0])
print('3x3 @ 3x1', {k: v for k, v in bad.items() if v < 200})
# R^T view @ column
bad = {k: 0 for k in cands}
for _ in range(5000):
    R = rng.normal(0, 1, (3, 3)).astype(f32)
    s = rng.normal(0, 1000, (3, 1)).astype(f32)
    want = np.matmul(R.transpose(), s)
    for i in range(3):
        for k, fn in cands.items():
            bad[k] += int(fn(R[:, i], s[:, 0]) != want[i, 0])
print('3x3^T @ 3x1', {k: v for k, v in bad.items() if v < 200})
# 2x2 @ 2x4 (corners)
b1 = b2 = 0
for _ in range(5000):
    A = rng.normal(0, 1, (2, 2)).astype(f32)
    P = rng.normal(0, 1000, (4, 2)).astype(f32)
    want = np.matmul(A, P.transpose())[0]
    for j in range(4):
        b1 += int(fma32(A[0, 1], P[j, 1], f32(A[0, 0]*P[j, 0])) != want[j])
        b2 += int(f32(f32(A[0, 0]*P[j, 0]) + f32(A[0, 1]*P[j, 1])) != want[j])
print('2x2 @ 2x4: fma(a1,y,a0 x) bad', b1, 'nofma bad', b2)
