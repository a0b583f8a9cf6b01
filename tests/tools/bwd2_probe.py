import sys, numpy as np
sys.path.insert(0, '.')
from envidr_amd import scenes
from tests.util import run_op, rel_l2
sc = scenes.toaster_scene()
rng = np.random.default_rng(3)
B, D, C, L = 36608, 3, 2, 16
d = rng.normal(size=(B, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
x = ((d * rng.uniform(0.44, 0.56, size=(B, 1)) + 1) / 2).astype(np.float32)
S = float(np.log2(sc.per_level_scale)); H = 16
offs = np.ascontiguousarray(sc.offsets, np.int32)
out = np.zeros((L, B, C), np.float32); dy = np.zeros((B, L * D * C), np.float32)
run_op("oracle", "hash_encode_forward", x, sc.table, offs, out, B, D, C, L, S, H, 1, dy)
o_or = run_op("oracle", "hash_encode_forward", x, sc.table, offs, out, B, D, C, L, S, H, 1, dy)
dy = o_or[4]
grad = rng.normal(size=(L, B, C)).astype(np.float32)
ggi = rng.normal(size=(B, D)).astype(np.float32)
args = (grad, x, sc.table, offs, B, D, C, L, S, H, 1, dy, ggi, np.zeros((L, B, C), np.float32), np.zeros_like(sc.table))
a = run_op("oracle", "hash_encode_second_backward", *args)
b = run_op("hip", "hash_encode_second_backward", *args)
print("grad_grad rel", rel_l2(b[-2], a[-2]))
for l in range(L):
    print(l, "grad2_emb rel", rel_l2(b[-1][offs[l]:offs[l+1]], a[-1][offs[l]:offs[l+1]]), "grad_grad level", rel_l2(b[-2][l], a[-2][l]))
# first-order table gradient too
args1 = (grad, x, sc.table, offs, np.zeros_like(sc.table), B, D, C, L, S, H, 1, dy, np.zeros((B, D), np.float32))
a1 = run_op("oracle", "hash_encode_backward", *args1); b1 = run_op("hip", "hash_encode_backward", *args1)
print("first-order table grad per level", [float("%.1e" % rel_l2(b1[4][offs[l]:offs[l+1]], a1[4][offs[l]:offs[l+1]])) for l in range(L)], "grad_inputs", rel_l2(b1[-1], a1[-1]))
