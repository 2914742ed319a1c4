"""Structure check of the ordered traversal's 8-wide quantised tree (raytracer_amd/csrc/host_bvh.hpp Node8Q), shared by the
host-builder test (CPU) and the device-builder test (GPU)."""
import numpy as np


def prim_box(rec, is_sphere):
    if is_sphere:
        return rec[:3] - rec[3], rec[:3] + rec[3]
    p = rec.reshape(3, 3)
    return p.min(0), p.max(0)


def check_own_tree(nodes, f2f, flat_pos, flat_sph, n, max_leaf=3):
    """Every child box, decoded exactly as the kernel does (p + q * 2^(e-127)), contains the child's primitives; every
    primitive is in exactly one leaf; leaves hold 1..max_leaf primitives.  Returns the number of levels below the root."""
    flat_pos = np.asarray(flat_pos, np.float64)
    assert sorted(np.asarray(f2f).tolist()) == list(range(n))
    boxes = [prim_box(flat_pos[f2f[j]], flat_sph[f2f[j]]) for j in range(n)]
    plo = np.array([b[0] for b in boxes])
    phi = np.array([b[1] for b in boxes])
    seen = np.zeros(n, int)
    visited = np.zeros(len(nodes), int)
    depth = 0
    # (node, child) boxes must contain everything below: push the constraint down instead of collecting primitives up
    stack = [(0, np.full(3, -np.inf), np.full(3, np.inf), 0)]
    while stack:
        node, alo, ahi, d = stack.pop()
        visited[node] += 1
        depth = max(depth, d)
        w = nodes[node]
        p = w[:3].view(np.float32).astype(np.float64)
        ex = [(int(w[3]) >> (8 * a)) & 0xFF for a in range(3)]
        step = np.array([2.0 ** (e - 127) for e in ex])
        q = w[12:24].view(np.uint8).reshape(6, 8)                     # qlo_x qlo_y qlo_z qhi_x qhi_y qhi_z
        kids = 0
        for k in range(8):
            ref = int(w[4 + k])
            if ref == 0xFFFFFFFF:
                continue
            kids += 1
            lo = np.maximum(alo, p + q[0:3, k] * step)               # inside this child box AND every ancestor's
            hi = np.minimum(ahi, p + q[3:6, k] * step)
            if ref & 0x80000000:
                start, cnt = ref & ((1 << 27) - 1), (ref >> 27) & 15
                assert 1 <= cnt <= max_leaf and start + cnt <= n
                seen[start:start + cnt] += 1
                assert (plo[start:start + cnt] >= lo).all() and (phi[start:start + cnt] <= hi).all(), (node, k)
            else:
                assert 0 < ref < len(nodes)
                stack.append((ref, lo, hi, d + 1))
        assert kids >= 1
    assert (seen == 1).all()
    assert (visited == 1).all()
    return depth
