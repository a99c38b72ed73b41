"""CPU checks of host-side logic and of arithmetic claims the GPU kernels rest on (no GPU, no HIP library calls)."""
import numpy as np
import torch


def test_placement_corners_orders_and_pads():
    """optimization.placement_corners: a placement matrix with <= 3 non-zeros per row (compute_nearest_points with use_barycentric,
    reference optimization.py:494-523) as ascending corner ids + weights; rows with fewer non-zeros get zero-weight corners."""
    from uuo_mocap_amd.optimization import placement_corners

    P = torch.zeros(5, 6890)
    P[0, [4000, 17, 220]] = torch.tensor([0.2, 0.5, 0.3])
    P[1, [6889, 0]] = torch.tensor([0.75, 0.25])          # two non-zeros
    P[2, 1234] = 1.0                                        # one-hot
    P[3, [10, 11, 12]] = torch.tensor([-0.1, 0.6, 0.5])    # a negative weight (outside the triangle) keeps its sign
    # row 4: all zero (a marker that is never placed)
    i3, b3 = placement_corners(P)
    assert i3.dtype == torch.int32 and i3.shape == (5, 3) and b3.shape == (5, 3)
    assert bool((i3[:, 1:] >= i3[:, :-1]).all())
    dense = torch.zeros_like(P)
    for m in range(5):
        for k in range(3):
            dense[m, int(i3[m, k])] += b3[m, k]
    assert torch.equal(dense, P)
    assert i3[0].tolist() == [17, 220, 4000] and b3[0].tolist() == [0.5, 0.30000001192092896, 0.20000000298023224]
    assert int((b3[1] != 0).sum()) == 2 and int((b3[2] != 0).sum()) == 1 and int((b3[4] != 0).sum()) == 0


def test_fp16_split_products_are_fp32_class():
    """The arithmetic of k_skin3 (csrc/smpl_kernels.hip), emulated in numpy: both operands times a power of two, split into
    hi = fp16(x), lo = fp16(x - hi); sum a_hi b_hi + (sum a_hi b_lo + sum a_lo b_hi) with fp32 accumulation of exact products.
    On blend-like data (217 features against a basis column) the result must be at least as close to the float64 sum as a plain
    fp32 fma chain, and the split itself must reproduce every operand to 2^-21 relative."""
    rng = np.random.default_rng(7)
    K, N = 217, 4096
    a = np.concatenate([rng.normal(0, 0.3, 207), rng.normal(0, 1.5, 10)]).astype(np.float32)          # R - I entries | betas
    B = (rng.normal(0, 1, (K, N)) * np.concatenate([np.full(207, 3e-3), np.full(10, 2e-2)])[:, None]).astype(np.float32)
    sa = np.float32(128.0)
    e = 8 - int(np.frexp(np.abs(B).max())[1])
    sb = np.float32(np.ldexp(1.0, e))
    assert 128.0 <= np.abs(B).max() * sb < 256.0

    def split(x):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
        return hi, lo

    a_hi, a_lo = split(a * sa)
    b_hi, b_lo = split(B * sb)
    assert np.all(np.isfinite(a_hi.astype(np.float32))) and np.all(np.isfinite(b_hi.astype(np.float32)))
    rec_a = (a_hi.astype(np.float64) + a_lo.astype(np.float64)) / float(sa)
    rec_b = (b_hi.astype(np.float64) + b_lo.astype(np.float64)) / float(sb)
    assert np.abs(rec_a - a).max() <= 2.0 ** -21 * np.abs(a).max() and np.abs(rec_b - B).max() <= 2.0 ** -21 * np.abs(B).max()
    # fp32 accumulation of the exact products (a product of two fp16 numbers is exact in fp32), in K order
    big = np.zeros(N, np.float32)
    small = np.zeros(N, np.float32)
    for k in range(K):
        big += a_hi[k].astype(np.float32) * b_hi[k].astype(np.float32)
        small += a_hi[k].astype(np.float32) * b_lo[k].astype(np.float32)
        small += a_lo[k].astype(np.float32) * b_hi[k].astype(np.float32)
    split_sum = (big + small) * np.float32(1.0 / (float(sa) * float(sb)))
    fma32 = np.zeros(N, np.float32)
    for k in range(K):
        fma32 = (fma32.astype(np.float64) + np.float64(a[k]) * B[k].astype(np.float64)).astype(np.float32)  # one rounding per step = fma
    exact = a.astype(np.float64) @ B.astype(np.float64)
    err_split, err_fma = np.abs(split_sum - exact), np.abs(fma32 - exact)
    # the blend alone: within a small factor of an fp32 fma chain (three representation / truncation terms of 2^-22 each) ...
    assert err_split.mean() <= 2.5 * err_fma.mean() and err_split.max() < 1.2e-7  # (offsets up to 0.44 m here; 1.2e-7 = fp32 spacing near 1 m)
    # ... and the vertex: k_skin2 accumulates the chain ONTO the template (a rounding at the template's magnitude per step),
    # k_skin3 adds the template once at the end -- which is why its vertices are the closer ones (test_skin16_is_as_close_...)
    t = rng.uniform(-1.0, 1.0, N).astype(np.float32)
    onto = t.copy()
    for k in range(K):
        onto = (onto.astype(np.float64) + np.float64(a[k]) * B[k].astype(np.float64)).astype(np.float32)
    last = (split_sum.astype(np.float64) + t.astype(np.float64)).astype(np.float32)
    truth = t.astype(np.float64) + exact
    assert np.abs(last - truth).mean() < 0.6 * np.abs(onto - truth).mean()
    assert np.abs(last - truth).max() <= np.abs(onto - truth).max()
