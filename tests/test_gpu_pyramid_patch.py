"""Bit-exact parity of the pre-step kernels (pyramid builder, patch fetch) with the oracle."""
import numpy as np
import pytest

import invcompcamtrack_amd as ic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,lv_f,pad", [((96, 160), 3, 8), ((1080, 1920), 2, 8), ((135, 45), 2, 4),
                                            ((64, 64), 5, 4), ((37, 53), 1, 31)])
def test_pyramid_planes_bit_exact(oracle, shape, lv_f, pad):
    rng = np.random.default_rng(shape[0])
    for img in (rng.integers(0, 256, shape).astype(np.float32), rng.uniform(0, 255, shape).astype(np.float32)):
        o = oracle.Pyramid(img, lv_f, pad)
        g = ic.Pyramid(img, lv_f, pad)
        for l in range(lv_f + 1):
            assert g.level_dims(l) == (o.img[l].shape[1], o.img[l].shape[0])
            for w, ref in ((0, o.img), (1, o.dx), (2, o.dy)):
                assert np.array_equal(g.download(l, w), ref[l]), (l, w)


def test_pyramid_from_device_pointer_and_host_planes(oracle):
    import torch
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 255, (120, 200)).astype(np.float32)
    t = torch.from_numpy(img).cuda()
    g = ic.Pyramid(lv_f=2, imgpadding=8, device_ptr=t.data_ptr(), wh=(200, 120),
                   stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = oracle.Pyramid(img, 2, 8)
    for l in range(3):
        assert np.array_equal(g.download(l, 0), o.img[l]) and np.array_equal(g.download(l, 2), o.dy[l])
    h = ic.Pyramid(lv_f=2, imgpadding=8, wh=(200, 120), host_planes=(o.img, o.dx, o.dy))
    for l in range(3):
        assert np.array_equal(h.download(l, 1), o.dx[l])


def test_pyramid_rebuild_in_place_host_and_device(oracle):
    """A video loop refills one pyramid per incoming frame (run_track_nposes.cpp:180 builds one per image): same planes
    as a fresh build, bit for bit, also the packed texels the setup kernel reads (checked through a tracking)."""
    import torch
    from parity_util import scene
    sc = scene(200, 120, 150, seed=3)
    rng = np.random.default_rng(6)
    other = rng.uniform(0, 255, (120, 200)).astype(np.float32)
    g = ic.Pyramid(other, 2, 8)
    planes0 = [g.device_plane(l, w) for l in range(3) for w in range(3)] if hasattr(g, "device_plane") else None
    for source in ("host", "device"):
        for img in (sc["img_a"], other, sc["img_a"]):
            if source == "host":
                g.rebuild(img)
            else:
                t = torch.from_numpy(np.ascontiguousarray(img, np.float32)).cuda()
                g.rebuild(device_ptr=t.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
            o = oracle.Pyramid(img, 2, 8)
            for l in range(3):
                for w, ref in ((0, o.img), (1, o.dx), (2, o.dy)):
                    assert np.array_equal(g.download(l, w), ref[l]), (source, l, w)
    if planes0 is not None:
        assert planes0 == [g.device_plane(l, w) for l in range(3) for w in range(3)]
    with pytest.raises(ValueError):
        g.rebuild(np.zeros((10, 10), np.float32))
    # g now holds img_a: tracking against it == tracking against a freshly built pyramid
    op = ic.optparam(2, 0, 8, 4, 0.0, 0, 0, 150)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pb = ic.Pyramid(sc["img_b"], 2, 8)
    poses = []
    for pa in (g, ic.Pyramid(sc["img_a"], 2, 8)):
        e = ic.TrackBatch(cam, op, 1)
        e.Set3Dpoints(0, sc["pts3d"].copy())
        e.SetPose(0, sc["p_a"], pa, pb)
        e.track_async()
        poses.append(e.poses().copy())
    assert np.array_equal(poses[0], poses[1])


@pytest.mark.parametrize("psz,dpn", [(8, 0), (4, 0), (8, 1), (5, 0), (31, 0), (16, 1)])
def test_get_patch_bit_exact_incl_borders(oracle, psz, dpn):
    rng = np.random.default_rng(psz)
    h, w = 300, 400
    img = rng.uniform(0, 255, (h, w)).astype(np.float32)
    o = oracle.Pyramid(img, 1, psz)
    g = ic.Pyramid(img, 1, psz)
    op = ic.optparam(1, 0, psz, 1, 0, 0, dpn, 8)
    oop = oracle.make_op(1, 0, psz, 1, 0, 0, dpn, 8)
    for level, (ww, hh) in ((0, (w, h)), (1, (w // 2, h // 2))):
        mids = np.concatenate([
            rng.uniform([0, 0], [ww, hh], (200, 2)),
            # corners and edges are legal (inclusive bounds, odometer.cpp:273-276); integers; x >= 256 where
            # x + 1e-5f == x in float and the taps shift by one pixel (reference quirk, utilities.cpp:66-67)
            np.array([[0, 0], [ww, hh], [0, hh], [ww, 0], [17, 23], [256, 100], [300, 128.5], [299.99997, 7]],
                     np.float64)]).astype(np.float32)
        mids = mids[(mids[:, 0] <= ww) & (mids[:, 1] <= hh)]
        T = ic.util_getPatch(g, level, mids, op)
        T2, Gx, Gy = ic.util_getPatch_grad(g, level, mids, op)
        for k, m in enumerate(mids):
            ref = oracle.getpatch(o.img[level], m, oop)
            r3 = oracle.getpatch_grad(o.img[level], o.dx[level], o.dy[level], m, oop)
            if dpn:
                assert np.abs(T[k] - ref).max() <= 1e-4 and np.abs(T2[k] - r3[0]).max() <= 1e-4
            else:
                assert np.array_equal(T[k], ref) and np.array_equal(T2[k], r3[0]), (level, m)
            assert np.array_equal(Gx[k], r3[1]) and np.array_equal(Gy[k], r3[2])
    from invcompcamtrack_amd._lib import IctrError
    with pytest.raises(IctrError):
        ic.util_getPatch(g, 0, np.array([[w + 1.0, 5.0]], np.float32), op)  # outside the image: refused, not read
    assert ic.util_getPatch(g, 0, np.zeros((0, 2), np.float32), op).shape == (0, psz * psz)


@pytest.mark.parametrize("n,margin,cpw", [(700, 0.5, 0), (9000, 0.5, 0), (700, 0.5, 16), (9000, 0.5, 32), (2500, 0.5, 64)])
def test_gradients_on_the_fly_give_the_planes_bits(n, margin, cpw, monkeypatch):
    """the 8x8 setup kernel forms Gx / Gy from the IMAGE plane (I(x+1) - I(x-1) with the builder's
    reflect-101 / zero-padding rules, utilities.cpp:30-45) instead of reading the dx / dy / packed planes. Same
    subtraction, same blend: T, Gx, Gy, the coefficients and the poses must be the planes' bit for bit -- points right
    at the image border included (margin 0.5 px) -- (a) on an ordinary pyramid against variant bit 27 (which reads the planes), (b) with a reference
    pyramid that holds nothing but the image levels (getgrad = 2). 700 points: per-iteration launches (variant bit 13);
    9000 points: the resident-iteration form. cpw (points per wave chunk, ICTR_CPW) 16 / 32 / 64: the statically
    unrolled 16-patch groups of k_ref8 with the transposing reduction for S (what large batches run); 0: the chunk size
    the host picks for this problem (small: the dynamic patch loop)."""
    from invcompcamtrack_amd import synth
    if cpw:
        monkeypatch.setenv("ICTR_CPW", str(cpw))
    sc = synth.make_scene(640, 384, n_points=n, seed=21, margin=margin)
    op = ic.optparam(2, 0, 8, 4, 0.0, 0, 0, n)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    pa_img = ic.Pyramid(sc["img_a"], 2, 8, getgrad=2)
    base = 8192 if n < 8193 else 0
    res = []
    for variant, ref in ((base | (1 << 27), pa), (base, pa), (base, pa_img)):
        e = ic.TrackBatch(cam, op, 1)
        e.set_variant(variant)
        e.Set3Dpoints(0, sc["pts3d"].copy())
        e.SetPose(0, sc["p_a"], ref, pb)
        e.track_async()
        p = e.poses()
        res.append((p, e.read_buffer(0, 0, 64 * n), e.read_buffer(0, 1, 64 * n), e.read_buffer(0, 2, 64 * n),
                    e.read_buffer(0, 7, 16 * n), e.path_name()))
    assert ("k_level_resident" in res[0][5]) == (n >= 8193)
    for k in (1, 2):
        for q in range(1, 5):   # T, Gx, Gy, coefficients: bit for bit
            assert np.array_equal(res[0][q], res[k][q]), (k, q, np.abs(res[0][q] - res[k][q]).max())
        # poses: H is summed in another order by the static form (transposing reduction), nothing else differs
        # (the resident path's setup launches use chunks of at least 16 points: static form for the image plane, dynamic
        # loop for the gradient planes)
        tol = 0.0 if (cpw == 0 and n < 8193) else 2e-6
        assert np.abs(res[0][0] - res[k][0]).max() <= tol, np.abs(res[0][0] - res[k][0]).max()
    assert np.array_equal(res[1][0], res[2][0])   # planes or no planes in the pyramid: the same kernel, the same bits
    assert np.abs(res[0][1]).max() > 1 and np.abs(res[0][2]).max() > 0.1
    # border patches were really among them: some patch has an exactly-zero gradient column / row next to non-zero ones
    gx = res[0][2].reshape(n, 8, 8)
    assert np.any((np.abs(gx).sum(axis=(1, 2)) > 0) & (np.abs(gx).min(axis=(1, 2)) == 0))


@pytest.mark.parametrize("grid", [True, False])
def test_line_touches_on_small_planes(grid, monkeypatch):
    """The static form of the 8x8 setup kernel touches every cache line of a group's windows before its taps (bounding
    box of 16 patches, rows x 128-byte lines). On a 128 x 96 frame with two more levels (32 x 24 at the coarsest, patches
    all over the plane and at its border) the boxes span whole planes; scattered points have boxes that are not
    touched at all. Same bits as the dynamic patch loop (variant bit 28), which touches nothing, and as the planes
    (bit 27)."""
    from invcompcamtrack_amd import synth
    monkeypatch.setenv("ICTR_CPW", "64")
    w, h = 128, 96
    sc = (synth.make_scene(w, h, grid_step=8, margin=0.5, jitter=0.35, seed=5) if grid
          else synth.make_scene(w, h, n_points=900, margin=0.5, seed=6))
    n = sc["pts3d"].shape[1]
    op = ic.optparam(2, 0, 8, 3, 0.0, 0, 0, n)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    res = []
    for variant in (8192, 8192 | (1 << 28), 8192 | (1 << 27)):
        e = ic.TrackBatch(cam, op, 3)
        e.set_variant(variant)
        for k in range(3):
            e.Set3Dpoints(k, sc["pts3d"].copy())
            e.SetPose(k, sc["p_a"], pa, pb)
        e.track_async()
        p = e.poses()
        res.append((p, [e.read_buffer(k, q, 64 * n) for k in range(3) for q in (0, 1, 2)]))
    for k in (1, 2):
        for x, y in zip(res[0][1], res[k][1]):
            assert np.array_equal(x, y)
        assert np.abs(res[0][0] - res[k][0]).max() <= 2e-6
    assert np.abs(res[0][1][1]).max() > 0.1


def test_image_only_pyramid_small_problems_and_refusals():
    """getgrad = 2 reference pyramids: a small 8x8 problem leaves the one-launch tracker (which reads the gradient planes)
    for the per-iteration launches and gives THEIR result with a planes pyramid bit for bit; other patch sizes and the
    patch getter, which no on-the-fly kernel serves, are refused with a message."""
    from invcompcamtrack_amd import synth
    sc = synth.make_scene(320, 256, n_points=60, seed=3)
    op = ic.optparam(2, 0, 8, 3, 0.0, 0, 0, 60)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pa_img, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_a"], 2, 8, getgrad=2), ic.Pyramid(sc["img_b"], 2, 8, getgrad=0)
    res = []
    for ref, variant in ((pa_img, 0), (pa, 8192)):
        e = ic.TrackBatch(cam, op, 1)
        e.set_variant(variant)
        e.Set3Dpoints(0, sc["pts3d"].copy())
        e.SetPose(0, sc["p_a"], ref, pb)
        e.track_async()
        res.append((e.poses().copy(), e.path_name()))
    assert "k_track1" not in res[0][1] and np.array_equal(res[0][0], res[1][0])
    op4 = ic.optparam(2, 0, 4, 3, 0.0, 0, 0, 60)
    cam4 = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 4)
    e = ic.TrackBatch(cam4, op4, 1)
    e.Set3Dpoints(0, sc["pts3d"].copy())
    e.SetPose(0, sc["p_a"], ic.Pyramid(sc["img_a"], 2, 4, getgrad=2), ic.Pyramid(sc["img_b"], 2, 4, getgrad=0))
    with pytest.raises(ic.IctrError, match="on the fly"):
        e.track_async()
    with pytest.raises(ic.IctrError, match="gradient"):
        ic.util_getPatch_grad(pa_img, 0, np.array([[20.0, 20.0]], np.float32), op)


def test_image_only_and_host_plane_pyramids_do_not_mix_in_one_batch(oracle):
    """One launch serves all problems of a batch with ONE setup path: image-only reference pyramids (gradients on the
    fly) and pyramids made from caller-supplied planes (whose gradients are whatever the caller computed) cannot share it
    -- refused with a message, never a kernel reading planes that do not exist."""
    from invcompcamtrack_amd import synth
    sc = synth.make_scene(320, 256, n_points=700, seed=3)
    o = oracle.Pyramid(sc["img_a"], 2, 8)
    host = ic.Pyramid(lv_f=2, imgpadding=8, wh=(320, 256), host_planes=(o.img, o.dx, o.dy))
    img_only, pb = ic.Pyramid(sc["img_a"], 2, 8, getgrad=2), ic.Pyramid(sc["img_b"], 2, 8, getgrad=0)
    op = ic.optparam(2, 0, 8, 3, 0.0, 0, 0, 700)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    e = ic.TrackBatch(cam, op, 2)
    e.set_variant(8192)
    for k in range(2):
        e.Set3Dpoints(k, sc["pts3d"].copy())
    e.SetPose(0, sc["p_a"], host, pb)
    e.SetPose(1, sc["p_a"], img_only, pb)
    with pytest.raises(ic.IctrError, match="mix"):
        e.track_async()
    e.SetPose(1, sc["p_a"], host, pb)     # both from caller planes: the planes path, same poses for the same problem
    e.track_async()
    p = e.poses()
    assert np.array_equal(p[0], p[1])
