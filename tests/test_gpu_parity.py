"""Parity of the HIP path (through the C-ABI) with the CPU oracle on identical seeded inputs.

Bars (stated here, used below):
  * bit-exact: pyramid planes, projections (all levels), reference patches T/Gx/Gy, sd coefficients, current
    patches at iteration 0, visibility masks -- everything that is element-wise arithmetic.
  * sums: H and b (J^T J, J^T r over all patch pixels) agree to 2e-6 relative to their largest entry: the
    reference sums in f32 in Eigen's unspecified order, the GPU in per-lane f32 + fixed-order f64 tree.
  * Gauss-Newton steps: |dp_gpu - dp_cpu|_inf <= 2e-3 |dp|_inf at the first iteration (the 6x6 system has
    condition ~1e4..1e6, so 1e-7 relative noise on H,b moves dp by up to 1e-3 relative; both sides carry that
    noise); later iterations additionally absorb the pose difference e they start from (dp(p+e) ~ dp(p) - e), and
    the poses after every update stay within 2e-5 (normalised units) of each other.
  * final pose: |p_gpu - p_cpu|_inf <= 1e-4 (BASELINE.json north_star), observed ~1e-6.

Why the per-iteration trajectory checks use frames below 256 px: the reference selects the bilinear taps with
ceil(x + .00001f) (utilities.cpp:66-67) "to round up full natural numbers". In float32 that addition is absorbed
for x >= 256, so a projected coordinate that lands EXACTLY on an integer >= 256 samples the whole patch one pixel
to the left -- a genuine discontinuity of the reference (reproduced bit for bit by the oracle and by the HIP
kernels, see test_get_patch_bit_exact_incl_borders). With coordinates ~300 it fires with probability ~3e-5 per
coordinate and iteration; which iteration hits it depends on the last ulp of the pose, so two correct
implementations can take visibly different steps at one iteration (observed: b jumps by 64*sd*gradient ~ 7e5
from a 6e-8 pose difference) and then re-converge. Below 256 px that form of the quirk cannot fire and trajectories are
comparable step by step; above, only first-iteration sums and final poses are.
A second, rarer form exists at every size: the ONE float32 value just below an integer n (x = n - 1 ulp, ulp < 1e-5)
gives ceil(x + .00001f) = n + 1 with floor(x) = n - 1, i.e. weight ~1 on the tap one pixel to the right. It fires with
probability ~1 ulp per coordinate and iteration (~1e-5 around x = 100); in the middle of a trajectory the next
iterations heal it, in the LAST iteration of a tracking it leaves a visible mark (observed once while writing the team
tests: 699 points, poses equal to 2e-7 up to the last iteration, then b jumps and the final poses differ by 3.9e-4 --
the oracle, the per-iteration launches and one team size on one side, two other team sizes on the other). Seeds below
are chosen so that no compared run ends on such a value; a test that newly fails with ONE problem off by 1e-5..1e-3
after a change of summation order should be checked for this with tools/scratch-style traces before anything else.
"""
import numpy as np
import pytest

import invcompcamtrack_amd as ic
from parity_util import Pair, rel, scene

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4
SUM_TOL = 2e-6
DP_TOL = 2e-3

LAUNCHES = 8192   # variant bit 13: per-iteration launch pairs (k_ref* / k_iter* + tails) whatever the problem size
ONE_LAUNCH = 16384  # variant bit 14: the one-launch tracker k_track1 (default choice for small problems)
SEPARATE_BEGIN = 1 << 18  # variant bit 18: uploads, projection launch and read-back copy as separate operations
NO_GRAPH = 32768  # variant bit 15: the per-iteration launches as plain launches (default below 65 536 points: one hipGraph)
NO_TEAMS = 1 << 19  # variant bit 19: the one-launch tracker with ONE workgroup per problem whatever its size
NO_RESIDENT = 1 << 21  # variant bit 21: never the resident-iteration form (k_level_resident)
RESIDENT = 1 << 23  # variant bit 23: the resident-iteration form whatever the batch size (default: up to 8 pairs)


@pytest.fixture(params=["one_launch", "teams", "launches"])
def launch_form(request):
    """Every case below that takes this fixture runs three times: as ONE launch of k_track1 with the library's own
    choice of workgroups per problem (one up to 128 points, a team with shares of 40-128 points above), as one launch with teams of
    16-point shares whatever the size (8x8 patches: many workgroups per problem, ragged and empty shares, the in-launch
    all-gather of the partial sums), and as the per-iteration launch sequence."""
    import parity_util
    parity_util.FORCE_VARIANT = LAUNCHES if request.param == "launches" else ONE_LAUNCH
    parity_util.FORCE_TEAM = (16, 0, 1 << 30) if request.param == "teams" else None
    yield request.param
    parity_util.FORCE_VARIANT = 0
    parity_util.FORCE_TEAM = None


def _check_setup_bit_exact(pr, levels):
    M, n = pr.M, pr.n
    for l in levels:
        g2, o2 = pr.odo.read_buffer(100 + l, 2 * M), pr.otr.pt2d(l)
        assert np.array_equal(g2[:n], o2[:n]) and np.array_equal(g2[M:M + n], o2[M:M + n]), f"pt2d level {l}"
    g3, o3 = pr.odo.read_buffer(5, 3 * M), pr.otr.buffer(5, 3 * M)
    for k in range(3):
        assert np.array_equal(g3[k * M:k * M + n], o3[k * M:k * M + n]), "pt3d_ref"


def _check_patches(pr, exact_T=True):
    nv, n = pr.op.novals, pr.n
    for w in (1, 2):
        assert np.array_equal(pr.odo.read_buffer(w, nv * n), pr.otr.buffer(w, nv * n)), f"patch buffer {w}"
    gT, oT = pr.odo.read_buffer(0, nv * n), pr.otr.buffer(0, nv * n)
    if exact_T:
        assert np.array_equal(gT, oT), "T patches"
    else:  # dopatchnorm: the patch mean is a 64-term sum in a different order
        assert np.abs(gT - oT).max() <= 1e-4


def _check_trace(pr, check_iters=True, traj_tol=2e-5):
    to, tg = pr.otr.trace(), pr.odo.trace()
    if check_iters:
        assert [(r["level"], r["iter"]) for r in to] == [(r["level"], r["iter"]) for r in tg]
    for a, b in zip(to, tg):
        if (a["level"], a["iter"]) != (b["level"], b["iter"]):
            break
        assert rel(a["H"], b["H"]) <= SUM_TOL, ("H", a["level"], a["iter"])
        # The two runs enter iteration k from poses that differ by e (accumulated rounding). Gauss-Newton is
        # self-correcting: dp(p + e) ~ dp(p) - e, so the updates may differ by ~|e| plus the solve's own noise,
        # and the poses AFTER every update must stay together.
        pa, pb = a["p"].astype(np.float64), b["p"].astype(np.float64)
        e = np.abs((pa - a["dp"]) - (pb - b["dp"])).max()
        ddp = np.abs(a["dp"].astype(np.float64) - b["dp"]).max()
        scale = max(1.0, np.abs(pa).max())
        assert ddp <= 2.0 * e + DP_TOL * np.abs(a["dp"]).max() + 0.1 * traj_tol * scale, ("dp", a["level"], a["iter"])
        assert np.abs(pa - pb).max() <= traj_tol * scale, ("pose trajectory", a["level"], a["iter"])
    a, b = to[0], tg[0]
    assert rel(a["b"], b["b"]) <= SUM_TOL, "b at the very first iteration (bit-identical inputs)"
    assert np.abs(a["dp"] - b["dp"]).max() <= DP_TOL * np.abs(a["dp"]).max(), "first dp"


@pytest.mark.parametrize("args", [
    (4, 0, 4, 5, 0.01, 0, 0),    # run_odometer_test.m:140
    (4, 0, 8, 10, 0.01, 1, 1),   # run_odometer_test.m:232
    (4, 0, 8, 10, 0.01, 1, 0),   # run_ransac_test.m:221 / func_ransac_fitcameras_odom.m
    (3, 1, 8, 6, 0.0, 0, 0),     # fixed iteration count, lv_l > 0
    (2, 0, 4, 8, 0.0, 0, 1),
])
def test_tracker_matches_oracle_reference_parameter_sets(oracle, args, launch_form):
    lv_f, lv_l, psz, maxiter, ratio, donorm, dpn = args
    # 256 x 224: every coordinate < 256 (no ceil(x+1e-5f) quirk), even sizes down to level 4;
    # 257 points: not a multiple of 4 -> maxpttrack padding path
    sc = scene(256, 224, 257, seed=31 + psz, margin=12.0)
    pr = Pair(oracle, sc, lv_f, lv_l, psz, maxiter, ratio, donorm, dpn)
    a, b = pr.set_points()
    assert np.array_equal(a, b)  # Set3Dpoints mutated both inputs identically (donorm)
    if donorm:
        assert not np.array_equal(a, sc["pts3d"])
        ms_o, v_o = pr.otr.norm()
        ms_g, v_g = pr.odo.norm()
        assert np.array_equal(ms_o, ms_g) and v_o == v_g
    pr.set_pose()
    assert np.array_equal(pr.odo.Get2DPoints()[:pr.n], pr.otr.pt2d(lv_l)[:pr.n])
    _check_setup_bit_exact(pr, range(lv_l, lv_f + 1))
    po, pg = pr.track()
    _check_patches(pr, exact_T=not dpn)
    _check_trace(pr)
    assert np.abs(po - pg).max() <= POSE_TOL
    assert np.abs(pg - sc["p_b"]).max() < 5e-3  # and it is the right answer


@pytest.mark.parametrize("args", [(4, 0, 8, 10, 0.01, 1, 1), (4, 0, 4, 5, 0.01, 0, 0)])
def test_tracker_matches_oracle_vga_frames(oracle, args, launch_form):
    """640 x 368 frames (coordinates beyond 256: the tap-selection quirk may fire at some iteration, see the module
    docstring): bit-exact setup, first-iteration sums, final pose."""
    lv_f, lv_l, psz, maxiter, ratio, donorm, dpn = args
    sc = scene(640, 368, 257, seed=31 + psz)
    pr = Pair(oracle, sc, lv_f, lv_l, psz, maxiter, ratio, donorm, dpn)
    pr.set_points()
    pr.set_pose()
    _check_setup_bit_exact(pr, range(lv_l, lv_f + 1))
    po, pg = pr.track()
    _check_patches(pr, exact_T=not dpn)
    a, b = pr.otr.trace()[0], pr.odo.trace()[0]
    assert rel(a["H"], b["H"]) <= SUM_TOL and rel(a["b"], b["b"]) <= SUM_TOL
    assert np.abs(a["dp"] - b["dp"]).max() <= DP_TOL * np.abs(a["dp"]).max()
    assert np.abs(po - pg).max() <= POSE_TOL
    assert np.abs(pg - sc["p_b"]).max() < 2e-3


@pytest.mark.parametrize("psz", [2, 16, 31, 64])
def test_extension_patch_sizes_match_oracle(oracle, psz, launch_form):
    """Patch sizes the reference cannot run (Eigen alignment) but whose geometry it defines
    (offsets -(P - P/2) ...). Oracle = our restatement: parity unpinned by the reference."""
    sc = scene(256, 224, 40, seed=40 + psz, margin=40.0)  # < 256 px: see the module docstring
    pr = Pair(oracle, sc, 2, 0, psz, 5, 0.0, 0, 1 if psz == 31 else 0)
    pr.set_points()
    pr.set_pose()
    po, pg = pr.track()
    _check_patches(pr, exact_T=psz != 31)
    # 40 points with 2x2 or 64x64 patches are poorly conditioned problems: intermediate poses are held to the
    # north-star bar (1e-4) rather than the tighter band used for the reference-runnable configurations
    _check_trace(pr, traj_tol=POSE_TOL)
    assert np.abs(po - pg).max() <= POSE_TOL


def test_identity_kat_on_gpu(oracle, launch_form):
    """Same image twice => delta_p == 0 => p_out == (double)(float)p_in (run_io_reprojection_test.cpp:15)."""
    sc = scene(640, 368, 100, seed=3)
    pr = Pair(oracle, sc, 3, 0, 8, 5, 0.1, 0, 0)
    pr.set_points()
    pr.otr.setpose(sc["p_a"], pr.opa, pr.opa)
    pr.odo.SetPose(sc["p_a"], pr.gpa, pr.gpa)
    po, pg = pr.track()
    assert np.array_equal(pg, sc["p_a"].astype(np.float32).astype(np.float64)) and np.array_equal(po, pg)
    tg = pr.odo.trace()
    assert all(np.all(r["dp"] == 0) for r in tg) and len(tg) == 4  # one iteration per level, then 0/0 stops the loop


def test_results_are_deterministic(oracle, launch_form):
    sc = scene(640, 368, 300, seed=8)
    outs = []
    for _ in range(3):
        pr = Pair(oracle, sc, 2, 0, 8, 6, 0.0, 0, 0)
        pr.odo.Set3Dpoints(sc["pts3d"].copy())
        pr.odo.SetPose(sc["p_a"], pr.gpa, pr.gpb)
        outs.append((pr.odo.TrackPose(), [r["dp"] for r in pr.odo.trace()]))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0])
        assert all(np.array_equal(x, y) for x, y in zip(o[1], outs[0][1]))


def test_fast_path_equals_generic_path(oracle):
    """P = 8 has a specialised kernel pair (k_ref8 / k_iter8); variant bit 1 forces the any-size kernels instead.
    Same arithmetic per element, different summation order: identical setup buffers, poses to float noise."""
    sc = scene(256, 224, 300, seed=9, margin=12.0)
    res = []
    # variant 256: fast path with H accumulated by the setup kernel instead of by the first iteration launch
    for variant in (0, 2, 256):
        for dpn in (0, 1):
            pr = Pair(oracle, sc, 2, 0, 8, 6, 0.0, 0, dpn, variant=variant | LAUNCHES)
            pr.set_points()
            pr.set_pose()
            po, pg = pr.track()
            assert np.abs(po - pg).max() <= POSE_TOL
            res.append((pg, pr.odo.read_buffer(0, 64 * 300), pr.odo.read_buffer(1, 64 * 300),
                        pr.odo.read_buffer(7, 16 * 300), pr.odo.trace()))
    # the 4x4 fast path (k_ref4 / k_iter4: four patches per wave-step) against the any-size kernels
    res4 = []
    for variant in (0, 2):
        for dpn in (0, 1):
            pr = Pair(oracle, sc, 2, 0, 4, 6, 0.0, 0, dpn, variant=variant | LAUNCHES)
            pr.set_points()
            pr.set_pose()
            po, pg = pr.track()
            assert np.abs(po - pg).max() <= POSE_TOL
            res4.append((pg, pr.odo.read_buffer(0, 16 * 300), pr.odo.read_buffer(1, 16 * 300),
                         pr.odo.read_buffer(2, 16 * 300), pr.odo.read_buffer(7, 16 * 300), pr.odo.trace()))
    for k in (0, 1):
        fast, gen = res4[k], res4[2 + k]
        assert np.abs(fast[0] - gen[0]).max() <= 2e-6
        assert all(np.array_equal(fast[i], gen[i]) for i in (1, 2, 3, 4))
        assert rel(fast[5][0]["H"], gen[5][0]["H"]) <= SUM_TOL and rel(fast[5][0]["b"], gen[5][0]["b"]) <= SUM_TOL
    for k in (0, 1):
        fast = res[k]
        for other in (res[2 + k], res[4 + k]):
            assert np.abs(fast[0] - other[0]).max() <= 2e-6
            assert np.array_equal(fast[1], other[1]) and np.array_equal(fast[2], other[2]) and np.array_equal(fast[3], other[3])
            assert len(fast[4]) == len(other[4])
            assert rel(fast[4][0]["H"], other[4][0]["H"]) <= SUM_TOL      # H of the first level, whoever summed it


@pytest.mark.parametrize("cfg", [
    (8, 300, 0, 0),    # 8x8, templates resident in LDS
    (8, 300, 1, 0),    # ... with patch normalisation
    (8, 900, 0, 0),    # 8x8, too many points for LDS templates: re-read from the (L2-resident) global patch buffers
    (4, 301, 0, 0),    # 4x4: four patches per wave, ragged last group
    (4, 301, 1, 1),    # ... patch normalisation + cloud normalisation
    (16, 40, 0, 0),    # any-size form: lanes loop over the pixels
    (31, 24, 1, 0),    # odd size
])
def test_one_launch_tracker_equals_per_iteration_launches(oracle, cfg):
    """k_track1 (whole odometer.cpp:257-426 loop in one kernel, one workgroup per problem) against the per-iteration
    launch sequence and against the oracle: identical element-wise buffers (patches, coefficients), H and b to the
    summation-order bar, same iteration counts, poses to float noise."""
    psz, npts, dpn, donorm = cfg
    sc = scene(256, 224, npts, seed=50 + psz, margin=float(max(12, psz + 9)))
    out = []
    # one workgroup per problem; the library's choice (8x8 patches above 128 points: a team of workgroups that
    # all-gather their partial sums inside the launch); the per-iteration launch sequence
    for variant in (ONE_LAUNCH | NO_TEAMS, ONE_LAUNCH, LAUNCHES):
        pr = Pair(oracle, sc, 2, 0, psz, 6, 0.0, donorm, dpn, variant=variant)
        pr.set_points()
        pr.set_pose()
        po, pg = pr.track()
        assert np.abs(po - pg).max() <= POSE_TOL
        nv = pr.op.novals
        out.append((pg, [pr.odo.read_buffer(w, nv * pr.n) for w in (0, 1, 2)], pr.odo.read_buffer(7, 16 * pr.n),
                    pr.odo.trace(), pr.odo.read_buffer(8, 40)))
        if variant != LAUNCHES:
            _check_patches(pr, exact_T=not dpn)
            _check_trace(pr, traj_tol=POSE_TOL if psz > 8 else 2e-5)
    many = out[2]
    for one in out[:2]:
        assert np.abs(one[0] - many[0]).max() <= 5e-6
        assert all(np.array_equal(a, b) for a, b in zip(one[1], many[1])), "patch buffers differ between the launch forms"
        assert np.array_equal(one[2], many[2]), "sd coefficients differ between the launch forms"
        assert [(r["level"], r["iter"]) for r in one[3]] == [(r["level"], r["iter"]) for r in many[3]]
        assert rel(one[3][0]["H"], many[3][0]["H"]) <= SUM_TOL and rel(one[3][0]["b"], many[3][0]["b"]) <= SUM_TOL


def test_team_form_batches_ragged_shares_and_repeated_launches(oracle):
    """The one-launch tracker's team form on a batch: problems with different point counts (shares of the batch's
    largest problem, so small problems leave whole workgroups without points), more workgroups than the chip has CUs
    (a team waits for peers that are dispatched later), the same engine launched again and again (tag epochs: no
    granule of an earlier launch may satisfy a poll), and a second frame pair with swapped roles. Reference: the same
    problems through the per-iteration launches; poses to float noise, iteration counts equal, and the team launch
    itself bit-reproducible."""
    sc = scene(256, 224, 700, seed=31, margin=12.0)  # below 256 px: trajectories comparable (module docstring)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    op = ic.optparam(2, 0, 8, 6, 0.0, 0, 0, 700)  # fixed iteration count: poses comparable to float noise
    B = 24
    counts = [700, 1, 0, 17, 350, 699, 96, 97] * 3
    rng = np.random.default_rng(4)  # (seed 3 ends one tracking on the tap-selection discontinuity: module docstring)
    poses0 = sc["p_a"][None, :] + rng.normal(0, 1e-3, (B, 6))

    def run(variant, team, reps=1, swap=False):
        e = ic.TrackBatch(cam, op, B)
        e.set_variant(variant)
        if team:
            e.set_team(*team)
        for k in range(B):
            e.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :counts[k]].copy()))
        outs = []
        for r in range(reps):
            e.SetPoseAll(poses0, pb if swap else pa, pa if swap else pb)
            e.track_async()
            outs.append((e.poses().copy(), e.iterations().copy()))
        return outs, e.path_name(), e.last_team()

    ref, name, _ = run(LAUNCHES, None)
    assert "k_iter" in name
    # poses to float noise where the system is well determined; a 17-point problem carries more of the summation
    # order, a 1-point problem is rank deficient (its update is whatever the rank decision leaves): finite and close
    # (r03: the per-iteration launches take H from three sums per patch, the one-launch tracker from 21 sums per pixel:
    # the same H to ~1e-7, so poses of ~0.5 agree to a few 1e-6 after 18 iterations instead of a few 1e-7)
    tol = np.array([2e-5 if c >= 90 else (1e-4 if c >= 10 else 1e-2) for c in counts])[:, None]

    def close(a, b):
        return bool(np.all(np.isfinite(a)) and np.all(np.abs(a - b) <= tol))
    for team, want in (((32, 0, 1 << 30), 22), ((96, 192, 6144), 8)):
        got, name, nteam = run(0, team, reps=4)
        assert "k_track1" in name and nteam == want, (name, nteam)
        assert all(np.array_equal(g[0], got[0][0]) and np.array_equal(g[1], got[0][1]) for g in got[1:])
        assert close(got[0][0], ref[0][0]), np.abs(got[0][0] - ref[0][0]).max(axis=1)
        strong = np.array(counts) >= 90
        assert np.array_equal(got[0][1][strong], ref[0][1][strong])
    refs, _, _ = run(LAUNCHES, None, swap=True)
    gots, _, nteam = run(0, (32, 0, 1 << 30), swap=True)
    assert nteam == 22 and close(gots[0][0], refs[0][0]), np.abs(gots[0][0] - refs[0][0]).max(axis=1)


def test_team_launches_on_many_streams_are_admitted_without_starving_each_other(oracle):
    """A team's workgroups wait for each other inside the kernel, so all of them must become resident. Six engines on
    six streams, each a lone 2500-point problem shared by 63 workgroups, launched back to back without waiting: their
    dispatch fronts together (6 x 62) could occupy every CU with partly resident teams. The library admits team launches
    so that sum(team - 1) stays below the CU count (later launches start behind the oldest ones in flight): every
    tracking must complete -- no exchange time-out -- with the poses of the same problems run one after the other."""
    torch = pytest.importorskip("torch")
    sc = scene(640, 480, 2500, seed=71)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    op = ic.optparam(2, 0, 8, 6, 0.0, 0, 0, 2500)
    rng = np.random.default_rng(12)
    poses0 = sc["p_a"][None, :] + rng.normal(0, 1e-3, (6, 6))
    streams = [torch.cuda.Stream() for _ in range(6)]
    engines = []
    for k in range(6):
        e = ic.TrackBatch(cam, op, 1)
        e.set_team(40, 0, 1 << 30)
        e.Set3Dpoints(0, sc["pts3d"].copy())
        engines.append(e)
    alone = []
    for k, e in enumerate(engines):  # one after the other, default stream
        e.SetPose(0, poses0[k], pa, pb)
        e.track_async()
        alone.append(e.poses().copy())
        assert e.last_team() == 63
    for k, e in enumerate(engines):
        e.set_stream(streams[k].cuda_stream)
    for rep in range(5):
        for k, e in enumerate(engines):
            e.SetPose(0, poses0[k], pa, pb)
        for e in engines:
            e.track_async()
        for k, e in enumerate(engines):
            assert np.array_equal(e.poses(), alone[k]), (rep, k)
    torch.cuda.synchronize()


@pytest.mark.parametrize("npw", [32, 16])
def test_transposing_wave_reduction_of_the_resident_kernel(npw):
    """k_level_resident sums the per-lane values of a wave (A = Gx r and B = Gy r of its 32 or 16 patches) over the 64 lanes
    with a transposing reduction -- bank-masked DPP adds, v_permlane16/32_swap, quad permutes -- that leaves lane l with the
    complete sum of ONE value. The reduction alone, on integers (exact in float32): every lane must hold exactly the sum
    of the value the kernel believes it holds (tr_patch_of_lane / tr_kind_of_lane), and the lanes must cover all values
    (with 16 patches per wave every value sits in two lanes)."""
    from invcompcamtrack_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(5)
    vals = rng.integers(-500, 500, (64, 64)).astype(np.float32)
    out = np.zeros(64, np.float32)
    pl, kl = np.zeros(64, np.int32), np.zeros(64, np.int32)
    _lib.check(L.ictr_debug_transpose_reduce(_lib.fp(vals), _lib.fp(out), pl.ctypes.data_as(_lib.IP),
                                             kl.ctypes.data_as(_lib.IP), npw))
    idx = 2 * pl + kl
    assert sorted(set(idx.tolist())) == list(range(2 * npw)), "the lanes must hold every value"
    assert np.bincount(idx, minlength=2 * npw).tolist() == [64 // (2 * npw)] * (2 * npw)
    want = vals.sum(axis=0)[idx]
    assert np.array_equal(out, want), (np.flatnonzero(out != want), out[:8], want[:8])


@pytest.mark.parametrize("B,ratio,maxiter", [(1, 0.0, 6), (3, 0.0, 6), (5, 0.01, 10), (2, 0.0, 1)])
def test_resident_iterations_equal_per_iteration_launches(oracle, B, ratio, maxiter):
    """k_level_resident (all iterations of a level in ONE launch: templates resident in registers / LDS, a mailbox
    gather + broadcast per iteration, one solver workgroup per frame pair) against the per-iteration launches: problems
    of 8300-9000 points with different counts (the last worker workgroups of a pair partly or wholly without points),
    more pairs than pairs in flight (a slot walks through several pairs), repeated launches (tag epochs), early exit
    (normdp_ratio 0.01: the loop flag travels with the pose) and a single iteration. Same setup kernels => patches and
    coefficients bit-identical; H and b by another summation order => poses to float noise, iteration counts equal."""
    sc = scene(640, 480, 9000, seed=83)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    op = ic.optparam(2, 0, 8, maxiter, ratio, 0, 0, 9000)
    counts = [9000 - 173 * k for k in range(B)]
    poses0 = sc["p_a"][None, :] + np.random.default_rng(9).normal(0, 1e-3, (B, 6))
    out = {}
    for name, variant in (("resident", RESIDENT), ("launches", NO_RESIDENT)):
        e = ic.TrackBatch(cam, op, B)
        e.set_variant(variant)
        for k in range(B):
            e.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :counts[k]].copy()))
        runs = []
        for rep_, (ra, rb) in enumerate(((pa, pb), (pa, pb), (pb, pa))):
            e.SetPoseAll(poses0, ra, rb)
            e.track_async()
            runs.append((e.poses().copy(), e.iterations().copy(),
                         [e.read_buffer(k, w, 64 * counts[k]) for k in range(B) for w in (0, 1, 2)],
                         [e.read_buffer(k, 7, 16 * counts[k]) for k in range(B)]))
        out[name] = (runs, e.path_name())
    assert "k_level_resident" in out["resident"][1] and "k_iter" in out["launches"][1], (out["resident"][1], out["launches"][1])
    for r, l in zip(out["resident"][0], out["launches"][0]):
        # 640 px frames: the tap-selection quirk of the module docstring; with an exit threshold a level may also end one
        # iteration earlier in one form (the last steps are ~1e-4)
        assert np.abs(r[0] - l[0]).max() <= (5e-5 if ratio == 0.0 else 2e-4), np.abs(r[0] - l[0]).max(axis=1)
        if ratio == 0.0:
            assert np.array_equal(r[1], l[1])
        assert all(np.array_equal(x, y) for x, y in zip(r[2], l[2])), "patch buffers differ between the launch forms"
        assert all(np.array_equal(x, y) for x, y in zip(r[3], l[3])), "sd coefficients differ between the launch forms"
    a, b_ = out["resident"][0][0], out["resident"][0][1]
    assert np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1]), "the same launch twice must give the same bits"


def test_resident_and_team_launches_share_the_gpu(oracle):
    """The two in-launch-synchronised forms on different streams at the same time: two engines with one dense
    9000-point pair each (k_level_resident: every workgroup of a launch must be resident) and two with a 2500-point
    problem (a team of 63 workgroups), launched back to back without waiting, five times. The admission keeps the
    launches in flight within the chip's workgroup slots; every tracking must complete (no exchange time-out) with
    exactly the poses of the same problems run one after the other."""
    torch = pytest.importorskip("torch")
    sc = scene(640, 480, 9000, seed=72)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    rng = np.random.default_rng(13)
    engines, want = [], []
    for k, npts in enumerate((9000, 2500, 9000, 2500)):
        op = ic.optparam(2, 0, 8, 6, 0.0, 0, 0, npts)
        e = ic.TrackBatch(cam, op, 1)
        e._op_keep = op
        e.Set3Dpoints(0, np.ascontiguousarray(sc["pts3d"][:, :npts].copy()))
        p0 = sc["p_a"] + rng.normal(0, 1e-3, 6)
        e.SetPose(0, p0, pa, pb)
        e.track_async()
        want.append((p0, e.poses().copy()))
        assert ("k_level_resident" if npts == 9000 else "workgroups per problem") in e.path_name(), e.path_name()
        engines.append(e)
    streams = [torch.cuda.Stream() for _ in engines]
    for e, st in zip(engines, streams):
        e.set_stream(st.cuda_stream)
    for rep in range(5):
        for e, (p0, _) in zip(engines, want):
            e.SetPose(0, p0, pa, pb)
        for e in engines:
            e.track_async()
        for k, (e, (_, p)) in enumerate(zip(engines, want)):
            assert np.array_equal(e.poses(), p), (rep, k)
    torch.cuda.synchronize()


def test_graph_replay_equals_plain_launches_and_follows_option_changes(oracle):
    """Launch-bound sizes replay the per-iteration launch sequence as one instantiated hipGraph (enqueue_levels): the
    SAME kernels with the SAME arguments, so every bit must agree with the plain launches -- also on the second
    tracking (graph reused), after new points / a new frame pair (data behind the same pointers), and after option
    changes that alter the captured launches (maxiter, dopatchnorm, normdp_ratio: the graph must be rebuilt)."""
    sc = scene(320, 240, 700, seed=77)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    ops = [ic.optparam(2, 0, 8, 6, 0.0, 0, 0, 700) for _ in range(2)]
    eng = []
    for op, variant in zip(ops, (LAUNCHES, LAUNCHES | NO_GRAPH)):
        e = ic.TrackBatch(cam, op, 2)
        e.set_variant(variant)
        eng.append(e)

    def both(npts, pose_shift, swap=False):
        res = []
        for e in eng:
            for k in range(2):
                e.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :npts - 50 * k].copy()))
                e.SetPose(k, sc["p_a"] + pose_shift * (k + 1), pb if swap else pa, pa if swap else pb)
            e.track_async()
            res.append((e.poses().copy(), e.iterations().copy(), e.path_name()))
        assert "hipGraph" in res[0][2] and "hipGraph" not in res[1][2], (res[0][2], res[1][2])
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        return res[0]

    first = both(700, 0.0)
    again = both(700, 0.0)
    assert np.array_equal(first[0], again[0])
    both(400, 1e-3, swap=True)
    for field, value in (("maxiter", 3), ("dopatchnorm", 1), ("normdp_ratio", 0.05), ("maxiter", 6)):
        for op in ops:
            setattr(op, field, value)
        r = both(700, 0.0)
        if field == "maxiter":
            assert int(r[1].max()) <= 3 * value
    for op in ops:
        op.dopatchnorm, op.normdp_ratio = 0, 0.0
    assert np.array_equal(both(700, 0.0)[0], first[0])


@pytest.mark.parametrize("w,h,npts,psz,dp_tol", [(640, 480, 4000, 8, 1e-5), (256, 224, 150, 8, 1e-4),
                                                  (640, 480, 6000, 4, 1e-5), (256, 224, 9000, 8, 1e-5)])
def test_updates_match_the_summation_order_free_cpu_path(oracle, w, h, npts, psz, dp_tol, launch_form):
    """per-iteration delta_p relative error <= 1e-5. Two float32 paths that add ~10^5-10^6 products in
    different orders cannot agree to that (the solve amplifies the sums' 1e-6..1e-5 by cond(H) ~ 1e4), so the yardstick
    is the CPU path with its whole-buffer sums accumulated in float64 (orc_set_sum_mode: same float32 products, same
    solver, same pose update -- only the summation order no longer matters). Against it the HIP path's first update
    (bit-identical inputs) is held to 1e-5 relative (1e-4 for the 150-point case: the float32 LU of a system with
    cond(H) ~ 1e5 carries that much on its own), H and b to 1e-6; the pose trajectory to 2e-5 absolute. The 9000-point
    case is a single dense problem: by default (everything but the "launches" form) it runs as k_level_resident -- all
    iterations of a level in one launch, H summed by the setup kernel."""
    if launch_form != "launches" and psz != 8 and npts * psz * psz > 512 * 64:
        pytest.skip("the one-launch tracker takes large problems as teams of 8x8-patch workgroups only")
    sc = scene(w, h, npts, seed=90 + psz, margin=float(max(12, psz + 9)))
    pr = Pair(oracle, sc, 2, 0, psz, 6, 0.0, 0, 0)
    pr.set_points()
    pr.set_pose()
    oracle.lib().orc_set_sum_mode(1)
    try:
        po, pg = pr.track()
    finally:
        oracle.lib().orc_set_sum_mode(0)
    to, tg = pr.otr.trace(), pr.odo.trace()
    assert [(r["level"], r["iter"]) for r in to] == [(r["level"], r["iter"]) for r in tg]
    relinf = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
    assert relinf(tg[0]["H"], to[0]["H"]) <= 1e-6
    assert relinf(tg[0]["b"], to[0]["b"]) <= 1e-6
    # ... when b came out bit-identical to the float64-sum path's (the usual case). ONE ulp in one entry of b (6e-8:
    # per-lane float32 accumulation, which no launch form can exclude) already moves the update by ~3e-5 here
    # (cond(H) ~ 3e4; observed with the team form's 125-point shares): then the bar is 1e-4.
    if not np.array_equal(tg[0]["b"], to[0]["b"]):
        dp_tol = max(dp_tol, 1e-4)
    assert relinf(tg[0]["dp"], to[0]["dp"]) <= dp_tol, "first delta_p vs the float64-sum CPU path"
    if max(w, h) <= 256:  # beyond 256 px the reference's ceil(x + 1e-5f) makes trajectories branch on rounding noise
        for a, b in zip(to, tg):
            assert np.abs(a["p"].astype(np.float64) - b["p"]).max() <= 2e-5
    assert np.abs(po - pg).max() <= POSE_TOL


@pytest.mark.parametrize("B,psz,team", [(24, 8, 0), (24, 8, 24), (20, 4, 0)])
def test_one_launch_projects_and_mirrors_for_batches_beyond_the_argument_blob(B, psz, team):
    """A batch whose records do not fit the kernel arguments (4 KB) uploads them in one copy; the one-launch tracker then
    projects (step 3) itself and writes the final records into the pinned host mirror -- no projection launch in
    front, no read-back copy behind. Every bit (poses, iteration counts, Get2DPoints, the stored state) must equal the
    form with the separate operations (variant bit 18), over repeated trackings with swapped frames."""
    sc = scene(256, 224, 90, seed=44 + psz, margin=float(max(12, psz + 9)))
    op = ic.optparam(3, 0, psz, 6, 0.0, 0, 0, 90)
    cam = ic.CamClass(4, sc["fc"], sc["cc"], sc["wh"], psz)
    pa, pb = ic.Pyramid(sc["img_a"], 3, psz), ic.Pyramid(sc["img_b"], 3, psz)
    out = []
    for variant in (0, SEPARATE_BEGIN):
        e = ic.TrackBatch(cam, op, B)
        e.set_variant(variant)
        if team:
            e.set_team(team, 0, 1 << 30)
        for k in range(B):
            e.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :90 - 7 * (k % 6)].copy()))
        res = []
        for rep, (ra, rb) in enumerate(((pa, pb), (pb, pa), (pa, pb))):
            for k in range(B):
                e.SetPose(k, sc["p_a"] + 1e-4 * k, ra, rb)
            e.track_async()
            res.append((e.poses().copy(), e.iterations().copy(), [e.Get2DPoints(k).copy() for k in range(B)],
                        [e.read_buffer(k, 8, 40) for k in range(B)]))
        out.append((res, e.path_name()))
        assert "k_track1" in e.path_name() and ("workgroups per problem" in e.path_name()) == bool(team), e.path_name()
    for ra, rb in zip(out[0][0], out[1][0]):
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
        assert all(np.array_equal(x, y) for x, y in zip(ra[2], rb[2]))
        assert all(np.array_equal(x[:18], y[:18]) for x, y in zip(ra[3], rb[3]))  # p, G of the stored state
    assert np.abs(out[0][0][0][0][:, :3] - sc["p_a"][:3]).max() > 1e-4  # (something was tracked)


@pytest.mark.parametrize("B,psz,team", [(1, 8, 0), (3, 8, 0), (2, 4, 0), (1, 5, 0), (1, 8, 16), (3, 8, 24)])
def test_one_launch_with_begin_phase_equals_separate_operations(oracle, B, psz, team):
    """Small batches go out as ONE launch that carries ictr_batch_begin's device part in its arguments and writes the
    final states into the pinned host mirror (track_enqueue): every bit -- poses, iteration counts, the projections
    Get2DPoints returns, the stored state -- must equal the form with separate uploads / projection launch / copy.
    team > 0: the same with every problem shared by workgroups of `team` points each (every workgroup projects its own
    points in the prologue, the first one stores the tables and the final state)."""
    sc = scene(256, 224, 90, seed=40 + psz, margin=float(max(12, psz + 9)))
    op = ic.optparam(3, 0, psz, 6, 0.0, 0, 0, 90)
    cam = ic.CamClass(4, sc["fc"], sc["cc"], sc["wh"], psz)
    pa, pb = ic.Pyramid(sc["img_a"], 3, psz), ic.Pyramid(sc["img_b"], 3, psz)
    out = []
    for variant in (0, SEPARATE_BEGIN):
        e = ic.TrackBatch(cam, op, B)
        e.set_variant(variant)
        if team:
            e.set_team(team, 0, 1 << 30)
        for k in range(B):
            e.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :90 - 7 * k].copy()))
        res = []
        for rep, (ra, rb) in enumerate(((pa, pb), (pb, pa), (pa, pb))):
            for k in range(B):
                e.SetPose(k, sc["p_a"] + 1e-3 * k, ra, rb)
            e.track_async()
            res.append((e.poses().copy(), e.iterations().copy(), [e.Get2DPoints(k).copy() for k in range(B)],
                        [e.read_buffer(k, 8, 40) for k in range(B)]))
        out.append((res, e.path_name()))
        assert ("workgroups per problem" in e.path_name()) == bool(team), e.path_name()
    assert "begin phase" in out[0][1] and "begin phase" not in out[1][1], (out[0][1], out[1][1])
    for ra, rb in zip(out[0][0], out[1][0]):
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
        assert all(np.array_equal(x, y) for x, y in zip(ra[2], rb[2]))
        assert all(np.array_equal(x[:18], y[:18]) for x, y in zip(ra[3], rb[3]))  # p, G of the stored state
    # the reference's single-problem API: Get2DPoints right after SetPose, TrackPose twice without a second SetPose
    if B == 1:
        got = []
        for variant in (0, SEPARATE_BEGIN):
            cam1 = ic.CamClass(4, sc["fc"], sc["cc"], sc["wh"], psz)
            pose = ic.PoseClass(cam1, op)
            odo = ic.OdometerClass(pose, op)
            odo.set_variant(variant)
            if team:
                odo.set_team(team, 0, 1 << 30)
            odo.Set3Dpoints(sc["pts3d"].copy())
            odo.SetPose(sc["p_a"], pa, pb)
            p2 = odo.Get2DPoints().copy()
            odo.SetPose(sc["p_a"], pa, pb)
            p_first = odo.TrackPose().copy()
            p_second = odo.TrackPose().copy()   # continues from the pose just found (odometer.cpp: no reset)
            got.append((p2, p_first, p_second))
        assert all(np.array_equal(x, y) for x, y in zip(got[0], got[1]))


def test_one_launch_tracker_gives_the_same_bits_in_both_register_budgets(oracle):
    """k_track1_p8 exists in two builds: 128 registers (two workgroups per CU; chosen when a batch has more problems than
    the chip has CUs) and ~200 registers (everything else). Same operations in the same order: a batch of 300 problems
    (first build) must give exactly the poses of the same problems tracked as two batches of 150 (second build) --
    which is also what keeps run_track_nposes' output independent of how the samples are split over ranks."""
    sc = scene(320, 240, 40, seed=61)
    op = ic.optparam(3, 0, 8, 5, 0.01, 0, 0, 40)
    cam = ic.CamClass(4, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 3, 8), ic.Pyramid(sc["img_b"], 3, 8)
    poses = sc["p_a"][None, :] + np.random.default_rng(8).normal(0, 2e-3, (300, 6))

    def run(idx):
        b = ic.TrackBatch(cam, op, len(idx))
        for k, i in enumerate(idx):
            b.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :25 + i % 16].copy()))
        b.SetPoseAll(poses[idx], pa, pb)
        b.track_async()
        out = (b.poses().copy(), b.iterations().copy())
        assert "k_track1" in b.path_name()
        return out

    whole = run(np.arange(300))
    halves = [run(np.arange(0, 150)), run(np.arange(150, 300))]
    assert np.array_equal(whole[0], np.concatenate([h[0] for h in halves]))
    assert np.array_equal(whole[1], np.concatenate([h[1] for h in halves]))


def test_setpose_all_equals_one_setpose_per_problem(oracle):
    """ictr_batch_setpose_all (one call for every pose sample of a frame pair) == the SetPose loop, bit for bit."""
    sc = scene(320, 240, 60, seed=31)
    op = ic.optparam(2, 0, 8, 5, 0.0, 0, 0, 60)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    poses = sc["p_a"][None, :] + np.random.default_rng(2).normal(0, 1e-3, (24, 6))
    out = []
    for mode in ("loop", "all"):
        b = ic.TrackBatch(cam, op, 24)
        for k in range(24):
            b.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :30 + k].copy()))
        if mode == "loop":
            for k in range(24):
                b.SetPose(k, poses[k], pa, pb)
        else:
            b.SetPoseAll(poses, pa, pb)
        b.track_async()
        out.append(b.poses().copy())
    assert np.array_equal(out[0], out[1])
    with pytest.raises(ValueError):
        b.SetPoseAll(poses[:5], pa, pb)


def test_one_launch_tracker_is_the_default_for_small_batches(oracle):
    """run_track_nposes' shape: many small independent problems = one workgroup each, one launch per frame pair;
    large problems keep the per-iteration launches."""
    # the automatic choice: <= 192 points alone, <= 384 per problem in a batch of >= 16, else per-iteration launches
    sc = scene(320, 240, 60, seed=21)
    op = ic.optparam(2, 0, 8, 5, 0.01, 0, 0, 60)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    res = {}
    for name, variant in (("auto", 0), ("launches", LAUNCHES)):
        b = ic.TrackBatch(cam, op, 40)
        b.set_variant(variant)
        rng = np.random.default_rng(5)
        for k in range(40):
            b.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :20 + k].copy()))  # ragged point counts
            b.SetPose(k, sc["p_a"] + rng.normal(0, 1e-3, 6), pa, pb)
        b.track_async()
        res[name] = (b.poses(), b.iterations(), b.path_name())
    assert "k_track1" in res["auto"][2] and "k_iter" in res["launches"][2]
    assert np.abs(res["auto"][0] - res["launches"][0]).max() <= 2e-5
    assert np.array_equal(res["auto"][1], res["launches"][1])
    big = scene(640, 368, 9000, seed=22)
    pba, pbb = ic.Pyramid(big["img_a"], 1, 8), ic.Pyramid(big["img_b"], 1, 8)
    camb = ic.CamClass(2, big["fc"], big["cc"], big["wh"], 8)
    for npts, want, team in ((9000, "k_level_resident", 1), (3000, "k_track1", 63), (300, "k_track1", 8),
                             (128, "k_track1", 1)):
        opb = ic.optparam(1, 0, 8, 2, 0.0, 0, 0, npts)
        bb = ic.TrackBatch(camb, opb, 1)
        bb.Set3Dpoints(0, np.ascontiguousarray(big["pts3d"][:, :npts].copy()))
        bb.SetPose(0, big["p_a"], pba, pbb)
        bb.track_async()
        bb.poses()
        assert want in bb.path_name() and bb.last_team() == team, (npts, bb.path_name())


def test_points_out_of_view_and_stale_state_across_frames(oracle, launch_form):
    """run_track_nposes chains SetPose/TrackPose without Set3Dpoints (run_track_nposes.cpp:232-258): points that
    leave the reference view keep their previous patches and sd coefficients in H and b (odometer.cpp:304).
    GPU and oracle must agree through such a chain."""
    sc = scene(640, 368, 200, seed=13)
    pr = Pair(oracle, sc, 3, 0, 8, 5, 0.0, 0, 0)
    pr.set_points()
    pr.set_pose()
    po, pg = pr.track()
    assert np.abs(po - pg).max() <= POSE_TOL
    # second frame pair of the chain: start from a pose that pushes ~half of the points out of the reference view
    p2 = po.copy()
    p2[0] += 2.4  # translate the camera: points shift by ~ f * 2.4 / depth = 180 px
    pr.otr.setpose(p2, pr.opb, pr.opa)
    pr.odo.SetPose(p2, pr.gpb, pr.gpa)
    vis0 = pr.otr.pt2d(0)[:pr.n]
    assert 0.15 < np.mean((vis0 < 0) | (vis0 > 640)) < 0.85
    _check_setup_bit_exact(pr, range(0, 4))
    po2, pg2 = pr.track()
    ind = pr.otr.ind(0)[:pr.n]
    assert 0 < ind.sum() < pr.n
    _check_patches(pr)          # includes the stale patches of the invisible points
    coef = pr.odo.read_buffer(7, 16 * pr.M).reshape(-1, 16)
    assert np.all(np.abs(coef[:pr.n, :12]).sum(1) > 0)  # invisible points still carry (stale) coefficients
    to, tg = pr.otr.trace(), pr.odo.trace()
    assert len(to) == len(tg)
    assert rel(to[0]["H"], tg[0]["H"]) <= SUM_TOL and rel(to[0]["b"], tg[0]["b"]) <= 1e-5
    assert np.abs(po2 - pg2).max() <= POSE_TOL * max(1.0, np.abs(po2).max())


def test_all_points_out_of_view_gives_zero_update(oracle, launch_form):
    sc = scene(320, 240, 60, seed=6)
    pr = Pair(oracle, sc, 2, 0, 8, 4, 0.0, 0, 0)
    pts = sc["pts3d"].copy()
    pts[0] += 1000.0
    pr.set_points(pts)
    pr.set_pose()
    po, pg = pr.track()
    assert np.array_equal(pg, sc["p_a"].astype(np.float32).astype(np.float64)) and np.array_equal(po, pg)
    assert all(np.all(r["H"] == 0) and np.all(r["dp"] == 0) for r in pr.odo.trace())


def test_empty_and_truncated_point_sets(oracle, launch_form):
    sc = scene(320, 240, 50, seed=7)
    pr = Pair(oracle, sc, 1, 0, 8, 3, 0.0, 0, 0, maxpt=24)  # more points than maxpttrack: truncated (odometer.cpp:182)
    assert pr.M == 24
    pr.set_points()
    pr.set_pose()
    po, pg = pr.track()
    assert np.abs(po - pg).max() <= POSE_TOL
    empty = np.zeros((3, 0))
    pr.otr.set3dpoints(empty.copy())
    pr.odo.Set3Dpoints(empty.copy(), 0)
    pr.set_pose()
    po, pg = pr.track()
    assert np.array_equal(po, pg) and np.array_equal(pg, sc["p_a"].astype(np.float32).astype(np.float64))


def test_nan_points_are_masked_not_fatal(oracle, launch_form):
    """The reference would index out of bounds on a NaN projection; the HIP path treats it as out of view."""
    sc = scene(320, 240, 64, seed=17)
    pr = Pair(oracle, sc, 1, 0, 8, 3, 0.0, 0, 0)
    pts = sc["pts3d"].copy()
    pts[:, 5] = np.nan
    pts[2, 9] = 0.0  # zero depth after the pose transform is unlikely exactly, but inf/NaN paths are covered by col 5
    pr.odo.Set3Dpoints(np.ascontiguousarray(pts))
    pr.odo.SetPose(sc["p_a"], pr.gpa, pr.gpb)
    out = pr.odo.TrackPose()
    assert np.all(np.isfinite(out)) and np.abs(out - sc["p_b"]).max() < 5e-3


def test_setpose_host_planes_equals_device_pyramids(oracle):
    """The reference's literal SetPose signature (host level-pointer arrays) against the device-pyramid form."""
    sc = scene(320, 240, 120, seed=19)
    pr = Pair(oracle, sc, 2, 0, 8, 5, 0.0, 0, 0)
    pr.set_points()
    pr.set_pose()
    _, pg = pr.track()
    odo2 = ic.OdometerClass(pr.pose, pr.op)
    odo2.Set3Dpoints(sc["pts3d"].copy())
    odo2.SetPose_host(sc["p_a"], pr.opa.img, pr.opa.dx, pr.opa.dy, pr.opb.img)  # the oracle's host planes
    assert np.array_equal(odo2.TrackPose(), pg)


def test_pose_class_projection_api(oracle):
    sc = scene(320, 240, 30, seed=23)
    pr = Pair(oracle, sc, 2, 0, 8, 1, 0.0, 0, 0)
    pr.set_points()
    pr.set_pose()
    M, n = pr.M, pr.n
    p3 = pr.otr.buffer(4, 3 * M)
    for sc_l in range(3):
        got = pr.pose.project_pt(p3, n, sc_l)
        assert np.array_equal(got[:n], pr.otr.pt2d(sc_l)[:n]) and np.array_equal(got[M:M + n], pr.otr.pt2d(sc_l)[M:M + n])
    rot, got = pr.pose.project_pt_save_rotated(p3, n, 2)
    assert np.array_equal(rot[:n], pr.otr.buffer(5, 3 * M)[:n])
    pr.pose.addpose_se3(np.array([1e-2, 0, 0, 0, 1e-3, 0], np.float32))
    assert not np.array_equal(pr.pose.project_pt(p3, n, 0)[:n], pr.otr.pt2d(0)[:n])


def test_batch_equals_single_problems(oracle):
    """B problems in one launch == B OdometerClass runs (different grid => sums to tolerance, not bitwise)."""
    scs = [scene(256, 224, 200 + 16 * k, seed=50 + k, margin=12.0) for k in range(3)]
    M = 256
    op = ic.optparam(2, 0, 8, 6, 0.01, 1, 0, M)
    cam = ic.CamClass(3, scs[0]["fc"], scs[0]["cc"], scs[0]["wh"], 8)
    batch = ic.TrackBatch(cam, op, 3)
    pyrs, singles = [], []
    for k, sc in enumerate(scs):
        pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
        pyrs.append((pa, pb))
        batch.Set3Dpoints(k, sc["pts3d"].copy())
        batch.SetPose(k, sc["p_a"], pa, pb)
        pose = ic.PoseClass(cam, op)
        odo = ic.OdometerClass(pose, op)
        odo.Set3Dpoints(sc["pts3d"].copy())
        odo.SetPose(sc["p_a"], pa, pb)
        singles.append((odo.TrackPose(), odo.Get2DPoints(), len(odo.enable_trace() or []) ))
    batch.track_async()
    poses = batch.poses()
    for k, sc in enumerate(scs):
        assert np.abs(poses[k] - singles[k][0]).max() <= 2e-6
        assert np.abs(poses[k] - sc["p_b"]).max() < 5e-3
        n = sc["pts3d"].shape[1]
        assert np.array_equal(batch.Get2DPoints(k)[:n], singles[k][1][:n])
    assert np.all(batch.iterations() >= 6) and np.all(batch.iterations() <= 18)


def test_sharded_phases_equal_fused_run(oracle):
    """Points split over two engines, their H / b partial sums added between the phases (what the RCCL
    all-reduce does across GPUs, emulated in one process): same poses as the unsharded run."""
    import torch
    from invcompcamtrack_amd.dist import RED_STRIDE, run_sharded_levels, shard_slices
    sc = scene(256, 224, 301, seed=61, margin=12.0)
    op = ic.optparam(2, 0, 8, 6, 0.01, 0, 0, 304)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    full = ic.TrackBatch(cam, op, 1)
    full.Set3Dpoints(0, sc["pts3d"].copy())
    full.SetPose(0, sc["p_a"], pa, pb)
    full.track_async()
    p_full = full.poses()[0]

    class TwoShards:
        def __init__(self):
            self.parts, self.red, self.n_allreduce = [], [], 0
            for lo, hi in shard_slices(301, 2):
                b = ic.TrackBatch(cam, op, 1)
                b.enable_sharding(True)
                r = torch.zeros(RED_STRIDE, dtype=torch.float32, device="cuda")
                b.set_reduction_buffer(r.data_ptr())
                b.Set3Dpoints(0, np.ascontiguousarray(sc["pts3d"][:, lo:hi]))
                b.SetPose(0, sc["p_a"], pa, pb)
                self.parts.append(b)
                self.red.append(r)

        @property
        def needs_level_allreduce(self):
            return self.parts[0].needs_level_allreduce

        def __getattr__(self, name):
            def call(*a):
                for b in self.parts:
                    getattr(b, name)(*a)
            return call

        def allreduce(self):
            self.n_allreduce += 1
            torch.cuda.synchronize()
            s = self.red[0] + self.red[1]
            for r in self.red:
                r.copy_(s)
            torch.cuda.synchronize()

    eng = TwoShards()
    run_sharded_levels(eng, op, eng.allreduce)
    p0, p1 = eng.parts[0].poses()[0], eng.parts[1].poses()[0]
    assert np.array_equal(p0, p1)                 # both "ranks" hold the same pose bits
    assert np.abs(p0 - p_full).max() <= 2e-6
    assert list(eng.parts[0].iterations()) == list(full.iterations())
    assert not eng.needs_level_allreduce and eng.n_allreduce == 3 * 6   # 8x8 fast path: H rides with the first b


@pytest.mark.parametrize("wh", [(1920, 1080), (1280, 720)])
def test_full_size_properties(oracle, wh):
    """BASELINE.json sizes: properties instead of the (slow) oracle: identity => exact pose; known motion is
    recovered; a batch repeats bit for bit; the frame-tiling grid really covers 2 073 600 px at 1080p."""
    w, h = wh
    sc = scene(w, h, None, seed=71, grid_step=8, margin=4.0)
    n = sc["pts3d"].shape[1]
    assert n == (w // 8) * (h // 8)
    op = ic.optparam(2, 0, 8, 10, 0.0, 0, 0, n)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    batch = ic.TrackBatch(cam, op, 3)
    for k in range(3):
        batch.Set3Dpoints(k, sc["pts3d"].copy())
    batch.SetPose(0, sc["p_a"], pa, pb)
    batch.SetPose(1, sc["p_a"], pa, pa)   # identity
    batch.SetPose(2, sc["p_a"], pa, pb)
    batch.track_async()
    poses = batch.poses()
    assert np.array_equal(poses[1], sc["p_a"].astype(np.float32).astype(np.float64))
    assert np.array_equal(poses[0], poses[2])
    assert np.abs(poses[0] - sc["p_b"]).max() < 1e-3
    assert list(batch.iterations()) == [30, 3, 30]
    # oracle on the same full-size input, one problem (about a second of CPU)
    oop = oracle.make_op(2, 0, 8, 10, 0.0, 0, 0, n)
    tr = oracle.Tracker(oop, sc["fc"], sc["cc"], sc["wh"])
    tr.set3dpoints(sc["pts3d"].copy())
    tr.setpose(sc["p_a"], oracle.Pyramid(sc["img_a"], 2, 8), oracle.Pyramid(sc["img_b"], 2, 8))
    assert np.abs(tr.trackpose() - poses[0]).max() <= POSE_TOL


def test_maxiter_zero_and_timing_getters(oracle):
    """maxiter = 0: no iteration launch at all (with deferred H nothing ever sums H), the pose comes back as the
    float32 round trip of the input, like the reference's loop that never runs (odometer.cpp:344-346)."""
    sc = scene(256, 224, 64, seed=5, margin=20.0)
    op = ic.optparam(2, 0, 8, 0, 0.0, 0, 0, 64)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    b = ic.TrackBatch(cam, op, 1)
    b.Set3Dpoints(0, sc["pts3d"].copy())
    b.SetPose(0, sc["p_a"], ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8))
    b.track_async()
    assert np.array_equal(b.poses()[0], sc["p_a"].astype(np.float32).astype(np.float64))
    assert list(b.iterations()) == [0]
    # per-launch event timing: first launch of a level reported separately from the sum
    op2 = ic.optparam(2, 0, 8, 3, 0.0, 0, 0, 64)
    b2 = ic.TrackBatch(cam, op2, 1)
    b2.set_timing(True)
    b2.Set3Dpoints(0, sc["pts3d"].copy())
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    b2.SetPose(0, sc["p_a"], pa, pb)
    b2.track_async()
    b2.poses()
    tot, first = b2.kernel_times(), b2.first_iter_times()
    assert np.all(first > 0) and np.all(tot > first)


def test_two_host_threads_drive_spin_waiting_launches_at_once(oracle):
    """admission of the in-launch-synchronised forms used to be check-then-act -- two host threads (ctypes
    releases the GIL) could both pass the budget test before either launch was visible, oversubscribe the CUs and run
    every poll into its time-out. Admit + launch + record is one critical section per device now: a resident-form engine
    (two pairs of 9000 points: 2 x 72 workgroups) and a team-form engine (eight problems x 63 workgroups) tracked from two
    threads at once, ten times, give their single-thread poses bit for bit and no time-out."""
    import threading
    torch = pytest.importorskip("torch")
    from invcompcamtrack_amd import synth
    sc = synth.make_scene(640, 384, n_points=9000, seed=3)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res = ic.TrackBatch(cam, ic.optparam(2, 0, 8, 5, 0.0, 0, 0, 9000), 2)
    team = ic.TrackBatch(cam, ic.optparam(2, 0, 8, 5, 0.0, 0, 0, 2500), 8)
    team.set_team(40, 0, 1 << 30)
    for k in range(2):
        res.Set3Dpoints(k, sc["pts3d"].copy())
    for k in range(8):
        team.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, :2500]))
    P2, P8 = np.tile(sc["p_a"], (2, 1)), np.tile(sc["p_a"], (8, 1))

    def run(e, P, n, out, errs):
        try:
            for _ in range(n):
                e.SetPoseAll(P, pa, pb)
                e.track_async()
                out.append(e.poses().copy())
        except Exception as exc:   # an exchange time-out surfaces here
            errs.append(exc)

    ref_r, ref_t, errs = [], [], []
    run(res, P2, 1, ref_r, errs)
    run(team, P8, 1, ref_t, errs)
    assert not errs and "k_level_resident" in res.path_name() and team.last_team() > 1
    res.set_stream(s1.cuda_stream)
    team.set_stream(s2.cuda_stream)
    out_r, out_t = [], []
    th = [threading.Thread(target=run, args=(res, P2, 10, out_r, errs)),
          threading.Thread(target=run, args=(team, P8, 10, out_t, errs))]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    torch.cuda.synchronize()
    assert not errs, errs
    assert len(out_r) == 10 and len(out_t) == 10
    assert all(np.array_equal(p, ref_r[0]) for p in out_r) and all(np.array_equal(p, ref_t[0]) for p in out_t)


@pytest.mark.timeout(600)
def test_headline_batch_properties_at_full_size():
    """The headline's configuration itself (32 frame pairs of 32 400 patches at 1080p, resident form: four pairs in
    flight, eight rounds per slot): 32 copies of ONE problem must come out bit-identical whichever slot and round ran
    them, and equal the same problem in a batch of 5 (two rounds) to float noise; an identity pair among them returns its
    start pose exactly; all iterations executed. The reference frame is an image-only pyramid (getgrad = 2)."""
    from invcompcamtrack_amd import synth
    sc = synth.make_scene(1920, 1080, grid_step=8, margin=4.0, jitter=0.35, seed=100)
    n = sc["pts3d"].shape[1]
    assert n == 32400
    op = ic.optparam(2, 0, 8, 10, 0.0, 0, 0, n)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8, getgrad=2), ic.Pyramid(sc["img_b"], 2, 8, getgrad=0)
    out = {}
    for B in (32, 5):
        e = ic.TrackBatch(cam, op, B)
        for k in range(B):
            e.Set3Dpoints(k, sc["pts3d"].copy())
            e.SetPose(k, sc["p_a"], pa, pa if k == 3 else pb)
        e.track_async()
        out[B] = e.poses().copy()
        assert "k_level_resident" in e.path_name()
        it = e.iterations()
        assert all(it[k] == (3 if k == 3 else 30) for k in range(B)), it
        del e
    p32, p5 = out[32], out[5]
    assert np.array_equal(p32[3], sc["p_a"].astype(np.float32).astype(np.float64))
    others = [k for k in range(32) if k != 3]
    assert all(np.array_equal(p32[k], p32[0]) for k in others)
    # another batch size: the setup kernel's chunk size follows the batch's total point count, so H is summed in another
    # order (float noise in the pose); the identity pair is exact either way
    assert np.abs(p5[0] - p32[0]).max() <= 2e-6 and np.array_equal(p5[3], p32[3])
    assert np.abs(p32[0] - sc["p_b"]).max() < 1e-3


def test_points_in_locality_order_give_the_same_tracking():
    """locality_order only permutes the caller's points: same poses up to the order of summation, and what comes back per
    point (Get2DPoints) is the unordered run's after undoing the permutation."""
    from invcompcamtrack_amd import synth
    sc = synth.make_scene(640, 384, n_points=9000, seed=8)
    order = ic.locality_order(sc["pts3d"], sc["p_a"], sc["fc"], sc["cc"])
    inv = np.argsort(order)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    op = ic.optparam(2, 0, 8, 6, 0.0, 0, 0, 9000)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    out = []
    for pts in (sc["pts3d"], np.ascontiguousarray(sc["pts3d"][:, order])):
        e = ic.TrackBatch(cam, op, 1)
        e.Set3Dpoints(0, pts.copy())
        e.SetPose(0, sc["p_a"], pa, pb)
        e.track_async()
        out.append((e.poses()[0].copy(), e.Get2DPoints(0).copy()))
    assert np.abs(out[0][0] - out[1][0]).max() <= 5e-6
    M = op.maxpttrack
    x0, y0 = out[0][1][:9000], out[0][1][M:M + 9000]
    x1, y1 = out[1][1][:9000], out[1][1][M:M + 9000]
    assert np.array_equal(x1[inv], x0) and np.array_equal(y1[inv], y0)
