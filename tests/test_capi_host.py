"""CPU-side checks of the C ABI: the library loads, exports every symbol of include/rsba.h, reads the
reference's file formats, writes its output formats, and refuses to solve without a GPU (no fallback)."""
import os
import re

import numpy as np
import pytest

import oracle_lib as ol
from realsensecalibration_amd import capi

G = ol.GOLDEN
ROOT = ol.ROOT


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__
    __graft_entry__.build()


def test_header_and_exports_agree():
    hdr = open(os.path.join(ROOT, "include", "rsba.h")).read()
    declared = set(re.findall(r"\b(rsba_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rsba_version() == 100


def test_load_correspondence_matches_reference_reader():
    intr = ol.read_intrinsics(ol.SERIALS_MAIN)
    p = capi.Problem.correspondence(os.path.join(G, "hongo", "correspondence.txt"), capi.MODEL_MARKER_CHAIN, ol.MARKER_SIDE_MAIN, intr)
    ref = ol.read_correspondence(os.path.join(G, "hongo", "correspondence.txt"))
    assert (p.num_times, p.num_cameras, p.num_markers, p.num_observations, p.num_parameters) == (6, 4, 11, 68, 126)
    assert np.array_equal(p.params, ref["params"])
    lib = capi.load()
    for i in range(68):
        assert lib.rsba_problem_camera_idx(p.h, i) == ref["c"][i]
        assert lib.rsba_problem_time_idx(p.h, i) == ref["t"][i]
        assert lib.rsba_problem_marker_idx(p.h, i) == ref["m"][i]
    # count x 4, as BALProblem::num_observations_per_time_camera returns
    assert p.num_observations_per_time_camera(0, 0) == 12 and p.num_observations_per_time_camera(5, 3) == 8
    p.close()


def test_marker_chain_from_arrays_equals_the_file_reader():
    """rsba_problem_create_marker_chain on the arrays of the committed file = the file reader; bad indices are refused."""
    intr = ol.read_intrinsics(ol.SERIALS_MAIN)
    path = os.path.join(G, "hongo", "correspondence.txt")
    ref = ol.read_correspondence(path)
    ref = dict(ref, intr=intr, marker_side=ol.MARKER_SIDE_MAIN)
    a = capi.Problem.correspondence(path, capi.MODEL_MARKER_CHAIN, ol.MARKER_SIDE_MAIN, intr)
    b = capi.Problem.marker_chain(ref)
    assert (b.num_times, b.num_cameras, b.num_markers, b.num_observations, b.num_parameters) == (6, 4, 11, 68, 126)
    assert np.array_equal(a.params, b.params)
    lib = capi.load()
    obs_a = np.ctypeslib.as_array(lib.rsba_problem_observations(a.h), shape=(68 * 8,))
    obs_b = np.ctypeslib.as_array(lib.rsba_problem_observations(b.h), shape=(68 * 8,))
    assert np.array_equal(obs_a, obs_b)
    for t in range(6):
        for c in range(4):
            assert a.num_observations_per_time_camera(t, c) == b.num_observations_per_time_camera(t, c)
    assert np.array_equal(a.point3d(), b.point3d())
    a.close(); b.close()
    bad = dict(ref, m=ref["m"].copy())
    bad["m"][3] = 11
    with pytest.raises(capi.RsbaError):
        capi.Problem.marker_chain(bad)
    with pytest.raises(capi.RsbaError):
        capi.Problem.marker_chain(ref, model=capi.MODEL_POINTS)


def test_synthetic_marker_chain_is_consistent(oracle):
    """The generator's truth reprojects to the noise level and the rows come in the reference file's order."""
    from realsensecalibration_amd import synthetic as syn
    p = syn.make_marker_chain(4, 30, 6, seed=3)
    assert p["obs"].shape == (p["N"], 8) and p["params"].shape == (6 * (4 + 30 + 6),)
    key = p["t"].astype(np.int64) * 10_000 + p["c"] * 100 + p["m"]
    assert np.all(np.diff(key) > 0)          # sorted, no detection twice
    cost = oracle.marker_chain_cost(p, 0, p["marker_side"], p["intr"], p["truth"])
    assert abs(cost / (0.5 * 8 * p["N"] * 0.3 ** 2) - 1.0) < 0.15
    assert np.all(p["params"][:6] == 0) and np.all(p["params"][6 * 34:6 * 35] == 0)


def test_set_camera_constant_argument_checks():
    """Problem::SetParameterBlockConstant mirror: point model only, camera index checked (no GPU needed)."""
    from realsensecalibration_amd import synthetic as syn
    p = capi.Problem.points(syn.make_problem(3, 20, 3, seed=1))
    p.set_camera_constant(0)
    p.set_camera_constant(2, False)
    for bad in (-1, 3):
        with pytest.raises(capi.RsbaError):
            p.set_camera_constant(bad)
    p.close()
    intr = ol.read_intrinsics(ol.SERIALS_MAIN)
    m = capi.Problem.correspondence(os.path.join(G, "hongo", "correspondence.txt"), capi.MODEL_MARKER_CHAIN, ol.MARKER_SIDE_MAIN, intr)
    with pytest.raises(capi.RsbaError):
        m.set_camera_constant(1)      # the marker-chain wiring fixes camera 0 / marker 0 itself
    m.close()


def test_configure_run_rejects_a_null_solver():
    """rsba_solver_configure_run (per-run Solver::Options): argument check only, no GPU needed."""
    lib = capi.load()
    assert lib.rsba_solver_configure_run(None, 10, 0) == capi.ERR_ARG


def test_intrinsics_xml_reader():
    for sn, ref in zip(ol.SERIALS_MAIN, ol.read_intrinsics(ol.SERIALS_MAIN)):
        assert np.array_equal(capi.read_intrinsics_xml(os.path.join(G, "intrinsics", sn + ".xml")), ref)
    with pytest.raises(capi.RsbaError) as e:
        capi.read_intrinsics_xml("/nonexistent.xml")
    assert e.value.code == capi.ERR_IO


def test_load_points_file_reference_and_extended(tmp_path):
    K = ol.read_intrinsics([ol.SERIALS_TEST2[1]])[0]
    p = capi.Problem.points_file(os.path.join(G, "two_cam_data.txt"), K)
    ref = ol.read_two_cam_data(os.path.join(G, "two_cam_data.txt"))
    assert (p.num_cameras, p.num_points, p.num_observations) == (1, 16, 16)
    assert np.array_equal(p.params, ref["params"])
    p.close()
    # extended header `C P N`
    f = tmp_path / "ext.txt"
    f.write_text("2 2 3\n0 0 1 2\n1 0 3 4\n1 1 5 6\n" + " ".join(str(i) for i in range(12 + 6)) + "\n")
    p = capi.Problem.points_file(str(f), K)
    assert (p.num_cameras, p.num_points, p.num_observations) == (2, 2, 3)
    assert capi.load().rsba_problem_point_idx(p.h, 2) == 1
    p.close()
    bad = tmp_path / "bad.txt"
    bad.write_text("1 2\n0 0 1\n")
    with pytest.raises(capi.RsbaError) as e:
        capi.Problem.points_file(str(bad), K)
    assert e.value.code == capi.ERR_FORMAT


def test_writers_reproduce_committed_outputs(tmp_path, oracle):
    """BAManager::Write on the reference's own final state: parameters from the oracle (pinned to the
    committed XML at 1e-15) go through the product's writers and are compared with the committed files."""
    intr = ol.read_intrinsics(ol.SERIALS_MAIN)
    ref = ol.read_correspondence(os.path.join(G, "hongo", "correspondence.txt"))
    final, _, _ = oracle.solve_marker_chain(ref, 0, ol.MARKER_SIDE_MAIN, intr)
    p = capi.Problem.correspondence(os.path.join(G, "hongo", "correspondence.txt"), capi.MODEL_MARKER_CHAIN, ol.MARKER_SIDE_MAIN, intr)
    p.params[:] = final
    xml, p3d = str(tmp_path / "Camera_Transform.xml"), str(tmp_path / "point3d.txt")
    p.write_outputs(xml, str(tmp_path), p3d)
    got, want = ol.read_opencv_xml(xml), ol.read_opencv_xml(os.path.join(G, "hongo", "Camera_Transform.xml"))
    assert set(got) == set(want)
    for k in want:
        assert got[k].shape == want[k].shape and np.abs(got[k] - want[k]).max() < 1e-12
    n, counts, pts = ol.read_point3d(p3d)
    n0, counts0, pts0 = ol.read_point3d(os.path.join(G, "hongo", "point3d.txt"))
    assert n == n0 and np.array_equal(counts, counts0) and np.abs(pts - pts0).max() < 2e-6
    assert np.abs(p.point3d() - pts0).max() < 6e-7
    for i in range(4):
        a = np.loadtxt(str(tmp_path / ("mat%d.txt" % i)))
        b = np.loadtxt(os.path.join(G, "extrinsics", "mat%d.txt" % i))
        assert np.abs(a - b).max() < 2e-6
    p.close()


def test_no_cpu_fallback():
    """Without a GPU every solve entry point must fail loudly, never compute on the host."""
    if capi.load().rsba_device_count() > 0:
        pytest.skip("GPU present")
    from realsensecalibration_amd import synthetic as syn
    prob = syn.make_problem(4, 50, 3, seed=1)
    p = capi.Problem.points(prob)
    before = p.params.copy()
    with pytest.raises(capi.RsbaError) as e:
        p.solve()
    assert e.value.code == capi.ERR_NO_DEVICE
    assert np.array_equal(before, p.params)
    with pytest.raises(capi.RsbaError):
        p.reprojection_error()
    p.close()


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "realsensecalibration_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(d, f)).read()
                for needle in ("oracle/", "liboracle", "oracle_lib", "ba_oracle", "oracle_capi", "import oracle"):
                    assert needle not in txt, (os.path.join(d, f), needle)
