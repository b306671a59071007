"""The host-only entry points (A16 FindRelativeMeasurementMatch, N2 Delaunay, N3 measurement picker / loop-closure region /
loop candidate) are tested WITHOUT a device in tests/test_abi.py.  The driver's round-end run selects `-m gpu` only, so the same
test functions are executed once more here on the GPU box ("the driver's -m gpu run never executed them")."""
import pytest

import test_abi

HOST_TESTS = ["test_host_logic_closest_stamp", "test_find_relative_meas_match_host_logic", "test_delaunay_equals_qhull",
              "test_pick_next_measurement_reference_scenarios", "test_in_loop_closure_region_matches_oracle",
              "test_loop_candidate_idx_matches_oracle", "test_every_declared_symbol_is_exported",
              "test_default_params_mirror_reference_defaults", "test_product_never_imports_oracle"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", HOST_TESTS)
def test_host_entry_point_on_gpu_box(gpu, name):
    getattr(test_abi, name)()


@pytest.mark.gpu
def test_merge_refuses_unknown_keys_loudly_gpu(gpu):
    """isam->update(fgraph, fvalues) throws on a factor whose key is unknown and on a value inserted twice (graph.cpp:262); the product
    refuses the entry, merges the rest, and fails the consuming call loudly."""
    import numpy as np
    g = gpu.SlideGraph(gpu.default_params())
    I7 = np.array([0, 0, 0, 0, 0, 0, 1.0])
    g.set_prior(0, I7)
    g.add_keypose_between(0, 0, 1, np.array([1.0, 0, 0, 0, 0, 0, 1]), np.array([1.0, 0, 0, 0, 0, 0, 1]))
    g.add_point_landmark(0, np.array([2.0, 1.0, 0.0]))
    g.add_range_bearing(0, 0, 0, np.array([2.0, 1.0, 0.0]) / np.sqrt(5.0), np.sqrt(5.0))
    g.add_range_bearing(0, 1, 0, np.array([1.0, 1.0, 0.0]) / np.sqrt(2.0), np.sqrt(2.0))
    g.solve()
    assert g.rejected_count() == 0
    g.add_range_bearing(0, 1, 7, np.array([1.0, 0, 0]), 1.0)          # landmark u7 was never inserted
    with pytest.raises(gpu.SlideError, match="u7"):
        g.solve()
    assert g.rejected_count() == 1
    g.solve()                                                          # the rest of the graph is intact
    assert g.get_pose(0, 1)[0] == 0
    g.add_loop_closure(I7, 0, 0, 5, 0)                                 # pose x5 does not exist
    with pytest.raises(gpu.SlideError, match="x5"):
        g.solve()
    g.add_point_landmark(0, np.array([9.0, 9.0, 9.0]))                 # ValuesKeyAlreadyExists
    with pytest.raises(gpu.SlideError, match="twice"):
        g.solve()
    assert g.rejected_count() == 3
    st, lm = g.get_landmark(2, 0)
    assert st == 0 and abs(lm[0] - 9.0) > 1.0                          # the resident value was not overwritten


@pytest.mark.gpu
def test_cpp_adaptor_runs(gpu, tmp_path):
    """The S1 / S2 adaptor classes of include/slide_sloam_adaptor.hpp driven the way graphWrapper.cpp / sloamNode.cpp drive the
    reference's: priors, odometry, all three landmark factor kinds, a loop closure, solve, the read-back methods with the reference's
    absent-key conventions (false + identity, zero point, std::out_of_range)."""
    import subprocess
    exe = test_abi._build_adaptor_check(tmp_path)
    r = subprocess.run([exe, "run"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adaptor ok=1 missing=1 threw=1" in r.stdout and "wrapper poses=1 counter=1" in r.stdout
