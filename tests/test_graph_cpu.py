"""CPU: host logic of the product FactorGraph (edge order, de-duplication, ages, distance classes, loop candidates)
replayed on the reference's golden graph fixture with the oracle overlap backend -- topology must be bit-exact."""
import os

import numpy as np
import torch

from cut3r_slam_amd.factor_graph import FactorGraph, SubmapStore
from cut3r_slam_amd import geom_host as gh
from oracle.slam_oracle import OracleOverlapBackend

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _replay(use_store):
    f = np.load(os.path.join(GOLD, "graph.npz"))
    pm = torch.from_numpy(f["pointmaps"])
    n = pm.shape[0]
    c2w = gh.pose_vec_to_matrix(f["poses"])
    np.testing.assert_allclose(c2w, f["c2w"], atol=1e-6)                    # pose_vec_to_matrix parity
    graph = FactorGraph(None, device="cpu", max_factors=48, backend=OracleOverlapBackend())
    store = torch.zeros(n // 5 + 1, 6, *pm.shape[1:])
    for j in range(n):
        store[j // 5, j % 5] = pm[j]
    graph.add_neighborhood_factors(0, 3, r=3)
    for i in range(n):
        if i >= 6:
            graph.add_neighborhood_factors(i - 3, i + 1, r=3)
        if i > 2:
            allpm = SubmapStore(store, i) if use_store else pm[:i]
            graph.add(i, c2w[:i], allpm, c2w[i], pm[i], f["K"])
        ii, jj, age = graph.edges_numpy()
        np.testing.assert_array_equal(ii, f[f"ii_{i}"], err_msg=f"ii after keyframe {i}")
        np.testing.assert_array_equal(jj, f[f"jj_{i}"], err_msg=f"jj after keyframe {i}")
        np.testing.assert_array_equal(age, f[f"age_{i}"], err_msg=f"age after keyframe {i}")
    loop = graph.detect_loop(n - 1)
    assert sorted(loop.tolist()) == sorted(f["loop_last"].tolist())
    assert torch.equal(graph.ii, torch.from_numpy(f[f"ii_{n-1}"]))
    return graph


def test_graph_topology_matches_reference_tensor_inputs():
    _replay(False)


def test_graph_topology_matches_reference_resident_store():
    _replay(True)


def test_add_factors_filters_only_existing_edges_and_rm():
    g = FactorGraph(None, device="cpu", backend=OracleOverlapBackend())
    g.add_factors([0, 0, 1], [1, 1, 0])                 # duplicates INSIDE a batch are kept (reference :29-39)
    assert g.edges_numpy()[0].tolist() == [0, 0, 1]
    g.add_factors([0, 2], [1, 0])
    assert list(zip(*[a.tolist() for a in g.edges_numpy()[:2]])) == [(0, 1), (0, 1), (1, 0), (2, 0)]
    g.rm_factors(torch.tensor([True, False, False, True]))
    assert list(zip(*[a.tolist() for a in g.edges_numpy()[:2]])) == [(0, 1), (1, 0)]


def test_end_to_end_oracle_graph_replays_the_reference_edge_lists():
    """oracle/slam_run.RefGraph (the independent restatement used by the end-to-end trajectory oracle) against the reference
    FactorGraph run recorded in graph.npz: identical ordered edge lists and ages after every add."""
    import os
    import numpy as np
    from oracle import slam_run as SR
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "graph.npz"))
    c2w, pm, K = f["c2w"], f["pointmaps"], f["K"]
    K4 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], np.float32)
    n = c2w.shape[0]
    g = SR.RefGraph()
    g.add_neighborhood_factors(0, 3, r=3)
    for i in range(n):
        if i >= 6:
            g.add_neighborhood_factors(i - 3, i + 1, r=3)
        if i > 2:
            g.add(i, c2w[:i], pm[:i], c2w[i], pm[i], K4)
        ii, jj, age = g.edges_numpy()
        np.testing.assert_array_equal(ii, f[f"ii_{i}"])
        np.testing.assert_array_equal(jj, f[f"jj_{i}"])
        np.testing.assert_array_equal(age, f[f"age_{i}"])


def test_host_pose_helpers_equal_the_reference_fixture():
    import os
    import numpy as np
    import torch
    from cut3r_slam_amd import geom_host as gh
    from oracle import slam_oracle as SO
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "camera.npz"))
    np.testing.assert_allclose(gh.pose_encoding_to_camera(f["enc"]), f["c2w"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(gh.pose_vec_to_matrix(f["pose_vec"]), f["pose_vec_c2w"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(SO.pose_encoding_to_camera(torch.from_numpy(f["enc"])).numpy(), f["c2w"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(SO.pose_vec_to_matrix(torch.from_numpy(f["pose_vec"])).numpy(), f["pose_vec_c2w"], rtol=0, atol=1e-6)


def test_ate_alignment_equals_the_reference_umeyama():
    """the Sim(3) alignment behind every ATE number of the bench and the e2e tests (cut3r_slam_amd/eval_ate.umeyama) against the
    reference's util.utils.umeyama_alignment (:738-763, in camera.npz): a proper similarity and one through a reflection (the det rule)"""
    import os
    import numpy as np
    from cut3r_slam_amd.eval_ate import umeyama
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "camera.npz"))
    for name in ("proper", "reflected"):
        s, R, t = umeyama(f[f"um_{name}_src"], f[f"um_{name}_dst"], True)
        assert abs(s - float(f[f"um_{name}_scale"])) < 1e-12
        np.testing.assert_allclose(R, f[f"um_{name}_R"], atol=1e-12)
        np.testing.assert_allclose(t, f[f"um_{name}_t"], atol=1e-12)
        assert np.linalg.det(R) > 0.999
