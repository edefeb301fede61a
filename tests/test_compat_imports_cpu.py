"""CPU: the zero-edit model boundary (VERDICT r3 missing #2 / next #4).  With `cut3r_slam_amd/compat` on sys.path, every import line of the
reference's tracker files that names the model package (`src.dust3r.*`), `lietorch` or `curope` resolves to the MI355X runtime, and the
objects it yields carry the signatures SURVEY section 8(b) lists.  The import lines are written out here as strings (the reference does
not travel to the GPU box); where /root/reference is present (the build container) they are also compared with the files' own lines."""
import inspect
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "cut3r_slam_amd", "compat")
REF = "/root/reference"

# (file, line number, text) of every `import` / `from` line in lines 1-15 of the four tracker files that names src.dust3r / lietorch / curope
REFERENCE_IMPORTS = [
    ("hislam2/hi2.py", 5, "from src.dust3r.model import ARCroco3DStereo"),
    ("hislam2/track_frontend.py", 8, "from src.dust3r.inference import inference"),
    ("hislam2/track_frontend.py", 9, "from src.dust3r.utils.camera import pose_encoding_to_camera"),
    ("hislam2/track_frontend.py", 11, "from src.dust3r.utils.geometry import geotrf"),
    ("hislam2/track_backend.py", 6, "from lietorch import SE3"),
    ("hislam2/track_backend.py", 7, "from src.dust3r.inference import inference"),
    ("hislam2/track_backend.py", 8, "from src.dust3r.utils.camera import pose_encoding_to_camera"),
    ("hislam2/track_backend.py", 9, "from src.dust3r.utils.geometry import geotrf"),
    ("hislam2/factor_graph.py", 2, "import lietorch"),
    ("src/croco/models/curope/curope2d.py", 7, "import curope as _kernels  # run `python setup.py install`"),
]
# the second spelling of the same modules (the reference's model files import each other as `dust3r.*`)
SECOND_SPELLING = ["from dust3r.model import ARCroco3DStereo", "from dust3r.inference import inference",
                   "from dust3r.utils.camera import pose_encoding_to_camera", "from dust3r.utils.geometry import geotrf"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_the_import_strings_are_the_reference_files_own_lines():
    for rel, ln, text in REFERENCE_IMPORTS:
        with open(os.path.join(REF, rel)) as f:
            lines = f.read().splitlines()
        assert lines[ln - 1].strip() == text, (rel, ln, lines[ln - 1])
    # and no OTHER line in the first 15 of the four tracker files names these packages
    for rel in ("hislam2/hi2.py", "hislam2/track_frontend.py", "hislam2/track_backend.py", "hislam2/motion_filter.py"):
        with open(os.path.join(REF, rel)) as f:
            head = f.read().splitlines()[:15]
        named = {(rel, i + 1, l.strip()) for i, l in enumerate(head)
                 if l.strip().startswith(("import ", "from ")) and any(p in l for p in ("src.dust3r", "lietorch", "curope"))}
        assert named <= set(REFERENCE_IMPORTS), named - set(REFERENCE_IMPORTS)


def test_every_reference_import_line_resolves_from_compat_to_the_hip_runtime():
    """a fresh interpreter with ONLY compat/ prepended to sys.path (the one-line change INTEGRATION.md asks of a maintainer)"""
    prog = [f"import sys; sys.path.insert(0, {COMPAT!r})"]
    prog += [t.split("#")[0].strip() for _, _, t in REFERENCE_IMPORTS] + SECOND_SPELLING
    prog += [
        "import cut3r_slam_amd.model as M, cut3r_slam_amd.inference as I, cut3r_slam_amd.lietorch as L, cut3r_slam_amd.dust3r_utils as U, cut3r_slam_amd.ops as O",
        "import src.dust3r.model, dust3r.model",
        "assert src.dust3r.model.ARCroco3DStereo is M.Cut3rModel is dust3r.model.ARCroco3DStereo",
        "assert inference is I.inference and SE3 is L.SE3 and lietorch.SO3 is L.SO3 and lietorch.Sim3 is L.Sim3",
        "assert pose_encoding_to_camera is U.pose_encoding_to_camera and geotrf is U.geotrf and _kernels.rope_2d is O.rope_2d",
        "assert 'cut3r_slam_amd' in sys.modules['src.dust3r.model'].ARCroco3DStereo.__module__",
        "print('RESOLVED')",
    ]
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", "\n".join(prog)], cwd="/", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "RESOLVED" in r.stdout, r.stderr[-2000:]


def test_signatures_of_the_boundary_objects():
    """from_pretrained(path), normalize(img), encode_image(view), forward(views, ret_state), inference(groups, model, device,
    verbose=False), pose_encoding_to_camera(pose_encoding, pose_encoding_type='absT_quaR'), geotrf(Trf, pts, ncol=None, norm=False),
    rope_2d(tokens, positions, base, fwd), SE3.exp / .matrix / .data"""
    sys.path.insert(0, ROOT)
    from cut3r_slam_amd import dust3r_utils as U
    from cut3r_slam_amd import lietorch as L
    from cut3r_slam_amd import ops
    from cut3r_slam_amd.inference import inference
    from cut3r_slam_amd.model import ARCroco3DStereo
    par = lambda f: list(inspect.signature(f).parameters)
    assert par(ARCroco3DStereo.from_pretrained)[0] == "path"
    assert par(ARCroco3DStereo.normalize)[1:] == ["img_tensor"] and par(ARCroco3DStereo.encode_image)[1:] == ["view"]
    assert par(ARCroco3DStereo.forward)[1:] == ["views", "ret_state"] and ARCroco3DStereo.__call__ is ARCroco3DStereo.forward
    assert all(hasattr(ARCroco3DStereo, m) for m in ("to", "eval"))
    s = inspect.signature(inference)
    assert list(s.parameters) == ["groups", "model", "device", "verbose"] and s.parameters["verbose"].default is False
    s = inspect.signature(U.pose_encoding_to_camera)
    assert list(s.parameters) == ["pose_encoding", "pose_encoding_type"] and s.parameters["pose_encoding_type"].default == "absT_quaR"
    s = inspect.signature(U.geotrf)
    assert list(s.parameters) == ["Trf", "pts", "ncol", "norm"] and s.parameters["ncol"].default is None and s.parameters["norm"].default is False
    assert par(ops.rope_2d)[:4] == ["tokens", "positions", "base", "fwd"]
    for cls in (L.SE3, L.SO3, L.Sim3):
        assert all(hasattr(cls, m) for m in ("exp", "log", "inv", "matrix", "retr", "adjT", "__mul__", "__getitem__"))
    with pytest.raises(ValueError, match="Unknown pose encoding"):
        U.pose_encoding_to_camera(torch.zeros(1, 7), "relT")


def test_tensor_helpers_equal_the_reference_functions():
    """tests/golden/camera.npz = the reference's own pose_encoding_to_camera / quaternion_to_matrix / geotrf (every branch) / inv on the CPU"""
    sys.path.insert(0, ROOT)
    from cut3r_slam_amd import dust3r_utils as U
    f = np.load(os.path.join(ROOT, "tests", "golden", "camera.npz"))
    enc, c2w, pts, pl = (torch.from_numpy(f[k]) for k in ("enc", "c2w", "pts", "pts_list"))
    np.testing.assert_allclose(U.quaternion_to_matrix(enc[:, 3:]).numpy(), f["R"], rtol=0, atol=1e-6)
    got = U.pose_encoding_to_camera(enc)
    np.testing.assert_allclose(got.numpy(), f["c2w"], rtol=0, atol=1e-6)
    assert got.dtype == torch.float32 and got.shape == (9, 4, 4)
    np.testing.assert_allclose(U.geotrf(c2w, pts).numpy(), f["geotrf"], rtol=1e-6, atol=1e-6)                    # the trackers' call shape: [B,4,4] x [B,H,W,3]
    np.testing.assert_allclose(U.geotrf(c2w[2], pl[0]).numpy(), f["geotrf_single"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(U.geotrf(c2w, pl).numpy(), f["geotrf_batch"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(U.geotrf(c2w[:, :3, :3].contiguous(), pl).numpy(), f["geotrf_rot3"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(U.geotrf(c2w[:, :3, :3].contiguous(), pl, norm=2.0, ncol=2).numpy(), f["geotrf_norm"], rtol=1e-5, atol=1e-5)
    out = U.geotrf(c2w[3].numpy(), pl[1].numpy())
    assert isinstance(out, np.ndarray)
    np.testing.assert_allclose(out, f["geotrf_numpy"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(U.inv(c2w).numpy(), f["inv"], rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        U.inv([[1.0]])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_the_reference_tracker_files_import_unedited_on_top_of_compat():
    """the zero-edit claim itself, as far as a machine without a GPU can show it: the reference's OWN hislam2/track_frontend.py,
    track_backend.py and factor_graph.py are imported as they are, with compat/ ahead of everything and the reference's `src/` NOT on the
    path; the names they bound are this package's objects.  (cv2 / open3d / torchvision are absent from the image: empty harness-side
    stand-ins, as in tests/golden/make_fixtures.py; tqdm, scipy, networkx, sklearn, matplotlib are installed.)"""
    prog = f"""
import sys, types
for m in ("cv2", "open3d", "torchvision", "torchvision.transforms", "plyfile", "munch", "natsort"):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path[:0] = [{COMPAT!r}, {REF + "/hislam2"!r}]
assert not any(p.rstrip("/") in ({REF!r}, {REF + "/src"!r}) for p in sys.path)
import track_frontend, track_backend, factor_graph
import cut3r_slam_amd.inference as I, cut3r_slam_amd.dust3r_utils as U, cut3r_slam_amd.lietorch as L
assert track_frontend.__file__.startswith({REF!r}) and track_backend.__file__.startswith({REF!r})
assert track_frontend.inference is I.inference and track_backend.inference is I.inference
assert track_frontend.pose_encoding_to_camera is U.pose_encoding_to_camera and track_frontend.geotrf is U.geotrf
assert track_backend.SE3 is L.SE3 and factor_graph.lietorch.SE3 is L.SE3
print("UNEDITED")
"""
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", prog], cwd="/", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "UNEDITED" in r.stdout, r.stderr[-3000:]
