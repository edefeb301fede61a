"""`simple_knn` under the reference's import name (`from simple_knn._C import distCUDA2`:
/root/reference/hislam2/gaussian/scene/gaussian_model.py:18).  The extension is not vendored in the reference tree; `_C.distCUDA2` here is the
gfx950 kernel `cut3r_knn3_mean_dist2`."""
