"""`src` under the reference's import name: hislam2 imports the model package as `src.dust3r.*` (/root/reference/hislam2/hi2.py:5,
track_frontend.py:8-11, track_backend.py:7-9).  Only the names the SLAM trackers import are provided; they resolve to the MI355X
runtime (`cut3r_slam_amd`), so the reference's tracker files run on the HIP kernels without an edit."""
