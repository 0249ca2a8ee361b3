"""Modules with the exact pybind11 signatures of the reference's three torch extensions
(setup.py:12-37).  Either add this directory to sys.path or call cdv_slam_amd.install_dropin()."""
