"""Import-name shim (see ../README.md): `mmdet` as far as ufpmp_det_eval.py uses it."""
__version__ = "2.19.1"      # ufp/mmdet/version.py:3 (the fork the reference vendors)
