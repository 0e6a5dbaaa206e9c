"""Import-name shim (see ../README.md): bbox COCO API = glsdet_amd.eval (matching on the GPU)."""
