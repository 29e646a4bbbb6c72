"""Stand-in for the un-vendored third-party `mmcv` package (oracle import only).

Only used by tools/gen_golden.py in the build container to import the reference
from /root/reference; never shipped to the GPU box, never imported by the product.
"""
