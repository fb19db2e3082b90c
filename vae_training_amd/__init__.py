"""MI355X-native ELBO train step: hand-written gfx950 HIP kernels behind a C ABI (libvaek.so),
with Python host code mirroring the reference's networks.py / vae.py / run.py interface.

The compute path has NO CPU fallback: anything that launches work raises if libvaek.so is not
built (``python -c "import __graft_entry__ as g; g.build()"``) or no MI355X is visible.
"""
__version__ = "0.1.0"
