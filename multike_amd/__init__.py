"""multike_amd — MI355X (gfx950) native hot path of MultiKE training behind the reference's op surface.

Product code: hand-written HIP kernels in `csrc/` exported through the C-ABI `include/multike_hip.h`
(`libmultike_hip.so`), and the Python host side mirroring the reference's `losses.py`,
`MultiKE_model.py`, `base/batch.py`, `attr_batch.py` interfaces.  There is no CPU fallback: importing the
package works anywhere, calling an op without the built library or without a GPU raises.
"""
__version__ = "0.1.0"
