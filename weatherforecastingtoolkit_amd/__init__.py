"""weatherforecastingtoolkit_amd — MI355X-native (gfx950) conv-autoencoder
training hot path of Autobot37/weatherforecastingtoolkit: hand-written HIP
kernels behind a C ABI (libwfae.so), driven from the reference's own module
API.  Importing the package does not touch the GPU; the compute path raises
if libwfae.so is missing (there is no CPU fallback)."""
__version__ = "0.1.0"


def set_float32_matmul_precision(precision):
    """'highest' | 'high' (fp32 MFMA) | 'medium' (bf16 operands, fp32 accumulate) — see ops.set_float32_matmul_precision"""
    from . import ops
    ops.set_float32_matmul_precision(precision)


def get_float32_matmul_precision():
    from . import ops
    return ops.get_float32_matmul_precision()
