"""weatherforecastingtoolkit_amd — MI355X-native (gfx950) conv-autoencoder
training hot path of Autobot37/weatherforecastingtoolkit: hand-written HIP
kernels behind a C ABI (libwfae.so), driven from the reference's own module
API.  Importing the package does not touch the GPU; the compute path raises
if libwfae.so is missing (there is no CPU fallback)."""
__version__ = "0.1.0"
