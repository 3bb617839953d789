"""turbdiff_amd -- MI355X-native drop-in for the denoising-diffusion hot path of
martenlienen/generative-turbulence (``turbdiff.models.ddpm`` and helpers).

    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion

keeps the reference's constructor signatures, method names and state_dict keys; the compute
runs in hand-written HIP kernels for gfx950 (``libtdx_hip.so``, C ABI in ``include/tdx.h``).
``turbdiff_amd.dropin.install()`` registers these modules under the reference's import paths.
"""

__version__ = "0.1.0"
