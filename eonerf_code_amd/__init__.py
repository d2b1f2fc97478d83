"""eonerf_code_amd -- MI355X-native implementation of the EO-NeRF per-ray hot path (rogermm14/eonerf_code).

Drop-in surface (same names and signatures as the reference modules they replace):
    eonerf_code_amd.radiance_fields.eonerf.EONerfMLP      <- radiance_fields/eonerf.py:69-248
    eonerf_code_amd.sat_rendering.render_image            <- sat_rendering.py:176-335
    eonerf_code_amd.datasets.satellite.SatRays, define_satrays_from_tensors   <- datasets/satellite.py:21-26
Everything numerical runs in csrc/libeonerf_hip.so (hand-written HIP for gfx950) behind include/eonerf_hip.h.
"""
__version__ = "0.1.0"
