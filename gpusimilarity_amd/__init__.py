"""gpusimilarity_amd -- MI355X-native brute-force fingerprint similarity scan.

The product is the C-ABI library ``libgsim_hip.so`` (``include/gpusim_hip.h``):
hand-written gfx950 kernels behind the reference's FingerprintDB boundary.  This
package holds its sources (``csrc/``), the ctypes binding (``capi``), the Python
twin of ``gpusim::FingerprintDB`` (``fingerprintdb``) and the ``.fsim`` reader.
"""
__version__ = "0.1.0"
