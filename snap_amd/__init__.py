"""snap_amd: MI355X-native hot path of SNAP (BEV neural-map fusion + pose matching).

Host code mirrors the reference's ``snap.models`` / ``snap.utils`` / ``snap.configs``
API (same class names, config keys and output pytrees); all arithmetic of the hot
path runs in hand-written HIP kernels behind the C ABI of ``include/snap_hip.h``.
"""
__version__ = '0.1.0'
